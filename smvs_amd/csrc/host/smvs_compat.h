// Source compatibility with the reference's namespace: a translation unit
// written against flanggut/smvs (smvs::DepthOptimizer, smvs::StereoView,
// smvs::Surface, smvs::SGMStereo, smvs::GaussNewtonStep,
// smvs::ConjugateGradient, smvs::ViewSelection -- lib/defines.h:13-15 opens
// `namespace smvs`) compiles against this library by including this header
// instead of the reference's: the classes live in smvs_amd:: (so that both
// libraries can be linked into one program side by side, which is how the
// integration stub of INTEGRATION.md section 2 uses them), and `smvs` becomes
// an alias of it here.  Do not include it in a program that also includes the
// reference's own headers.
#pragma once

#include "conjugate_gradient.h"
#include "depth_optimizer.h"
#include "gauss_newton_step.h"
#include "sgm_stereo.h"
#include "stereo_view.h"
#include "surface.h"
#include "view_selection.h"

namespace smvs = smvs_amd;
