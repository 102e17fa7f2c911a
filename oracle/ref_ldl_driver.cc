// Driver that exposes the reference's own ldl_inverse (compiled from
// /root/reference/lib/ldl_decomposition.h where it lies) through a C symbol.
// Test infrastructure only; output goes to oracle/_ref/ (git-ignored).
#include <algorithm>
#include "defines.h"
#include "ldl_decomposition.h"

extern "C" void ref_ldl_inverse(double* A, int size)
{
    smvs::ldl_inverse<double>(A, size);
}
