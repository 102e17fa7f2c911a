// SGM entry points (implemented in a later commit of this round).
#include "common.h"
using namespace smvs_hip;
extern "C" int
smvs_sgm_run(int, const uint8_t *, int, int, const uint8_t *, int, int,
    const float *, const float *, float, float, int, uint16_t, uint16_t,
    float *, int32_t *, uint16_t *, uint16_t *)
{
    set_error("smvs_sgm_run: not implemented yet");
    return SMVS_ERR_STATE;
}
extern "C" int
smvs_bilateral_upsample(int, const float *, int, int, const float *, int, int,
    int, float, int, float *)
{
    set_error("smvs_bilateral_upsample: not implemented yet");
    return SMVS_ERR_STATE;
}
