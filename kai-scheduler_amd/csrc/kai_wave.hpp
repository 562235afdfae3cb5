// kai_wave.hpp — wave64 reductions for gfx950 written with DPP (no LDS traffic, no ds_bpermute round trips).
#pragma once
#include <hip/hip_runtime.h>

namespace kai {

// Maximum of an unsigned 64-bit value over the 64 lanes of a fully active wavefront, returned in every lane.
// Inclusive max-scan with the gfx9 DPP controls row_shr:1,2,4,8 then row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3);
// lane 63 ends with the total.  `old` = 0 is the identity of unsigned max for lanes without a source.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#define KAI_DPP_STEP(ctrl, rmask)                                                                  \
    {                                                                                              \
        unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, ctrl, rmask, 0xf, false); \
        unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, ctrl, rmask, 0xf, false); \
        bool gt = ohi > hi || (ohi == hi && olo > lo);                                             \
        lo = gt ? olo : lo; hi = gt ? ohi : hi;                                                    \
    }
    KAI_DPP_STEP(0x111, 0xf) KAI_DPP_STEP(0x112, 0xf) KAI_DPP_STEP(0x114, 0xf) KAI_DPP_STEP(0x118, 0xf)
    KAI_DPP_STEP(0x142, 0xa) KAI_DPP_STEP(0x143, 0xc)
#undef KAI_DPP_STEP
    unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63), rhi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    return ((unsigned long long)rhi << 32) | rlo;
}

// Arg-max of (key, n) where ties go to the LOWEST LANE (lanes hold nodes / blocks in ascending index order, so that is the
// reference's name-ascending tie-break).  On return every lane holds the winning key and its n; key == 0 means "none".
__device__ __forceinline__ void wave_argmax_first(unsigned long long& key, int& n) {
    unsigned long long m = wave_max_u64(key);
    unsigned long long win = __ballot(key == m);
    int lane = __ffsll((long long)win) - 1;  // uniform
    n = __builtin_amdgcn_readlane(n, lane);
    key = m;
}

}  // namespace kai
