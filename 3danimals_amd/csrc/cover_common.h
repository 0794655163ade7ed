// Covered-pixel list helpers shared by cover.hip and gbuffer.hip (a3d_cover_gbuffer_fwd: list + G-buffer rows in one launch).
#pragma once
#include "a3d_common.h"

#ifdef __HIPCC__
// (all 32-bit: B*H*W < 2^31 is checked by the entry points, and a 64-bit division costs ~150 instructions per thread)
__device__ __forceinline__ long long cv_flat(long long k64, int H, int W, int tile) {
    if (tile == 0) return k64;
    const unsigned k = (unsigned)k64;
    const unsigned in_tile = k & 63u;
    unsigned t = k >> 6;
    const unsigned tw = (unsigned)W >> 3, th = (unsigned)H >> 3;
    const unsigned tx = t % tw; t /= tw;
    const unsigned ty = t % th; t /= th;  // t = image
    return (long long)((t * (unsigned)H + (ty * 8u + (in_tile >> 3))) * (unsigned)W + tx * 8u + (in_tile & 7u));
}

// entries of the list before work-group ``blk``: whole groups from the group sums, the rest of its own group from the block counts.
// Computed by the FIRST WAVE only (<= nb/64 + 63 loads, a few per lane, all in flight at once) while the other waves are busy with
// their pixels; the result reaches them through LDS at the barrier the kernel has anyway.
__device__ __forceinline__ int cv_block_offset_wave0(const int* __restrict__ block_count, const int* __restrict__ group_sum, int blk) {
    const int g = blk / A3D_COVER_GROUP, r = blk - g * A3D_COVER_GROUP;
    const int lane = threadIdx.x & 63;
    int mine = lane < r ? block_count[g * A3D_COVER_GROUP + lane] : 0;
    for (int j = lane; j < g; j += 64) mine += group_sum[(long long)j * A3D_COVER_GROUP_STRIDE];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    return mine;
}

#endif
