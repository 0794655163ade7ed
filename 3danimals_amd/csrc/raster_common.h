// Pieces of the rasteriser shared with the kernels that walk a triangle's pixel box again (csrc/gbuffer.hip).
#pragma once
#include "a3d_common.h"

#ifdef __HIPCC__
// conservative pixel box (oracle/raster_ref.c's, up to the rounding of its six divisions); returns the number of candidate pixels (0 = culled)
__device__ __forceinline__ int rs_box(const float4 p0, const float4 p1, const float4 p2, int H, int W, int& x0, int& y0, int& bw) {
    int x1, y1;
    if (p0.w > 0.f && p1.w > 0.f && p2.w > 0.f) {
        // (v_rcp_f32 instead of six correctly rounded divisions -- ~150 of this function's ~250 instructions, in a launch whose setup
        // is three quarters of its time: the box only has to CONTAIN the covered pixel centres, which lie >= 1/32 px inside it, and
        // 1 ulp on a screen coordinate is 1.5e-5 px)
        const float r0 = __builtin_amdgcn_rcpf(p0.w), r1 = __builtin_amdgcn_rcpf(p1.w), r2 = __builtin_amdgcn_rcpf(p2.w);
        const float sx0 = p0.x * r0, sx1 = p1.x * r1, sx2 = p2.x * r2;
        const float sy0 = p0.y * r0, sy1 = p1.y * r1, sy2 = p2.y * r2;
        const float mnx = fminf(sx0, fminf(sx1, sx2)), mxx = fmaxf(sx0, fmaxf(sx1, sx2));
        const float mny = fminf(sy0, fminf(sy1, sy2)), mxy = fmaxf(sy0, fmaxf(sy1, sy2));
        // pixel centres px+0.5 inside [min,max], widened by 1/32 px (coverage itself is decided by rs_frag)
        const float fx0 = ceilf((mnx + 1.f) * 0.5f * W - 0.53125f), fx1 = floorf((mxx + 1.f) * 0.5f * W - 0.46875f);
        const float fy0 = ceilf((mny + 1.f) * 0.5f * H - 0.53125f), fy1 = floorf((mxy + 1.f) * 0.5f * H - 0.46875f);
        if (!((fx1 >= 0.f) && (fy1 >= 0.f) && (fx0 <= (float)(W - 1)) && (fy0 <= (float)(H - 1)))) return 0;
        x0 = fx0 < 0.f ? 0 : (int)fx0;
        y0 = fy0 < 0.f ? 0 : (int)fy0;
        x1 = fx1 > (float)(W - 1) ? W - 1 : (int)fx1;
        y1 = fy1 > (float)(H - 1) ? H - 1 : (int)fy1;
    } else if (p0.w <= 0.f && p1.w <= 0.f && p2.w <= 0.f) {
        return 0;
    } else {
        x0 = 0; y0 = 0; x1 = W - 1; y1 = H - 1;  // straddles the eye plane: every pixel is a candidate
    }
    bw = x1 - x0 + 1;
    const int bh = y1 - y0 + 1;
    return (bw > 0 && bh > 0) ? bw * bh : 0;
}
#endif
