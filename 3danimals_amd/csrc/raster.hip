// Tiled rasteriser on gfx950: one workgroup owns a screen tile whose depth/id buffer lives in LDS.
//
// Replaces dr.DepthPeeler(...).rasterize_next_layer() / dr.rasterize (model/render/render.py:292-294, :351;
// nvdiffrast, third party, goes through OpenGL).  The meshes on this path are marching-tets surfaces with
// tens of thousands of 1-4 pixel triangles per 256x256 image, so the design is triangle-parallel inside a
// pixel tile instead of pixel-parallel over per-tile bins:
//   * workgroup = (image, tile); the tile's 64-bit (depth | triangle id) keys sit in LDS (64x64 px = 32 KiB);
//   * every thread walks the triangle list with stride blockDim: sets up its triangle (3 x 16 B gathers that
//     stay in the XCD's L2: blockIdx is laid out so all tiles of an image share an XCD), clips its pixel box to
//     the tile and depth-tests the covered pixels with ds_min_u64 -- min over (z/w, id) is order independent,
//     so the image is deterministic with no sorting and no global atomics;
//   * triangles whose clipped box is large are detected with a wave ballot and rasterised cooperatively by
//     all 64 lanes;
//   * at the end the tile is resolved: winner's barycentrics recomputed and the float4 texels written as
//     full, coalesced rows (64 px x 16 B = 1 KiB per row).
// HBM traffic per image: 16 B/vertex + 12 B/face (read once per XCD, then L2) + 16 B/pixel written.
// The per-fragment arithmetic mirrors oracle/raster_ref.c operation by operation; this TU is compiled with
// -ffp-contract=off so that edge functions of a shared edge are exact negations (watertight) and ids match
// the oracle bit for bit.
#include "a3d_common.h"

#define RS_SMALL_AREA 24  // clipped boxes up to this many pixels are walked by the owning lane alone
#define RS_EMPTY 0xFFFFFFFFFFFFFFFFull

struct RsFrag {
    float u, v, zw;
    bool hit;
};

__device__ __forceinline__ RsFrag rs_frag(const float4 p0, const float4 p1, const float4 p2, float fx, float fy) {
    RsFrag r;
    r.u = 0.f; r.v = 0.f; r.zw = 0.f; r.hit = false;
    const float q0x = __builtin_fmaf(-fx, p0.w, p0.x), q0y = __builtin_fmaf(-fy, p0.w, p0.y);
    const float q1x = __builtin_fmaf(-fx, p1.w, p1.x), q1y = __builtin_fmaf(-fy, p1.w, p1.y);
    const float q2x = __builtin_fmaf(-fx, p2.w, p2.x), q2y = __builtin_fmaf(-fy, p2.w, p2.y);
    const float a0 = q1x * q2y - q1y * q2x;
    const float a1 = q2x * q0y - q2y * q0x;
    const float a2 = q0x * q1y - q0y * q1x;
    const float s = (a0 + a1) + a2;
    if (!(s != 0.f) || s != s) return r;
    const float sg = s > 0.f ? 1.f : -1.f;
    // edge i is opposite vertex i and joins vertices (i+1, i+2)
    {
        const float e = a0 * sg;
        if (!(e > 0.f)) {
            if (e < 0.f || e != e) return r;
            const float A = (p1.y * p2.w - p1.w * p2.y) * sg, B = (p1.w * p2.x - p1.x * p2.w) * sg;
            if (!(A > 0.f || (A == 0.f && B > 0.f))) return r;
        }
    }
    {
        const float e = a1 * sg;
        if (!(e > 0.f)) {
            if (e < 0.f || e != e) return r;
            const float A = (p2.y * p0.w - p2.w * p0.y) * sg, B = (p2.w * p0.x - p2.x * p0.w) * sg;
            if (!(A > 0.f || (A == 0.f && B > 0.f))) return r;
        }
    }
    {
        const float e = a2 * sg;
        if (!(e > 0.f)) {
            if (e < 0.f || e != e) return r;
            const float A = (p0.y * p1.w - p0.w * p1.y) * sg, B = (p0.w * p1.x - p0.x * p1.w) * sg;
            if (!(A > 0.f || (A == 0.f && B > 0.f))) return r;
        }
    }
    const float zn = (p0.z * a0 + p1.z * a1) + p2.z * a2;
    const float wn = (p0.w * a0 + p1.w * a1) + p2.w * a2;
    if (!(wn * sg > 0.f)) return r;
    const float zw = zn / wn;
    if (!(zw >= -1.f && zw <= 1.f)) return r;
    const float iw = 1.f / s;
    r.u = fminf(fmaxf(a0 * iw, 0.f), 1.f);
    r.v = fminf(fmaxf(a1 * iw, 0.f), 1.f);
    r.zw = zw;
    r.hit = true;
    return r;
}

__device__ __forceinline__ unsigned rs_order(float f) {  // monotone float -> uint
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// conservative pixel box of a triangle with all w > 0, matching oracle/raster_ref.c; false if off screen
__device__ __forceinline__ bool rs_bbox(const float4 p0, const float4 p1, const float4 p2, int W, int H, int& x0, int& y0, int& x1,
                                        int& y1) {
    const float sx0 = p0.x / p0.w, sx1 = p1.x / p1.w, sx2 = p2.x / p2.w;
    const float sy0 = p0.y / p0.w, sy1 = p1.y / p1.w, sy2 = p2.y / p2.w;
    const float mnx = fminf(sx0, fminf(sx1, sx2)), mxx = fmaxf(sx0, fmaxf(sx1, sx2));
    const float mny = fminf(sy0, fminf(sy1, sy2)), mxy = fmaxf(sy0, fmaxf(sy1, sy2));
    const float fx0 = (mnx + 1.f) * 0.5f * W - 1.5f, fx1 = (mxx + 1.f) * 0.5f * W + 0.5f;
    const float fy0 = (mny + 1.f) * 0.5f * H - 1.5f, fy1 = (mxy + 1.f) * 0.5f * H + 0.5f;
    if (!(fx1 >= 0.f) || !(fy1 >= 0.f) || !(fx0 <= (float)W) || !(fy0 <= (float)H)) return false;
    x0 = fx0 < 0.f ? 0 : (int)fx0;
    y0 = fy0 < 0.f ? 0 : (int)fy0;
    x1 = fx1 > (float)(W - 1) ? W - 1 : (int)fx1;
    y1 = fy1 > (float)(H - 1) ? H - 1 : (int)fy1;
    return true;
}

template <int TW, int TH, int NT>
__global__ __launch_bounds__(NT) void rs_fwd_kernel(const float4* __restrict__ clip, int clip_batch, const int* __restrict__ tri,
                                                    int B, int Bpad, int V, int F, int H, int W, int tiles_x,
                                                    float4* __restrict__ rast) {
    __shared__ unsigned long long s_key[TW * TH];
    const int b = blockIdx.x % Bpad;
    if (b >= B) return;  // padding so that image b always lands on XCD b % 8
    const int tile = blockIdx.x / Bpad;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int tx1 = min(tx0 + TW, W) - 1, ty1 = min(ty0 + TH, H) - 1;
    const float4* pb = clip + (clip_batch == 1 ? 0ll : (long long)b * V);
    const float xs = 2.f / (float)W, xo = 1.f / (float)W - 1.f;
    const float ys = 2.f / (float)H, yo = 1.f / (float)H - 1.f;
    for (int i = threadIdx.x; i < TW * TH; i += NT) s_key[i] = RS_EMPTY;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    for (int f0 = 0; f0 < F; f0 += NT) {
        const int f = f0 + threadIdx.x;
        float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
        int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
        if (f < F) {
            const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
            if ((unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V) {
                p0 = pb[i0]; p1 = pb[i1]; p2 = pb[i2];
                bool live = true;
                if (p0.w > 0.f && p1.w > 0.f && p2.w > 0.f) live = rs_bbox(p0, p1, p2, W, H, x0, y0, x1, y1);
                else if (p0.w <= 0.f && p1.w <= 0.f && p2.w <= 0.f) live = false;
                else { x0 = 0; y0 = 0; x1 = W - 1; y1 = H - 1; }  // straddles the eye plane: test the whole tile
                if (live) {
                    x0 = max(x0, tx0); y0 = max(y0, ty0); x1 = min(x1, tx1); y1 = min(y1, ty1);
                } else {
                    x1 = -1; y1 = -1; x0 = 0; y0 = 0;
                }
            }
        }
        const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
        const int area = (bw > 0 && bh > 0) ? bw * bh : 0;
        // ---- small boxes: the owning lane walks its own pixels
        if (area > 0 && area <= RS_SMALL_AREA) {
            for (int py = y0; py <= y1; ++py) {
                const float fy = __builtin_fmaf(ys, (float)py, yo);
                for (int px = x0; px <= x1; ++px) {
                    const float fx = __builtin_fmaf(xs, (float)px, xo);
                    const RsFrag fr = rs_frag(p0, p1, p2, fx, fy);
                    if (fr.hit) {
                        const unsigned long long key = ((unsigned long long)rs_order(fr.zw) << 32) | (unsigned)f;
                        atomicMin(&s_key[(py - ty0) * TW + (px - tx0)], key);
                    }
                }
            }
        }
        // ---- large boxes: all 64 lanes of the wave share one triangle
        unsigned long long big = __ballot(area > RS_SMALL_AREA);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            float4 c0, c1, c2;
            c0.x = __shfl(p0.x, src); c0.y = __shfl(p0.y, src); c0.z = __shfl(p0.z, src); c0.w = __shfl(p0.w, src);
            c1.x = __shfl(p1.x, src); c1.y = __shfl(p1.y, src); c1.z = __shfl(p1.z, src); c1.w = __shfl(p1.w, src);
            c2.x = __shfl(p2.x, src); c2.y = __shfl(p2.y, src); c2.z = __shfl(p2.z, src); c2.w = __shfl(p2.w, src);
            const int cx0 = __shfl(x0, src), cy0 = __shfl(y0, src), cbw = __shfl(bw, src), carea = __shfl(area, src);
            const int cf = f0 + (threadIdx.x & ~63) + src;
            for (int i = lane; i < carea; i += 64) {
                const int py = cy0 + i / cbw, px = cx0 + i % cbw;
                const float fy = __builtin_fmaf(ys, (float)py, yo);
                const float fx = __builtin_fmaf(xs, (float)px, xo);
                const RsFrag fr = rs_frag(c0, c1, c2, fx, fy);
                if (fr.hit) {
                    const unsigned long long key = ((unsigned long long)rs_order(fr.zw) << 32) | (unsigned)cf;
                    atomicMin(&s_key[(py - ty0) * TW + (px - tx0)], key);
                }
            }
        }
    }
    __syncthreads();
    // ---- resolve: recompute the winner's fragment, write whole rows
    float4* out = rast + (long long)b * H * W;
    for (int i = threadIdx.x; i < TW * TH; i += NT) {
        const int py = ty0 + i / TW, px = tx0 + i % TW;
        if (px >= W || py >= H) continue;
        const unsigned long long key = s_key[i];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (key != RS_EMPTY) {
            const int f = (int)(unsigned)(key & 0xFFFFFFFFull);
            const float4 p0 = pb[tri[3 * f]], p1 = pb[tri[3 * f + 1]], p2 = pb[tri[3 * f + 2]];
            const RsFrag fr = rs_frag(p0, p1, p2, __builtin_fmaf(xs, (float)px, xo), __builtin_fmaf(ys, (float)py, yo));
            o = make_float4(fr.u, fr.v, fr.zw, (float)(f + 1));
        }
        out[(long long)py * W + px] = o;
    }
}

// backward of (u,v) w.r.t. clip-space x, y, w of the three vertices; one thread per pixel
__global__ __launch_bounds__(256) void rs_bwd_kernel(const float4* __restrict__ g_rast, const float4* __restrict__ rast,
                                                     const float4* __restrict__ clip, int clip_batch, const int* __restrict__ tri,
                                                     int V, int F, int H, int W, long long npix, float* __restrict__ g_clip) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float4 r = rast[i];
    const int f = (int)r.w - 1;
    if (f < 0 || f >= F) return;
    const float4 g = g_rast[i];
    if (g.x == 0.f && g.y == 0.f) return;
    const int b = (int)(i / ((long long)H * W));
    const int rem = (int)(i - (long long)b * H * W);
    const int py = rem / W, px = rem - py * W;
    const long long vb = clip_batch == 1 ? 0ll : (long long)b * V;
    const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    const float4 p0 = clip[vb + i0], p1 = clip[vb + i1], p2 = clip[vb + i2];
    const float fx = ((float)px + 0.5f) * (2.f / (float)W) - 1.f;
    const float fy = ((float)py + 0.5f) * (2.f / (float)H) - 1.f;
    const float q0x = p0.x - fx * p0.w, q0y = p0.y - fy * p0.w;
    const float q1x = p1.x - fx * p1.w, q1y = p1.y - fy * p1.w;
    const float q2x = p2.x - fx * p2.w, q2y = p2.y - fy * p2.w;
    const float a0 = q1x * q2y - q1y * q2x, a1 = q2x * q0y - q2y * q0x, a2 = q0x * q1y - q0y * q1x;
    const float s = a0 + a1 + a2;
    if (s == 0.f) return;
    const float is = 1.f / s;
    const float u = a0 * is, v = a1 * is;
    const float t = g.x * u + g.y * v;
    const float ga0 = (g.x - t) * is, ga1 = (g.y - t) * is, ga2 = -t * is;
    const float g0x = -ga1 * q2y + ga2 * q1y, g0y = ga1 * q2x - ga2 * q1x;
    const float g1x = ga0 * q2y - ga2 * q0y, g1y = -ga0 * q2x + ga2 * q0x;
    const float g2x = -ga0 * q1y + ga1 * q0y, g2y = ga0 * q1x - ga1 * q0x;
    float* o0 = g_clip + (vb + i0) * 4;
    float* o1 = g_clip + (vb + i1) * 4;
    float* o2 = g_clip + (vb + i2) * 4;
    atomicAdd(o0, g0x); atomicAdd(o0 + 1, g0y); atomicAdd(o0 + 3, -fx * g0x - fy * g0y);
    atomicAdd(o1, g1x); atomicAdd(o1 + 1, g1y); atomicAdd(o1 + 3, -fx * g1x - fy * g1y);
    atomicAdd(o2, g2x); atomicAdd(o2 + 1, g2y); atomicAdd(o2 + 3, -fx * g2x - fy * g2y);
}

extern "C" int a3d_rast_fwd(const float* clip, int clip_batch, const int32_t* tri, int B, int V, int F, int H, int W, float* rast,
                            a3d_stream_t stream) {
    A3D_CHECK_ARG(clip && rast && B > 0 && V > 0 && F >= 0 && H > 0 && W > 0);
    A3D_CHECK_ARG(F == 0 || tri);
    A3D_CHECK_ARG(clip_batch == 1 || clip_batch == B);
    hipStream_t s = (hipStream_t)stream;
    if (F == 0) {
        A3D_HIP(hipMemsetAsync(rast, 0, sizeof(float) * 4 * (size_t)B * H * W, s));
        return A3D_OK;
    }
    const int Bpad = (B + 7) & ~7;
    auto tiles = [&](int t) { return a3d_div_up(W, t) * a3d_div_up(H, t); };
    // biggest tile that still gives every CU a workgroup (256 CUs)
    if ((long long)B * tiles(64) >= 256) {
        hipLaunchKernelGGL((rs_fwd_kernel<64, 64, 1024>), dim3(tiles(64) * Bpad), dim3(1024), 0, s, (const float4*)clip, clip_batch, tri, B,
                           Bpad, V, F, H, W, a3d_div_up(W, 64), (float4*)rast);
    } else if ((long long)B * tiles(32) >= 256) {
        hipLaunchKernelGGL((rs_fwd_kernel<32, 32, 256>), dim3(tiles(32) * Bpad), dim3(256), 0, s, (const float4*)clip, clip_batch, tri, B,
                           Bpad, V, F, H, W, a3d_div_up(W, 32), (float4*)rast);
    } else {
        hipLaunchKernelGGL((rs_fwd_kernel<16, 16, 256>), dim3(tiles(16) * Bpad), dim3(256), 0, s, (const float4*)clip, clip_batch, tri, B,
                           Bpad, V, F, H, W, a3d_div_up(W, 16), (float4*)rast);
    }
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_rast_bwd(const float* g_rast, const float* rast, const float* clip, int clip_batch, const int32_t* tri, int B, int V,
                            int F, int H, int W, float* g_clip, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_rast && rast && clip && g_clip && B > 0 && V > 0 && F >= 0 && H > 0 && W > 0);
    A3D_CHECK_ARG(clip_batch == 1 || clip_batch == B);
    hipStream_t s = (hipStream_t)stream;
    A3D_HIP(hipMemsetAsync(g_clip, 0, sizeof(float) * 4 * (size_t)clip_batch * V, s));
    if (F == 0) return A3D_OK;
    const long long npix = (long long)B * H * W;
    hipLaunchKernelGGL(rs_bwd_kernel, dim3(a3d_div_up(npix, 256)), dim3(256), 0, s, (const float4*)g_rast, (const float4*)rast,
                       (const float4*)clip, clip_batch, tri, V, F, H, W, npix, g_clip);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
