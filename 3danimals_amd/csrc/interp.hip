// Attribute interpolation on gfx950 -- replaces dr.interpolate(attr, rast, tri) with rast_db=None
// (model/render/render.py:23-24; call sites :182-209).  One thread per pixel: a 16-byte texel read
// (coalesced), three index loads, 3 x C gathered floats from the (L2-resident) vertex array and C floats
// written as part of a contiguous row.  Backward scatters bary*g onto the three vertices with float atomics
// (G lanes per pixel, lane = channel) and emits d/du, d/dv for the rasteriser's backward.
// HBM traffic per pixel: 16 B (rast) + 4C B (out); backward 16 + 4C B in, 16 B out (+ atomics in L2).
#include "a3d_common.h"
#include "tile_scatter.h"

#define IP_MAXC 64

__global__ __launch_bounds__(256) void ip_fwd_kernel(const float* __restrict__ attr, int attr_batch, int C, const float4* __restrict__ rast,
                                                     const int* __restrict__ tri, int V, int F, long long hw, long long npix,
                                                     float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float4 r = rast[i];
    const int f = (int)r.w - 1;
    float* o = out + i * C;
    if (f < 0 || f >= F) {
        for (int c = 0; c < C; ++c) o[c] = 0.f;
        return;
    }
    const long long vb = attr_batch == 1 ? 0ll : (i / hw) * V;
    const float* a0 = attr + (vb + tri[3 * f]) * C;
    const float* a1 = attr + (vb + tri[3 * f + 1]) * C;
    const float* a2 = attr + (vb + tri[3 * f + 2]) * C;
    const float u = r.x, v = r.y, w = 1.f - u - v;
    for (int c = 0; c < C; ++c) o[c] = u * a0[c] + v * a1[c] + w * a2[c];
}

// G lanes per pixel (4, 8 or 16; lane = channel, channels beyond G in further rounds): the incoming gradient row and the three
// attribute rows are read as contiguous runs, the bary-weighted adds of one (pixel, vertex) pair are G adjacent floats = one request to
// the L2's atomic unit (line-coalesced atomics), and d/du, d/dv are summed over the group with shuffles.
// (One thread per pixel walked the channels serially and issued 3C single-lane atomics: 128 us for the 16-channel case at B = 16.)
template <int G>
__global__ __launch_bounds__(256) void ip_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ attr, int attr_batch, int C,
                                                     const float4* __restrict__ rast, const int* __restrict__ tri, int V, int F,
                                                     long long hw, long long npix, float* __restrict__ g_attr,
                                                     float4* __restrict__ g_rast) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long i = t / G;
    const int sub = (int)(t - i * G);
    if (i >= npix) return;  // (whole groups leave together: blockDim is a multiple of G)
    const float4 r = rast[i];
    const int f = (int)r.w - 1;
    float gu = 0.f, gv = 0.f;
    if (f >= 0 && f < F) {
        const long long vb = attr_batch == 1 ? 0ll : (i / hw) * V;
        const long long o0 = (vb + tri[3 * f]) * C, o1 = (vb + tri[3 * f + 1]) * C, o2 = (vb + tri[3 * f + 2]) * C;
        const float u = r.x, v = r.y, w = 1.f - u - v;
        const float* g = g_out + i * C;
        for (int c = sub; c < C; c += G) {
            const float gc = g[c];
            const float a2 = attr[o2 + c];
            gu += gc * (attr[o0 + c] - a2);
            gv += gc * (attr[o1 + c] - a2);
            if (g_attr && gc != 0.f) {
                atomicAdd(g_attr + o0 + c, u * gc);
                atomicAdd(g_attr + o1 + c, v * gc);
                atomicAdd(g_attr + o2 + c, w * gc);
            }
        }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        gu += __shfl_xor(gu, o, 64);
        gv += __shfl_xor(gv, o, 64);
    }
    if (sub == 0) g_rast[i] = make_float4(gu, gv, 0.f, 0.f);
}

// Round 6, C <= 16: one thread per pixel of a 16 x 16 tile; pixels on the same triangle merge their three bary-weighted gradient rows
// inside the wave (ts_merge), the survivors meet their neighbours' in the work-group's LDS table (tile_scatter.h) and leave as one row
// of adjacent atomics per vertex and tile.  Kernel us at B = 16, 256 x 256 (rocprofv3, tools/shim_bwd_prof.sh; per-pixel form ->
// this one): 3 channels 50.7 -> 27.6 on the fresh mesh, 115 -> 30 on the trained-like one (its long thin triangles make the per-pixel
// form's atomics collide); 16 channels 137 -> 92.  105 VGPRs (4 waves per SIMD): forcing 6 spills and loses (33 us).
// CM = the channel count rounded up (register rows), ROUNDS = merge rounds.
template <int CM, int ROUNDS>
__global__ __launch_bounds__(256) void ip_bwd_tile_kernel(const float* __restrict__ g_out, const float* __restrict__ attr, int attr_batch, int C,
                                                          const float4* __restrict__ rast, const int* __restrict__ tri, int V, int F, int H, int W,
                                                          int tiles_x, float* __restrict__ g_attr, float4* __restrict__ g_rast) {
    extern __shared__ __align__(16) unsigned char ip_bwd_lds[];
    A3D_STAMP(0, 0);
    const int b = blockIdx.y, tile = blockIdx.x;
    int px, py;
    ts_pixel((tile % tiles_x) * TS_TILE, (tile / tiles_x) * TS_TILE, px, py);
    const bool inside = px < W && py < H;
    const long long i = ((long long)b * H + py) * W + px;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inside) r = rast[i];
    const int f = (int)r.w - 1;
    const bool live = inside && f >= 0 && f < F;
    if (!__syncthreads_or(live)) {
        if (inside) g_rast[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    TileScatter ts;
    if (g_attr) ts.init(ip_bwd_lds, C);
    float gu = 0.f, gv = 0.f;
    float c[3 * CM];
#pragma unroll
    for (int n = 0; n < 3 * CM; ++n) c[n] = 0.f;
    int row[3] = {0, 0, 0}, key = -1;
    if (live) {
        const int vb = attr_batch == 1 ? 0 : b * V;
        row[0] = vb + tri[3 * f]; row[1] = vb + tri[3 * f + 1]; row[2] = vb + tri[3 * f + 2];
        const float* a0 = attr + (long long)row[0] * C;
        const float* a1 = attr + (long long)row[1] * C;
        const float* a2 = attr + (long long)row[2] * C;
        const float w2 = 1.f - r.x - r.y;
        const float* g = g_out + i * C;
#pragma unroll
        for (int ch = 0; ch < CM; ++ch) {
            if (ch < C) {
                const float gc = g[ch], x2 = a2[ch];
                gu += gc * (a0[ch] - x2);
                gv += gc * (a1[ch] - x2);
                c[ch] = r.x * gc; c[CM + ch] = r.y * gc; c[2 * CM + ch] = w2 * gc;
            }
        }
        key = f;
    }
    if (inside) g_rast[i] = make_float4(gu, gv, 0.f, 0.f);
    if (!g_attr) return;
    A3D_STAMP(0, 1);
    ts_merge<3 * CM, ROUNDS>(key, c);
    A3D_STAMP(0, 2);
    __syncthreads();  // (table initialised)
    const int e0 = ts.entries(key >= 0, 3);
    if (key >= 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int sl = ts.slot(row[k]);
#pragma unroll
            for (int ch = 0; ch < CM; ++ch) {
                if (ch < C) {
                    const float v = c[k * CM + ch];
                    if (sl >= 0) ts.e_val[(e0 + k) * C + ch] = v;
                    else if (v != 0.f) atomicAdd(g_attr + (long long)row[k] * C + ch, v);
                }
            }
            if (sl >= 0) ts.link(e0 + k, sl);
        }
    }
    A3D_STAMP(0, 3);
    __syncthreads();
    A3D_STAMP(0, 4);
    ts.flush<CM>(g_attr, C, -1);
    A3D_STAMP(0, 5);
}

extern "C" int a3d_interp_fwd(const float* attr, int attr_batch, int C, const float* rast, const int32_t* tri, int B, int V, int F, int H,
                              int W, float* out, a3d_stream_t stream) {
    A3D_CHECK_ARG(attr && rast && out && B > 0 && V > 0 && F >= 0 && H > 0 && W > 0 && C > 0 && C <= IP_MAXC);
    A3D_CHECK_ARG(F == 0 || tri);
    A3D_CHECK_ARG(attr_batch == 1 || attr_batch == B);
    const long long hw = (long long)H * W, npix = hw * B;
    hipLaunchKernelGGL(ip_fwd_kernel, dim3(a3d_div_up(npix, 256)), dim3(256), 0, (hipStream_t)stream, attr, attr_batch, C,
                       (const float4*)rast, tri, V, F, hw, npix, out);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_interp_bwd(const float* g_out, const float* attr, int attr_batch, int C, const float* rast, const int32_t* tri, int B,
                              int V, int F, int H, int W, float* g_attr_or_null, float* g_rast, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_out && attr && rast && g_rast && B > 0 && V > 0 && F >= 0 && H > 0 && W > 0 && C > 0 && C <= IP_MAXC);
    A3D_CHECK_ARG(F == 0 || tri);
    A3D_CHECK_ARG(attr_batch == 1 || attr_batch == B);
    hipStream_t s = (hipStream_t)stream;
    if (g_attr_or_null) A3D_HIP(hipMemsetAsync(g_attr_or_null, 0, sizeof(float) * (size_t)attr_batch * V * C, s));
    const long long hw = (long long)H * W, npix = hw * B;
    if (C <= 16 && B <= 65535 && (long long)attr_batch * V < 0x7fffffffll && a3d_exp() != 140) {
        const int tiles_x = a3d_div_up(W, TS_TILE), tiles_y = a3d_div_up(H, TS_TILE);
#define IP_TILE(CM_, R_)                                                                                                                      \
    hipLaunchKernelGGL((ip_bwd_tile_kernel<CM_, R_>), dim3(tiles_x * tiles_y, B), dim3(256), g_attr_or_null ? TileScatter::lds_bytes(C) : 0, s, g_out, \
                       attr, attr_batch, C, (const float4*)rast, tri, V, F, H, W, tiles_x, g_attr_or_null, (float4*)g_rast)
        if (C <= 4) IP_TILE(4, 6); else if (C <= 8) IP_TILE(8, 4); else IP_TILE(16, 4);
#undef IP_TILE
    } else if (C <= 4)
        hipLaunchKernelGGL(ip_bwd_kernel<4>, dim3(a3d_div_up(4 * npix, 256)), dim3(256), 0, s, g_out, attr, attr_batch, C, (const float4*)rast, tri,
                           V, F, hw, npix, g_attr_or_null, (float4*)g_rast);
    else if (C <= 8)
        hipLaunchKernelGGL(ip_bwd_kernel<8>, dim3(a3d_div_up(8 * npix, 256)), dim3(256), 0, s, g_out, attr, attr_batch, C, (const float4*)rast, tri,
                           V, F, hw, npix, g_attr_or_null, (float4*)g_rast);
    else
        hipLaunchKernelGGL(ip_bwd_kernel<16>, dim3(a3d_div_up(16 * npix, 256)), dim3(256), 0, s, g_out, attr, attr_batch, C, (const float4*)rast,
                           tri, V, F, hw, npix, g_attr_or_null, (float4*)g_rast);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

A3D_PROFILE_TU(interp)
