// Attribute interpolation on gfx950 -- replaces dr.interpolate(attr, rast, tri) with rast_db=None
// (model/render/render.py:23-24; call sites :182-209).  One thread per pixel: a 16-byte texel read
// (coalesced), three index loads, 3 x C gathered floats from the (L2-resident) vertex array and C floats
// written as part of a contiguous row.  Backward scatters bary*g onto the three vertices with float atomics
// (G lanes per pixel, lane = channel) and emits d/du, d/dv for the rasteriser's backward.
// HBM traffic per pixel: 16 B (rast) + 4C B (out); backward 16 + 4C B in, 16 B out (+ atomics in L2).
#include "a3d_common.h"

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
    if (C <= 4)
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
