// Clip-space transform of the posed vertices on gfx950 -- xfm_points(points, matrix, use_python=True),
// /root/reference/model/render/renderutils/ops.py:515-531: out = matmul(pad(points, (0, 1), value=1), matrix^T), the first thing
// render_mesh does (render.py:279).  As torch ops that is a pad, a batched [V,4] x [4,4] GEMM and -- backward -- two more GEMMs of that
// shape and the pad's slice: six launches of a GEMM library whose smallest tile is larger than the problem (27 us for one of them on the
// bench workload, tools/glue_attribution.py).  Here: one thread per (image, vertex) each way; the backward reduces the 16 entries of
// d/d matrix per work-group (wave sums -> LDS -> 16 atomics per work-group and image).
//   fwd : out[b,v,i] = M[b,i,0] x + M[b,i,1] y + M[b,i,2] z + M[b,i,3]
//   bwd : g_points[b,v,j] = sum_i g[b,v,i] M[b,i,j]  (j < 3);   g_M[b,i,j] += sum_v g[b,v,i] [x,y,z,1]_j
#include "a3d_common.h"

namespace {

__global__ __launch_bounds__(256) void xf_fwd_kernel(const float* __restrict__ points, int points_batch, const float* __restrict__ M, int m_batch, int V,
                                                     float4* __restrict__ out, float* __restrict__ clear, int n_clear) {
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < n_clear; i += gridDim.x * gridDim.y * 256) clear[i] = 0.f;
    if (v >= V) return;
    const float* p = points + ((points_batch == 1 ? 0ll : (long long)b * V) + v) * 3;
    const float* m = M + (m_batch == 1 ? 0 : 16 * b);
    const float x = p[0], y = p[1], z = p[2];
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = __builtin_fmaf(m[4 * i], x, __builtin_fmaf(m[4 * i + 1], y, __builtin_fmaf(m[4 * i + 2], z, m[4 * i + 3])));
    out[(long long)b * V + v] = make_float4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void xf_bwd_kernel(const float* __restrict__ g, int g_stride, const float* __restrict__ points, int points_batch,
                                                     const float* __restrict__ M, int m_batch, int V, float* __restrict__ g_points,
                                                     float* __restrict__ g_M, const float* __restrict__ addend, int addend_stride,
                                                     const float* __restrict__ addend2, int addend2_stride, const float* __restrict__ g2, int g2_stride) {
    __shared__ float s_red[16][17];
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    const bool live = v < V;
    float gm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) gm[k] = 0.f;
    if (live) {
        const float* gp = g + ((long long)b * V + v) * g_stride;
        float gi[4] = {gp[0], gp[1], gp[2], gp[3]};
        if (g2) {  // (404) the clip positions' second gradient -- the antialiasing's, render.py:264-268 -- summed here, not by the engine
            const float* q = g2 + ((long long)b * V + v) * g2_stride;
#pragma unroll
            for (int i = 0; i < 4; ++i) gi[i] = q[i] + gi[i];
        }
        const float* m = M + (m_batch == 1 ? 0 : 16 * b);
        if (g_points) {
            float* o = g_points + ((long long)b * V + v) * 3;
#pragma unroll
            for (int j = 0; j < 3; ++j) o[j] = gi[0] * m[j] + gi[1] * m[4 + j] + gi[2] * m[8 + j] + gi[3] * m[12 + j];
            if (addend) {  // a second gradient of the same points (their other consumer's), summed on the way out: no accumulation launch
                const float* a = addend + ((long long)b * V + v) * addend_stride;
#pragma unroll
                for (int j = 0; j < 3; ++j) o[j] += a[j];
            }
            if (addend2) {  // ... and a third (404: the one the vertex normals' backward produced for the same points)
                const float* a = addend2 + ((long long)b * V + v) * addend2_stride;
#pragma unroll
                for (int j = 0; j < 3; ++j) o[j] += a[j];
            }
        }
        if (g_M) {
            const float* p = points + ((points_batch == 1 ? 0ll : (long long)b * V) + v) * 3;
            const float h[4] = {p[0], p[1], p[2], 1.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) gm[4 * i + j] = gi[i] * h[j];
        }
    }
    if (!g_M) return;
    // 16 sums over the work-group: 16-lane DPP rows first (four register moves per value, no LDS traffic), the 16 rows meet in LDS
    // (a full-wave butterfly through ds_bpermute -- 16 values x 6 steps -- was most of this kernel's 5.6 us)
    const int lane = threadIdx.x & 63, row = threadIdx.x >> 4;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float r = gm[k];
        r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
        r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x141, 0xF, 0xF, true));  // row_half_mirror
        r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x140, 0xF, 0xF, true));  // row_mirror
        if ((lane & 15) == 0) s_red[row][k] = r;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += s_red[r][threadIdx.x];
        if (t != 0.f) atomicAdd(g_M + (m_batch == 1 ? 0 : 16 * b) + threadIdx.x, t);
    }
}

// ---- per-vertex 2-D motion to the next frame (render.py:281-288): ndc = clip.xy / clip.w per frame, delta[b,f] = ndc[b,f+1] - ndc[b,f],
// zeros for a sequence's last frame.  As torch ops: a slice, a division, a view, two slices, a subtraction, a zeros_like and a cat forward and
// eleven slice / division / subtraction adjoints backward -- ~25 launches of 4-5 us for 2.4 MB (tools/glue_attribution.py, the Ponymation
// step: ~100 us).  Here one thread per (frame, vertex) each way.
__global__ __launch_bounds__(256) void xf_flow_fwd_kernel(const float4* __restrict__ clip, int N, int F, int V, float2* __restrict__ delta) {
    const int n = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const int f = n % F;
    float2 d = make_float2(0.f, 0.f);
    if (f + 1 < F) {
        const float4 a = clip[(long long)n * V + v], b = clip[(long long)(n + 1) * V + v];
        d.x = b.x / b.w - a.x / a.w;
        d.y = b.y / b.w - a.y / a.w;
    }
    delta[(long long)n * V + v] = d;
}

// g_ndc[n] = g_delta[n-1] (not for a sequence's first frame) - g_delta[n] (not for its last);  ndc = xy / w:  g_xy = g_ndc / w,
// g_w = -(g_ndc . xy / w) / w  (torch's div backward: -grad * (a / b) / b), g_z = 0
// (g_delta with a vertex stride of g_stride floats: the gradient may be two columns of the G-buffer backward's 16-float rows, read in place)
__global__ __launch_bounds__(256) void xf_flow_bwd_kernel(const float* __restrict__ g_delta, int g_stride, const float4* __restrict__ clip, int N, int F,
                                                          int V, float4* __restrict__ g_clip) {
    const int n = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const int f = n % F;
    const long long i = (long long)n * V + v;
    float gx = 0.f, gy = 0.f;
    if (f > 0) { const float* g = g_delta + (i - V) * g_stride; gx = g[0]; gy = g[1]; }
    if (f + 1 < F) { const float* g = g_delta + i * g_stride; gx -= g[0]; gy -= g[1]; }
    const float4 c = clip[i];
    float4 o;
    o.x = gx / c.w;
    o.y = gy / c.w;
    o.z = 0.f;
    o.w = -(gx * (c.x / c.w)) / c.w + -(gy * (c.y / c.w)) / c.w;
    g_clip[i] = o;
}

}  // namespace

extern "C" int a3d_xfm_points_fwd(const float* points, int points_batch, const float* matrix, int matrix_batch, int B, int V, float* out,
                                  float* g_matrix_to_clear_or_null, a3d_stream_t stream) {
    A3D_CHECK_ARG(B > 0 && V >= 0 && B <= 65535 && (points_batch == 1 || points_batch == B) && (matrix_batch == 1 || matrix_batch == B));
    hipStream_t s = (hipStream_t)stream;
    if (V == 0) {
        if (g_matrix_to_clear_or_null) A3D_HIP(hipMemsetAsync(g_matrix_to_clear_or_null, 0, sizeof(float) * 16 * (size_t)matrix_batch, s));
        return A3D_OK;
    }
    A3D_CHECK_ARG(points && matrix && out && ((uintptr_t)out & 15) == 0);
    hipLaunchKernelGGL(xf_fwd_kernel, dim3(a3d_div_up(V, 256), B), dim3(256), 0, s, points, points_batch, matrix, matrix_batch, V, (float4*)out,
                       g_matrix_to_clear_or_null, g_matrix_to_clear_or_null ? 16 * matrix_batch : 0);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_xfm_points_bwd(const float* g_out, int g_stride, const float* points, int points_batch, const float* matrix, int matrix_batch,
                                  int B, int V, float* g_points_or_null, float* g_matrix_or_null, int g_matrix_is_clear,
                                  const float* g_points_addend_or_null, int addend_stride, const float* g_points_addend2_or_null,
                                  int addend2_stride, const float* g_out2_or_null, int g2_stride, a3d_stream_t stream) {
    A3D_CHECK_ARG(!g_out2_or_null || g2_stride >= 4);
    A3D_CHECK_ARG(!g_points_addend_or_null || (addend_stride >= 3 && g_points_or_null && points_batch == B));
    A3D_CHECK_ARG(!g_points_addend2_or_null || (addend2_stride >= 3 && g_points_or_null && points_batch == B));
    A3D_CHECK_ARG(B > 0 && V >= 0 && B <= 65535 && g_stride >= 4 && (points_batch == 1 || points_batch == B) && (matrix_batch == 1 || matrix_batch == B));
    hipStream_t s = (hipStream_t)stream;
    if (g_matrix_or_null && !g_matrix_is_clear) A3D_HIP(hipMemsetAsync(g_matrix_or_null, 0, sizeof(float) * 16 * (size_t)matrix_batch, s));
    if (V == 0 || (!g_points_or_null && !g_matrix_or_null)) return A3D_OK;
    A3D_CHECK_ARG(g_out && points && matrix);
    hipLaunchKernelGGL(xf_bwd_kernel, dim3(a3d_div_up(V, 256), B), dim3(256), 0, s, g_out, g_stride, points, points_batch, matrix, matrix_batch, V,
                       g_points_or_null, g_matrix_or_null, g_points_addend_or_null, addend_stride, g_points_addend2_or_null, addend2_stride, g_out2_or_null, g2_stride);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_flow_delta_fwd(const float* clip, int N, int F, int V, float* delta, a3d_stream_t stream) {
    A3D_CHECK_ARG(N > 0 && F > 0 && N % F == 0 && N <= 65535 && V >= 0);
    if (V == 0) return A3D_OK;
    A3D_CHECK_ARG(clip && delta && ((uintptr_t)clip & 15) == 0 && ((uintptr_t)delta & 7) == 0);
    hipLaunchKernelGGL(xf_flow_fwd_kernel, dim3(a3d_div_up(V, 256), N), dim3(256), 0, (hipStream_t)stream, (const float4*)clip, N, F, V, (float2*)delta);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_flow_delta_bwd(const float* g_delta, int g_stride, const float* clip, int N, int F, int V, float* g_clip, a3d_stream_t stream) {
    A3D_CHECK_ARG(N > 0 && F > 0 && N % F == 0 && N <= 65535 && V >= 0 && g_stride >= 2);
    if (V == 0) return A3D_OK;
    A3D_CHECK_ARG(g_delta && clip && g_clip && ((uintptr_t)clip & 15) == 0 && ((uintptr_t)g_clip & 15) == 0);
    hipLaunchKernelGGL(xf_flow_bwd_kernel, dim3(a3d_div_up(V, 256), N), dim3(256), 0, (hipStream_t)stream, g_delta, g_stride, (const float4*)clip, N, F, V,
                       (float4*)g_clip);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

A3D_PROFILE_TU(xfm)
