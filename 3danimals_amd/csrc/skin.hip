// Linear-blend skinning on gfx950: one thread per (image, vertex).
//
// Replaces the per-vertex part of skinning() (model/geometry/skinning.py:377, :419-431): softmax-over-bones
// weights from point-to-segment distances (skinning.py:16-22, geometry/util.py:30-53) and the weighted sum
// of the K per-bone affine maps.  The reference materialises K copies of the [B,V,3] vertex array; here a
// vertex is read once (12 B), the image's bones (K x 7 floats) and transforms (K x 12 floats) sit in LDS,
// and the softmax runs online in registers.  Weights are recomputed in backward instead of stored.
// HBM traffic: 12 B/vertex in (shared prior: once per image from L2), 12 B/vertex out.
#include "a3d_common.h"

#define SK_THREADS 256
#define SK_MAXK 64
#define SK_VPT 4  // vertices per thread in the g_T reduction

struct SkBone {
    float ax, ay, az, dx, dy, dz, inv_len2;
};

__device__ __forceinline__ void sk_stage(const float* __restrict__ bones, const float* __restrict__ T, int K, SkBone* s_bone,
                                         float* s_T) {
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float* b = bones + 6 * k;
        SkBone sb;
        sb.ax = b[0]; sb.ay = b[1]; sb.az = b[2];
        sb.dx = b[3] - b[0]; sb.dy = b[4] - b[1]; sb.dz = b[5] - b[2];
        float l2 = sb.dx * sb.dx + sb.dy * sb.dy + sb.dz * sb.dz;
        sb.inv_len2 = 1.f / fmaxf(l2, 1e-6f);  // geometry/util.py:41
        s_bone[k] = sb;
    }
    for (int i = threadIdx.x; i < K * 12; i += blockDim.x) s_T[i] = T[i];
}

// -dist(p, segment_k) / temperature   (geometry/util.py:41-52, skinning.py:21)
__device__ __forceinline__ float sk_logit(const SkBone& b, float px, float py, float pz, float neg_inv_temp) {
    float rx = px - b.ax, ry = py - b.ay, rz = pz - b.az;
    float t = (rx * b.dx + ry * b.dy + rz * b.dz) * b.inv_len2;
    t = fminf(fmaxf(t, 0.f), 1.f);
    float sx = t * b.dx - rx, sy = t * b.dy - ry, sz = t * b.dz - rz;
    return sqrtf(sx * sx + sy * sy + sz * sz + 1e-6f) * neg_inv_temp;
}

__global__ __launch_bounds__(SK_THREADS) void sk_fwd_kernel(const float* __restrict__ v, int v_batch, const float* __restrict__ bones,
                                                            int bones_batch, const float* __restrict__ T, int V, int K,
                                                            float neg_inv_temp, float* __restrict__ out, float* __restrict__ weights) {
    __shared__ SkBone s_bone[SK_MAXK];
    __shared__ float s_T[SK_MAXK * 12];
    const int b = blockIdx.y;
    sk_stage(bones + (bones_batch == 1 ? 0ll : (long long)b * K * 6), T + (long long)b * K * 12, K, s_bone, s_T);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const float* p = v + ((v_batch == 1 ? 0ll : (long long)b * V) + i) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, sk_logit(s_bone[k], px, py, pz, neg_inv_temp));
    float s = 0.f, ox = 0.f, oy = 0.f, oz = 0.f;
    for (int k = 0; k < K; ++k) {
        float e = __expf(sk_logit(s_bone[k], px, py, pz, neg_inv_temp) - m);
        const float* t = s_T + 12 * k;
        s += e;
        ox += e * (t[0] * px + t[1] * py + t[2] * pz + t[3]);
        oy += e * (t[4] * px + t[5] * py + t[6] * pz + t[7]);
        oz += e * (t[8] * px + t[9] * py + t[10] * pz + t[11]);
    }
    const float inv = 1.f / s;
    float* o = out + ((long long)b * V + i) * 3;
    o[0] = ox * inv; o[1] = oy * inv; o[2] = oz * inv;
    if (weights) {  // [K, Bw, V]; only images that own distinct weights write
        const int Bw = (v_batch == 1 && bones_batch == 1) ? 1 : (int)gridDim.y;
        if (b < Bw)
            for (int k = 0; k < K; ++k)
                weights[((long long)k * Bw + b) * V + i] = __expf(sk_logit(s_bone[k], px, py, pz, neg_inv_temp) - m) * inv;
    }
}

// backward: g_v (through the affine maps only) and the per-bone 3x4 gradient reduced over the vertices
__global__ __launch_bounds__(SK_THREADS) void sk_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ v, int v_batch,
                                                            const float* __restrict__ bones, int bones_batch,
                                                            const float* __restrict__ T, int V, int K, float neg_inv_temp,
                                                            float* __restrict__ g_v, float* __restrict__ g_T) {
    __shared__ SkBone s_bone[SK_MAXK];
    __shared__ float s_T[SK_MAXK * 12];
    __shared__ float s_gT[SK_MAXK * 12];
    const int b = blockIdx.y;
    sk_stage(bones + (bones_batch == 1 ? 0ll : (long long)b * K * 6), T + (long long)b * K * 12, K, s_bone, s_T);
    for (int i = threadIdx.x; i < K * 12; i += blockDim.x) s_gT[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float px[SK_VPT], py[SK_VPT], pz[SK_VPT], gx[SK_VPT], gy[SK_VPT], gz[SK_VPT], mx[SK_VPT], is[SK_VPT];
    const int base = (blockIdx.x * blockDim.x) * SK_VPT + threadIdx.x;
#pragma unroll
    for (int j = 0; j < SK_VPT; ++j) {
        const int i = base + j * SK_THREADS;
        const bool ok = i < V;
        const float* p = v + ((v_batch == 1 ? 0ll : (long long)b * V) + (ok ? i : 0)) * 3;
        const float* g = g_out + ((long long)b * V + (ok ? i : 0)) * 3;
        px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
        gx[j] = ok ? g[0] : 0.f; gy[j] = ok ? g[1] : 0.f; gz[j] = ok ? g[2] : 0.f;
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) m = fmaxf(m, sk_logit(s_bone[k], px[j], py[j], pz[j], neg_inv_temp));
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += __expf(sk_logit(s_bone[k], px[j], py[j], pz[j], neg_inv_temp) - m);
        mx[j] = m;
        is[j] = 1.f / s;
    }
    float dvx[SK_VPT] = {0}, dvy[SK_VPT] = {0}, dvz[SK_VPT] = {0};
    for (int k = 0; k < K; ++k) {
        const float* t = s_T + 12 * k;
        float a[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) a[q] = 0.f;
#pragma unroll
        for (int j = 0; j < SK_VPT; ++j) {
            float w = __expf(sk_logit(s_bone[k], px[j], py[j], pz[j], neg_inv_temp) - mx[j]) * is[j];
            float wx = w * gx[j], wy = w * gy[j], wz = w * gz[j];
            a[0] += wx * px[j]; a[1] += wx * py[j]; a[2] += wx * pz[j]; a[3] += wx;
            a[4] += wy * px[j]; a[5] += wy * py[j]; a[6] += wy * pz[j]; a[7] += wy;
            a[8] += wz * px[j]; a[9] += wz * py[j]; a[10] += wz * pz[j]; a[11] += wz;
            // R^T (w g)
            dvx[j] += t[0] * wx + t[4] * wy + t[8] * wz;
            dvy[j] += t[1] * wx + t[5] * wy + t[9] * wz;
            dvz[j] += t[2] * wx + t[6] * wy + t[10] * wz;
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            float r = a3d_wave_sum(a[q]);
            if (lane == 0) atomicAdd(&s_gT[12 * k + q], r);  // 4 waves per block -> LDS
        }
    }
    if (g_v) {
#pragma unroll
        for (int j = 0; j < SK_VPT; ++j) {
            const int i = base + j * SK_THREADS;
            if (i >= V) continue;
            if (v_batch == 1) {
                float* o = g_v + 3ll * i;
                atomicAdd(o, dvx[j]); atomicAdd(o + 1, dvy[j]); atomicAdd(o + 2, dvz[j]);
            } else {
                float* o = g_v + ((long long)b * V + i) * 3;
                o[0] = dvx[j]; o[1] = dvy[j]; o[2] = dvz[j];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * 12; i += blockDim.x) atomicAdd(g_T + (long long)b * K * 12 + i, s_gT[i]);
}

extern "C" int a3d_skin_fwd(const float* v, int v_batch, const float* bones, int bones_batch, const float* T, int B, int V, int K,
                            float temperature, float* out, float* weights_or_null, a3d_stream_t stream) {
    A3D_CHECK_ARG(v && bones && T && out);
    A3D_CHECK_ARG(B > 0 && V > 0 && K > 0 && K <= SK_MAXK && temperature > 0.f);
    A3D_CHECK_ARG((v_batch == 1 || v_batch == B) && (bones_batch == 1 || bones_batch == B));
    hipLaunchKernelGGL(sk_fwd_kernel, dim3(a3d_div_up(V, SK_THREADS), B), dim3(SK_THREADS), 0, (hipStream_t)stream, v, v_batch, bones,
                       bones_batch, T, V, K, -1.f / temperature, out, weights_or_null);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_skin_bwd(const float* g_out, const float* v, int v_batch, const float* bones, int bones_batch, const float* T, int B,
                            int V, int K, float temperature, float* g_v_or_null, float* g_T, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_out && v && bones && T && g_T);
    A3D_CHECK_ARG(B > 0 && V > 0 && K > 0 && K <= SK_MAXK && temperature > 0.f);
    A3D_CHECK_ARG((v_batch == 1 || v_batch == B) && (bones_batch == 1 || bones_batch == B));
    hipStream_t s = (hipStream_t)stream;
    A3D_HIP(hipMemsetAsync(g_T, 0, sizeof(float) * (size_t)B * K * 12, s));
    if (g_v_or_null && v_batch == 1) A3D_HIP(hipMemsetAsync(g_v_or_null, 0, sizeof(float) * 3 * (size_t)V, s));
    hipLaunchKernelGGL(sk_bwd_kernel, dim3(a3d_div_up(V, SK_THREADS * SK_VPT), B), dim3(SK_THREADS), 0, s, g_out, v, v_batch, bones,
                       bones_batch, T, V, K, -1.f / temperature, g_v_or_null, g_T);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
