// Linear-blend skinning on gfx950: one thread per (image, vertex).
//
// Replaces the per-vertex part of skinning() (model/geometry/skinning.py:377, :419-431): softmax-over-bones
// weights from point-to-segment distances (skinning.py:16-22, geometry/util.py:30-53) and the weighted sum
// of the K per-bone affine maps.  The reference materialises K copies of the [B,V,3] vertex array; here a
// vertex is read once (12 B), the image's bones (K x 7 floats) and transforms (K x 12 floats) sit in LDS,
// and the softmax runs in registers over the K cached logits.  Weights are recomputed in backward instead of stored.
// HBM traffic: 12 B/vertex in (shared prior: once per image from L2), 12 B/vertex out.
#include "a3d_common.h"

#define SK_THREADS 256
#define SK_MAXK 64

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

// KMAX = compile-time bound on K: the K logits of a vertex (a sqrt each) are computed ONCE and stay in registers for the
// max / sum / blend passes (the first version recomputed them per pass: 3 sqrt chains per bone per vertex).
template <int KMAX>
__global__ __launch_bounds__(SK_THREADS) void sk_fwd_kernel(const float* __restrict__ v, int v_batch, const float* __restrict__ bones,
                                                            int bones_batch, const float* __restrict__ T, int V, int K,
                                                            float neg_inv_temp, float* __restrict__ out, float* __restrict__ weights,
                                                            float* __restrict__ clear, int n_clear) {
    __shared__ SkBone s_bone[SK_MAXK];
    __shared__ float s_T[SK_MAXK * 12];
    const int b = blockIdx.y;
    // the backward's per-image transform gradient (accumulated there with atomics) cleared here: one memset less on the backward path
    for (int z = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; z < n_clear; z += gridDim.x * gridDim.y * blockDim.x) clear[z] = 0.f;
    sk_stage(bones + (bones_batch == 1 ? 0ll : (long long)b * K * 6), T + (long long)b * K * 12, K, s_bone, s_T);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const float* p = v + ((v_batch == 1 ? 0ll : (long long)b * V) + i) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    float lg[KMAX];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        lg[k] = k < K ? sk_logit(s_bone[k], px, py, pz, neg_inv_temp) : -INFINITY;
        m = fmaxf(m, lg[k]);
    }
    float s = 0.f, ox = 0.f, oy = 0.f, oz = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            const float e = __expf(lg[k] - m);
            lg[k] = e;
            const float* t = s_T + 12 * k;
            s += e;
            ox += e * (t[0] * px + t[1] * py + t[2] * pz + t[3]);
            oy += e * (t[4] * px + t[5] * py + t[6] * pz + t[7]);
            oz += e * (t[8] * px + t[9] * py + t[10] * pz + t[11]);
        }
    }
    const float inv = 1.f / s;
    float* o = out + ((long long)b * V + i) * 3;
    o[0] = ox * inv; o[1] = oy * inv; o[2] = oz * inv;
    if (weights) {  // [K, Bw, V]; only images that own distinct weights write
        const int Bw = (v_batch == 1 && bones_batch == 1) ? 1 : (int)gridDim.y;
        if (b < Bw) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) weights[((long long)k * Bw + b) * V + i] = lg[k] * inv;
        }
    }
}

// ---- backward, one kernel, two phases per 256-vertex chunk.
// phase 1 (thread = vertex): softmax weights once (into LDS, [K][256]) and g_v = sum_k w_k R_k^T g (written per image; a shared
//          canonical mesh is summed over the batch by the caller).
// phase 2 (wave = bone group): g_T[b,k] = sum_v w_k(v) * g(v) (x) [v,1] -- a [K x V].[V x 12] product per image.  Wave w owns the bones
//          [w*KG, (w+1)*KG); every lane keeps KG x 12 partial sums in registers while it strides over the chunk's vertices in LDS, and
//          the cross-lane reduction happens once per block (KG*12 butterfly sums), not once per vertex.
template <int KG>
__global__ __launch_bounds__(SK_THREADS) void sk_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ v, int v_batch,
                                                            const float* __restrict__ bones, int bones_batch, const float* __restrict__ T,
                                                            int V, int K, float neg_inv_temp, int chunks_per_block, float* __restrict__ g_v,
                                                            float* __restrict__ g_T) {
    __shared__ SkBone s_bone[SK_MAXK];
    __shared__ float s_T[SK_MAXK * 12];
    __shared__ float s_w[4 * KG][SK_THREADS];
    __shared__ float s_x[6][SK_THREADS];  // px py pz gx gy gz
    const int b = blockIdx.y;
    sk_stage(bones + (bones_batch == 1 ? 0ll : (long long)b * K * 6), T + (long long)b * K * 12, K, s_bone, s_T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k0 = wave * KG;
    float acc[KG][12];
#pragma unroll
    for (int kk = 0; kk < KG; ++kk)
#pragma unroll
        for (int q = 0; q < 12; ++q) acc[kk][q] = 0.f;
    const float* vb = v + (v_batch == 1 ? 0ll : (long long)b * V * 3);
    const float* gb = g_out + (long long)b * V * 3;
    for (int ch = 0; ch < chunks_per_block; ++ch) {
        const int base = (blockIdx.x * chunks_per_block + ch) * SK_THREADS;
        if (base >= V) break;  // uniform
        __syncthreads();       // previous chunk's LDS fully consumed (also orders sk_stage on the first trip)
        {
            const int i = base + threadIdx.x;
            float px = 0.f, py = 0.f, pz = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
            const bool ok = i < V;
            if (ok) {
                px = vb[3ll * i]; py = vb[3ll * i + 1]; pz = vb[3ll * i + 2];
                gx = gb[3ll * i]; gy = gb[3ll * i + 1]; gz = gb[3ll * i + 2];
            }
            float lg[4 * KG];  // the logits (a sqrt each) once, in registers
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4 * KG; ++k) {
                lg[k] = k < K ? sk_logit(s_bone[k], px, py, pz, neg_inv_temp) : -INFINITY;
                m = fmaxf(m, lg[k]);
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4 * KG; ++k) {
                lg[k] = k < K ? __expf(lg[k] - m) : 0.f;
                s += lg[k];
            }
            const float inv = 1.f / s;
            float ox = 0.f, oy = 0.f, oz = 0.f;
#pragma unroll
            for (int k = 0; k < 4 * KG; ++k) {
                if (k < K) {
                    const float w = ok ? lg[k] * inv : 0.f;
                    s_w[k][threadIdx.x] = w;
                    const float* t = s_T + 12 * k;
                    ox += w * (t[0] * gx + t[4] * gy + t[8] * gz);
                    oy += w * (t[1] * gx + t[5] * gy + t[9] * gz);
                    oz += w * (t[2] * gx + t[6] * gy + t[10] * gz);
                }
            }
            s_x[0][threadIdx.x] = px; s_x[1][threadIdx.x] = py; s_x[2][threadIdx.x] = pz;
            s_x[3][threadIdx.x] = gx; s_x[4][threadIdx.x] = gy; s_x[5][threadIdx.x] = gz;
            if (ok && g_v) {
                float* o = g_v + ((long long)b * V + i) * 3;
                o[0] = ox; o[1] = oy; o[2] = oz;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SK_THREADS / 64; ++j) {
            const int t = j * 64 + lane;
            const float px = s_x[0][t], py = s_x[1][t], pz = s_x[2][t], gx = s_x[3][t], gy = s_x[4][t], gz = s_x[5][t];
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
                const int k = k0 + kk;
                const float w = k < K ? s_w[k < K ? k : 0][t] : 0.f;
                const float wx = w * gx, wy = w * gy, wz = w * gz;
                acc[kk][0] += wx * px; acc[kk][1] += wx * py; acc[kk][2] += wx * pz; acc[kk][3] += wx;
                acc[kk][4] += wy * px; acc[kk][5] += wy * py; acc[kk][6] += wy * pz; acc[kk][7] += wy;
                acc[kk][8] += wz * px; acc[kk][9] += wz * py; acc[kk][10] += wz * pz; acc[kk][11] += wz;
            }
        }
    }
    // cross-lane reduction: four DPP steps leave every 16-lane row's sum in its lanes (no LDS traffic), the 4 rows x 4 waves meet
    // in LDS, and thread (bone, component) adds the block's total to g_T: consecutive threads -> consecutive floats, i.e. the
    // block's K*12 atomics are K*12/16 line requests (line-coalesced device atomics are ~10x cheaper than scattered ones)
    __shared__ float s_red[4][SK_THREADS / 64][KG * 12];
#pragma unroll
    for (int kk = 0; kk < KG; ++kk)
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            float r = acc[kk][q];
            r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
            r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
            r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x141, 0xF, 0xF, true));  // row_half_mirror
            r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x140, 0xF, 0xF, true));  // row_mirror
            if ((lane & 15) == 0) s_red[lane >> 4][wave][kk * 12 + q] = r;
        }
    __syncthreads();
    for (int t = threadIdx.x; t < (SK_THREADS / 64) * KG * 12; t += SK_THREADS) {
        const int w = t / (KG * 12), j = t - w * (KG * 12);
        const int k = w * KG + j / 12;
        if (k < K) atomicAdd(g_T + ((long long)b * K + k) * 12 + (j % 12), s_red[0][w][j] + s_red[1][w][j] + s_red[2][w][j] + s_red[3][w][j]);
    }
}

extern "C" int a3d_skin_fwd(const float* v, int v_batch, const float* bones, int bones_batch, const float* T, int B, int V, int K,
                            float temperature, float* out, float* weights_or_null, float* g_T_to_clear_or_null, a3d_stream_t stream) {
    A3D_CHECK_ARG(v && bones && T && out);
    A3D_CHECK_ARG(B > 0 && V > 0 && K > 0 && K <= SK_MAXK && temperature > 0.f);
    A3D_CHECK_ARG((v_batch == 1 || v_batch == B) && (bones_batch == 1 || bones_batch == B));
    const dim3 grid(a3d_div_up(V, SK_THREADS), B), block(SK_THREADS);
    hipStream_t s = (hipStream_t)stream;
    const float nit = -1.f / temperature;
    const int ncl = g_T_to_clear_or_null ? B * K * 12 : 0;
    if (K <= 20) hipLaunchKernelGGL((sk_fwd_kernel<20>), grid, block, 0, s, v, v_batch, bones, bones_batch, T, V, K, nit, out, weights_or_null, g_T_to_clear_or_null, ncl);
    else if (K <= 32) hipLaunchKernelGGL((sk_fwd_kernel<32>), grid, block, 0, s, v, v_batch, bones, bones_batch, T, V, K, nit, out, weights_or_null, g_T_to_clear_or_null, ncl);
    else hipLaunchKernelGGL((sk_fwd_kernel<SK_MAXK>), grid, block, 0, s, v, v_batch, bones, bones_batch, T, V, K, nit, out, weights_or_null, g_T_to_clear_or_null, ncl);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_skin_bwd(const float* g_out, const float* v, int v_batch, const float* bones, int bones_batch, const float* T, int B,
                            int V, int K, float temperature, float* g_v_or_null, float* g_T, int g_T_is_clear, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_out && v && bones && T && g_T);
    A3D_CHECK_ARG(B > 0 && V > 0 && K > 0 && K <= SK_MAXK && temperature > 0.f);
    A3D_CHECK_ARG((v_batch == 1 || v_batch == B) && (bones_batch == 1 || bones_batch == B));
    hipStream_t s = (hipStream_t)stream;
    const float nit = -1.f / temperature;
    if (!g_T_is_clear) A3D_HIP(hipMemsetAsync(g_T, 0, sizeof(float) * (size_t)B * K * 12, s));
    // one 256-vertex chunk per block while that already gives >= 1024 blocks; more chunks per block for very large meshes
    const int chunks = a3d_div_up(V, SK_THREADS);
    int cpb = a3d_div_up((long long)chunks * B, 4096);
    if (cpb < 1) cpb = 1;
    const dim3 grid(a3d_div_up(chunks, cpb), B), block(SK_THREADS);
    if (K <= 20) hipLaunchKernelGGL((sk_bwd_kernel<5>), grid, block, 0, s, g_out, v, v_batch, bones, bones_batch, T, V, K, nit, cpb, g_v_or_null, g_T);
    else if (K <= 32) hipLaunchKernelGGL((sk_bwd_kernel<8>), grid, block, 0, s, g_out, v, v_batch, bones, bones_batch, T, V, K, nit, cpb, g_v_or_null, g_T);
    else hipLaunchKernelGGL((sk_bwd_kernel<16>), grid, block, 0, s, g_out, v, v_batch, bones, bones_batch, T, V, K, nit, cpb, g_v_or_null, g_T);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
