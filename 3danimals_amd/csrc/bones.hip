// Per-bone world transforms from the kinematic chain on gfx950: one work-group per instance, links built once in LDS.
//
// Replaces the inner loops of skinning() (model/geometry/skinning.py:389-417): for every bone the reference walks its
// kinematic chain leaf -> root composing Rest_i . Rot(euler_i) . Rest_i^-1 one 4x4 torch op at a time (~24k aten calls
// per forward at K=20, B=16).  Here the K links of an instance are built once and bone k multiplies the <= 8 links of its chain:
//     L_i = [ R_i Rot_i R_i^T | t_i - R_i Rot_i R_i^T t_i ],   M_k = L_root ... L_parent(k) L_k        (3x4 affine)
// with R_i the rest frame from the bone direction (columns right, up, forward; right ~ +x; skinning.py:251-270), t_i the
// bone's start joint and Rot_i = Rx Ry Rz (PyTorch3D 'XYZ', skinning.py:285-340).
// Backward: g_L_j = P_j^T g_M S_j^T with prefix/suffix products of the chain, pushed through the conjugation and the Euler
// factors onto the three angles (summed per link in a fixed order: bit-reproducible).  Bones carry no gradient.
// A few hundred threads in total; the point is 2 launches instead of ~80 tiny ones on a host-bound stretch of the step.
#include "bones_common.h"

// ---- one work-group per instance: the K link matrices are built ONCE (one thread each) into LDS, then every bone walks its
// chain over LDS copies (<= 8 products of 3x4 affines).  The first version rebuilt every link of every chain per thread
// (8 x sincos/normalise chains in series, 13 us / 44 us for 320 threads of work); this form is a few hundred flops deep.
#define BN_THREADS 256

__global__ __launch_bounds__(BN_THREADS) void bn_fwd_kernel(const float* __restrict__ bones, int bones_batch, const float* __restrict__ angles,
                                                            const int* __restrict__ chain, int K, int D, float* __restrict__ M) {
    __shared__ float s_L[BN_MAXK][13];  // 13: odd stride, conflict-free row access
    __shared__ int s_chain[BN_MAXK * BN_MAXD];  // a serial walk over global memory costs ~0.2 us per dependent load
    const int n = blockIdx.x;
    const float* bb = bones + (bones_batch == 1 ? 0ll : (long long)n * K * 6);
    const float* aa = angles + (long long)n * K * 3;
    for (int w = threadIdx.x; w < K * D; w += blockDim.x) s_chain[w] = chain[w];
    for (int i = threadIdx.x; i < K; i += blockDim.x) bn_store(s_L[i], bn_link(bb + 6 * i, aa + 3 * i));
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        A34 acc = bn_identity();
        for (int j = 0; j < D; ++j) {
            const int i = s_chain[k * D + j];
            if (i >= 0) acc = bn_mul(acc, bn_load(s_L[i]));
        }
        bn_store(M + ((long long)n * K + k) * 12, acc);
    }
}

// backward: one work-group per instance runs bn_chain_adjoint (bones_common.h)
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_kernel(const float* __restrict__ g_M, const float* __restrict__ bones, int bones_batch,
                                                            const float* __restrict__ angles, const int* __restrict__ chain, int K, int D,
                                                            float* __restrict__ g_angles) {
    extern __shared__ float s_dyn[];
    const int n = blockIdx.x;
    bn_chain_adjoint<false>(g_M + (long long)n * K * 12, nullptr, bones + (bones_batch == 1 ? 0ll : (long long)n * K * 6),
                            angles + (long long)n * K * 3, chain, K, D, g_angles + (long long)n * K * 3, s_dyn);
}

extern "C" int a3d_bone_transforms_fwd(const float* bones, int bones_batch, const float* angles, const int32_t* chain, int N, int K, int D,
                                       float* M, a3d_stream_t stream) {
    A3D_CHECK_ARG(bones && angles && chain && M && N > 0 && K > 0 && K <= BN_MAXK && D > 0 && D <= BN_MAXD);
    A3D_CHECK_ARG(bones_batch == 1 || bones_batch == N);
    hipLaunchKernelGGL(bn_fwd_kernel, dim3(N), dim3(BN_THREADS), 0, (hipStream_t)stream, bones, bones_batch, angles, chain, K, D, M);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_bone_transforms_bwd(const float* g_M, const float* bones, int bones_batch, const float* angles, const int32_t* chain, int N,
                                       int K, int D, float* g_angles, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_M && bones && angles && chain && g_angles && N > 0 && K > 0 && K <= BN_MAXK && D > 0 && D <= BN_MAXD);
    A3D_CHECK_ARG(bones_batch == 1 || bones_batch == N);
    // every g_angles element is written by its owner thread: no memset, no atomics
    if (bn_bwd_lds(K, D) > 64 * 1024)  // K > ~48 with the deepest chains: opt in to the large-LDS launch (cheap, per call: no cached state)
        A3D_HIP(hipFuncSetAttribute((const void*)bn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bn_bwd_lds(K, D)));
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(N), dim3(BN_THREADS), bn_bwd_lds(K, D), (hipStream_t)stream, g_M, bones, bones_batch, angles, chain, K,
                       D, g_angles);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

// ================================================================================================ estimate_bones on the device (round 6)
// The heuristic skeleton of /root/reference/model/geometry/skinning.py:50-248 as ONE launch of ONE work-group: restated on torch (round 2) it
// is ~245 launches of 2-5 us each -- 634 us of the Fauna step, which re-estimates the bones in every iteration
// (InstancePredictorFauna.py:79-100; tools/glue_attribution.py) -- for a few thousand vertices and three dozen output rows.
//   phase A  per instance: centroid (mean over the vertices), the two spine ends (arg-max / arg-min of z, optionally among the vertices not
//            far below the centroid: 'z_minmax_y+'; first index on ties, as torch.argmax / argmin)
//   phase B  the quantiles the leg quadrants hang on -- of ALL values of the call, as the reference's tensor.quantile() is --, as order
//            statistics found by a radix select over monotone 32-bit keys (four 8-bit passes, every wanted rank of the phase in the same
//            pass: one LDS histogram each), then torch.quantile's linear interpolation; the Fauna variant first finds the y threshold, then
//            the six quantiles of x and z among the vertices below it (the reference's xs[low].quantile(.), skinning.py:160-166)
//   phase C  per instance and quadrant: the lowest vertex (arg-min of y among the quadrant's vertices; an empty quadrant -- where the
//            reference drops into pdb, :183 -- yields vertex 0 and clears the ``ok`` flag: a deferred check of the caller)
//   phase D  one thread per instance: spine joints, body bones, the attachment joints of legs 0 / 1 (instance 0, unless prescribed;
//            handed out for the caller's kinematic chain), leg joints, leg bones.
// Same operations as the torch restatement up to the summation order of the centroid (a float32 tree sum here): the goldens taken from the
// reference itself hold to 1e-6 (tests/test_gpu_parity.py::test_estimate_bones_on_device_against_reference_golden).
#define EB_THREADS 1024
#define EB_MAXN 32
#define EB_MAXSEL 12

__device__ __forceinline__ unsigned eb_key(float f) {  // monotone float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float eb_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }
__device__ __forceinline__ float eb_lerp(float a, float b, float w) { return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w); }  // torch.lerp

struct EbArg { float v; int i; };
// arg-max (MAX) / arg-min over the work-group, first index on ties; every thread gets the result
template <bool MAX>
__device__ __forceinline__ EbArg eb_block_arg(EbArg a, EbArg* s_w) {
    auto better = [](const EbArg& x, const EbArg& y) { return MAX ? (x.v > y.v || (x.v == y.v && x.i < y.i)) : (x.v < y.v || (x.v == y.v && x.i < y.i)); };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        EbArg b;
        b.v = __shfl_xor(a.v, o, 64); b.i = __shfl_xor(a.i, o, 64);
        if (better(b, a)) a = b;
    }
    __syncthreads();  // (s_w free)
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = a;
    __syncthreads();
    EbArg r = s_w[0];
    for (int w = 1; w < EB_THREADS / 64; ++w)
        if (better(s_w[w], r)) r = s_w[w];
    return r;
}
__device__ __forceinline__ float eb_block_sum(float v, float* s_f) {
    v = a3d_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_f[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < EB_THREADS / 64; ++w) t += s_f[w];
    return t;
}

// the values of ranks s_rank[0..m) (0-based, ascending order) among f(0..total): radix select, all m targets in the same four passes
template <class F>
__device__ __forceinline__ void eb_multiselect(F f, int total, int m, int* s_rank, unsigned* s_prefix, int (*s_hist)[256]) {
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = threadIdx.x; i < m * 256; i += EB_THREADS) s_hist[i >> 8][i & 255] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < total; i += EB_THREADS) {
            const unsigned k = eb_key(f(i));
            for (int t = 0; t < m; ++t)
                if (pass == 0 || (k >> (shift + 8)) == s_prefix[t]) atomicAdd(&s_hist[t][(k >> shift) & 255u], 1);
        }
        __syncthreads();
        if ((int)threadIdx.x < m) {
            const int t = threadIdx.x;
            int r = s_rank[t], acc = 0, digit = 255;
            for (int b = 0; b < 256; ++b) {
                const int c = s_hist[t][b];
                if (acc + c > r) { digit = b; break; }
                acc += c;
            }
            s_rank[t] = r - acc;
            s_prefix[t] = pass == 0 ? (unsigned)digit : ((s_prefix[t] << 8) | (unsigned)digit);
        }
        __syncthreads();
    }
}

struct EbParams {
    const float* pos;  // [N, V, 3]
    int N, V, n_body, n_leg, yplus, fauna;
    float y_q;         // Fauna: bone_y_threshold
    float blend[17], ramp[9];
    int attach[4];     // body joint of every leg; attach[0], attach[1] < 0: found here (instance 0) and copied to legs 3 / 2
    float* bones;      // [N, n_body + 4 n_leg, 2, 3]
    int* nearest;      // [2] the attachment joints of legs 0 / 1 as used
    int* ok;           // [1] 1 = every quadrant of every instance holds a vertex
};

__global__ __launch_bounds__(EB_THREADS) void eb_kernel(const EbParams a) {
    __shared__ EbArg s_w[EB_THREADS / 64];
    __shared__ float s_f[EB_THREADS / 64];
    __shared__ float s_cent[EB_MAXN][3];
    __shared__ int s_ab[EB_MAXN][2], s_foot[EB_MAXN][4];
    __shared__ int s_rank[EB_MAXSEL];
    __shared__ unsigned s_prefix[EB_MAXSEL];
    __shared__ int s_hist[EB_MAXSEL][256];
    __shared__ float s_q[8];  // margins / centres of the quadrants
    __shared__ int s_ok;
    const int tid = threadIdx.x, N = a.N, V = a.V, total = N * V;
    const float* __restrict__ pos = a.pos;
    if (tid == 0) s_ok = 1;
    // ---- phase A
    for (int n = 0; n < N; ++n) {
        const float* p = pos + (long long)n * V * 3;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int v = tid; v < V; v += EB_THREADS) { sx += p[3 * v]; sy += p[3 * v + 1]; sz += p[3 * v + 2]; }
        sx = eb_block_sum(sx, s_f); sy = eb_block_sum(sy, s_f); sz = eb_block_sum(sz, s_f);
        const float cx = sx / (float)V, cy = sy / (float)V, cz = sz / (float)V;
        if (tid == 0) { s_cent[n][0] = cx; s_cent[n][1] = cy; s_cent[n][2] = cz; }
        EbArg hi = {-INFINITY, 0x7fffffff}, lo = {INFINITY, 0x7fffffff};
        for (int v = tid; v < V; v += EB_THREADS) {
            const float z = p[3 * v + 2];
            float ka = z, kb = z;
            if (a.yplus) {  // z * upper + (-+1e6) * (1 - upper)   (skinning.py:103-107)
                const float up = p[3 * v + 1] > (cy - 0.5f) ? 1.f : 0.f;
                ka = z * up + (-1e6f) * (1.f - up);
                kb = z * up + 1e6f * (1.f - up);
            }
            if (ka > hi.v) { hi.v = ka; hi.i = v; }
            if (kb < lo.v) { lo.v = kb; lo.i = v; }
        }
        hi = eb_block_arg<true>(hi, s_w);
        lo = eb_block_arg<false>(lo, s_w);
        if (tid == 0) { s_ab[n][0] = hi.i < V ? hi.i : 0; s_ab[n][1] = lo.i < V ? lo.i : 0; }
    }
    __syncthreads();
    if (a.n_leg > 0) {
        // ---- phase B: quantiles over ALL values of the call
        auto ranks_of = [&](float q, int n, int slot, float* w_out) {  // pos = q (n - 1), float32 as torch.quantile computes it
            const float ps = q * (float)(n - 1), lo = floorf(ps), hi = ceilf(ps);
            s_rank[slot] = max((int)lo, 0); s_rank[slot + 1] = max((int)hi, 0);
            *w_out = ps - lo;
        };
        if (!a.fauna) {
            float w95 = 0.f, w05 = 0.f;
            if (tid == 0) { ranks_of(0.95f, total, 0, &w95); ranks_of(0.05f, total, 2, &w05); s_q[6] = w95; s_q[7] = w05; }
            __syncthreads();
            eb_multiselect([&](int i) { return pos[3ll * i]; }, total, 4, s_rank, s_prefix, s_hist);
            if (tid == 0) {
                const float q95 = eb_lerp(eb_unkey(s_prefix[0]), eb_unkey(s_prefix[1]), s_q[6]);
                const float q05 = eb_lerp(eb_unkey(s_prefix[2]), eb_unkey(s_prefix[3]), s_q[7]);
                s_q[0] = (q95 - q05) * 0.2f;  // margin
            }
            __syncthreads();
        } else {
            float w = 0.f;
            if (tid == 0) { ranks_of(a.y_q, total, 0, &w); s_q[6] = w; }
            __syncthreads();
            eb_multiselect([&](int i) { return pos[3ll * i + 1]; }, total, 2, s_rank, s_prefix, s_hist);
            const float thr = eb_lerp(eb_unkey(s_prefix[0]), eb_unkey(s_prefix[1]), s_q[6]);
            int cnt = 0;
            for (int i = tid; i < total; i += EB_THREADS) cnt += pos[3ll * i + 1] < thr ? 1 : 0;
            const int nlow = (int)(eb_block_sum((float)cnt, s_f) + 0.5f);  // (exact: counts below 2^24)
            __syncthreads();
            if (tid == 0) {
                float w0, w1, w2;
                ranks_of(0.5f, nlow, 0, &w0); ranks_of(0.95f, nlow, 2, &w1); ranks_of(0.05f, nlow, 4, &w2);
                for (int k = 0; k < 6; ++k) s_rank[6 + k] = s_rank[k];
                s_q[5] = w0; s_q[6] = w1; s_q[7] = w2;
            }
            __syncthreads();
            // x among the low vertices in slots 0..5, z in slots 6..11: one set of four passes for both
            eb_multiselect([&](int i) { return pos[3ll * i + 1] < thr ? pos[3ll * i] : INFINITY; }, total, 6, s_rank, s_prefix, s_hist);
            eb_multiselect([&](int i) { return pos[3ll * i + 1] < thr ? pos[3ll * i + 2] : INFINITY; }, total, 6, s_rank + 6, s_prefix + 6, s_hist + 6);
            if (tid == 0) {
                float q[2][3];
                for (int c = 0; c < 2; ++c)
                    for (int k = 0; k < 3; ++k) q[c][k] = eb_lerp(eb_unkey(s_prefix[6 * c + 2 * k]), eb_unkey(s_prefix[6 * c + 2 * k + 1]), s_q[5 + k]);
                s_q[0] = q[0][0]; s_q[1] = q[1][0];                                      // x0, z0 (medians)
                s_q[2] = (q[0][1] - q[0][2]) * 0.2f; s_q[3] = (q[1][1] - q[1][2]) * 0.2f;  // mx, mz
            }
            __syncthreads();
        }
        // ---- phase C: the foot of every quadrant of every instance
        for (int n = 0; n < N; ++n) {
            const float* p = pos + (long long)n * V * 3;
            EbArg best[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { best[k].v = INFINITY; best[k].i = 0x7fffffff; }
            for (int v = tid; v < V; v += EB_THREADS) {
                const float x = p[3 * v], y = p[3 * v + 1], z = p[3 * v + 2];
                bool in[4];
                if (!a.fauna) {
                    const float m = s_q[0];
                    in[0] = x > m && z > 0.f; in[1] = x > m && z < 0.f; in[2] = x < -m && z < 0.f; in[3] = x < -m && z > 0.f;
                } else {
                    const float x0 = s_q[0], z0 = s_q[1], mx = s_q[2], mz = s_q[3];
                    in[0] = (x - x0 > mx) && (z - z0 > mz); in[1] = (x - x0 > mx) && (z < z0);
                    in[2] = (x - x0 < -mx) && (z < z0); in[3] = (x - x0 < -mx) && (z - z0 > mz);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (in[k] && y < best[k].v) { best[k].v = y; best[k].i = v; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const EbArg r = eb_block_arg<false>(best[k], s_w);
                if (tid == 0) {
                    s_foot[n][k] = r.i < V ? r.i : 0;  // (an empty quadrant: vertex 0, like argmin over all-inf)
                    if (r.i >= V) s_ok = 0;
                }
            }
        }
        __syncthreads();
    }
    // ---- phase D: joints and bones, one thread per instance (instance 0 first: its attachment joints serve all)
    __shared__ int s_attach[4];
    const int nj = a.n_body + 1, half = a.n_body / 2, nb2 = (nj + 1) / 2, K = a.n_body + 4 * a.n_leg;
    for (int round = 0; round < 2; ++round) {
        const int n = round == 0 ? 0 : tid;
        if ((round == 0 ? tid == 0 : (tid > 0 && tid < N))) {
            const float* p = pos + (long long)n * V * 3;
            float pa[3] = {0.f, p[3 * s_ab[n][0] + 1], p[3 * s_ab[n][0] + 2]}, pb[3] = {0.f, p[3 * s_ab[n][1] + 1], p[3 * s_ab[n][1] + 2]};
            float mid[3] = {0.f, s_cent[n][1] + (a.n_leg > 0 ? 0.5f : 0.f), s_cent[n][2]};
            float joints[33][3];
            for (int i = 0; i < nb2; ++i) {
                const float bl = a.blend[i];
                for (int c = 0; c < 3; ++c) {
                    const float ja = pa[c] * (1.f - bl) + mid[c] * bl, jb = pb[c] * bl + mid[c] * (1.f - bl);
                    if (i < nb2 - 1) joints[i][c] = ja;  // joints_a[:-1]
                    joints[nb2 - 1 + i][c] = jb;
                }
            }
            float* out = a.bones + (long long)n * K * 6;
            // body bones: (i + 1, i) for the first half, then (i, i + 1) for i = n_body - 1 .. half   (skinning.py:128-141)
            int bone = 0;
            for (int i = 0; i < half; ++i, ++bone)
                for (int c = 0; c < 3; ++c) { out[6 * bone + c] = joints[i + 1][c]; out[6 * bone + 3 + c] = joints[i][c]; }
            for (int i = a.n_body - 1; i >= half; --i, ++bone)
                for (int c = 0; c < 3; ++c) { out[6 * bone + c] = joints[i][c]; out[6 * bone + 3 + c] = joints[i + 1][c]; }
            if (a.n_leg > 0) {
                if (round == 0) {  // attachment joints (instance 0): nearest body bone end in z to the foot, first index on ties
                    int att[4] = {a.attach[0], a.attach[1], a.attach[2], a.attach[3]};
                    for (int l = 0; l < 2; ++l)
                        if (att[l] < 0) {
                            const float fz = p[3 * s_foot[0][l] + 2];
                            float bd = INFINITY;
                            int bi = 0;
                            for (int k = 0; k < a.n_body; ++k) {
                                const float d = fabsf(out[6 * k + 3 + 2] - fz);
                                if (d < bd) { bd = d; bi = k; }
                            }
                            att[l] = bi;
                        }
                    if (a.attach[2] < 0) att[2] = att[1];
                    if (a.attach[3] < 0) att[3] = att[0];
                    for (int l = 0; l < 4; ++l) s_attach[l] = att[l];
                    a.nearest[0] = att[0]; a.nearest[1] = att[1];
                    a.ok[0] = s_ok;
                }
            } else if (round == 0) {
                a.ok[0] = 1;
            }
        }
        __syncthreads();
        if (a.n_leg > 0 && (round == 0 ? tid == 0 : (tid > 0 && tid < N))) {
            const float* p = pos + (long long)n * V * 3;
            float* out = a.bones + (long long)n * K * 6;
            for (int l = 0; l < 4; ++l) {
                const float* foot = p + 3 * s_foot[n][l];
                const float* anchor = out + 6 * s_attach[l] + 3;  // bones_pred[:, :, body_bone_idx, 1]
                float lj[9][3];
                for (int j = 0; j <= a.n_leg; ++j)
                    for (int c = 0; c < 3; ++c) lj[j][c] = foot[c] * (1.f - a.ramp[j]) + anchor[c] * a.ramp[j];
                for (int i = 0; i < a.n_leg; ++i) {  // leg bone i = (joint i + 1, joint i)
                    float* o = out + 6 * (a.n_body + l * a.n_leg + i);
                    for (int c = 0; c < 3; ++c) { o[c] = lj[i + 1][c]; o[3 + c] = lj[i][c]; }
                }
            }
        }
        __syncthreads();
    }
}

extern "C" int a3d_estimate_bones(const a3d_estimate_bones_args* args, a3d_stream_t stream) {
    A3D_CHECK_ARG(args && args->size >= sizeof(a3d_estimate_bones_args));
    A3D_CHECK_ARG(args->pos && args->bones && args->nearest && args->ok && args->N > 0 && args->N <= EB_MAXN && args->V > 0);
    A3D_CHECK_ARG((long long)args->N * args->V <= (1 << 22) && args->n_body >= 2 && args->n_body % 2 == 0 && args->n_body <= 32 && args->n_leg >= 0 && args->n_leg <= 8);
    EbParams p;
    p.pos = args->pos; p.N = args->N; p.V = args->V; p.n_body = args->n_body; p.n_leg = args->n_leg; p.yplus = args->body_mode_y_plus;
    p.fauna = args->use_y_threshold; p.y_q = args->y_threshold;
    for (int i = 0; i < 17; ++i) p.blend[i] = args->blend[i];
    for (int i = 0; i < 9; ++i) p.ramp[i] = args->ramp[i];
    for (int i = 0; i < 4; ++i) p.attach[i] = args->attach[i];
    A3D_CHECK_ARG(args->n_leg == 0 || ((p.attach[0] < args->n_body && p.attach[1] < args->n_body && p.attach[2] < args->n_body && p.attach[3] < args->n_body)));
    p.bones = args->bones; p.nearest = args->nearest; p.ok = args->ok;
    hipLaunchKernelGGL(eb_kernel, dim3(1), dim3(EB_THREADS), 0, (hipStream_t)stream, p);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
