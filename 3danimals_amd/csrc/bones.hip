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
//   phase D  one thread per bone end: spine joints, body bones, the attachment joints of legs 0 / 1 (instance 0, unless prescribed;
//            handed out for the caller's kinematic chain), leg joints, leg bones.
// Same operations as the torch restatement up to the summation order of the centroid (a float32 tree sum here): the goldens taken from the
// reference itself hold to 1e-6 (tests/test_gpu_parity.py::test_estimate_bones_on_device_against_reference_golden).
#define EB_THREADS 1024
#define EB_MAXN 32
#define EB_MAXSEL 12
#define EB_MAXGROUP 6
#define EB_COPIES 16

__device__ __forceinline__ unsigned eb_key(float f) {  // monotone float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float eb_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }
__device__ __forceinline__ float eb_lerp(float a, float b, float w) { return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w); }  // torch.lerp

struct EbArg { float v; int i; };
__device__ __forceinline__ float eb_block_sum(float v, float* s_f) {
    v = a3d_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_f[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < EB_THREADS / 64; ++w) t += s_f[w];
    return t;
}

// the values of ranks s_rank[0..NG*mg) (0-based, ascending order): targets [g mg, (g+1) mg) among the values f(i, g), i in 0..total -- a
// radix select, all targets in the same four passes over the data.  Targets whose prefixes agree so far (the two neighbours a quantile
// interpolates between nearly always do, to the last pass) share one histogram: s_rep[t] = the first target of t's group with t's prefix.
template <int NG, class F>
__device__ __forceinline__ void eb_multiselect(F f, int total, int mg, int* s_rank, unsigned* s_prefix, int (*s_hist)[256], int (*s_h0)[256][EB_COPIES]) {
    __shared__ int s_rep[EB_MAXSEL], s_urep[NG][EB_MAXGROUP], s_nu[NG];
    __shared__ unsigned s_upre[NG][EB_MAXGROUP];
    const int m = NG * mg;  // (mg <= EB_MAXGROUP)
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if ((int)threadIdx.x < NG) {  // the distinct prefixes of a group and the targets that own their histograms (padded: no match)
            const int g = threadIdx.x;
            int nu = 0;
            for (int t = g * mg; t < (g + 1) * mg; ++t) {
                int rep = t;
                for (int u = t - 1; u >= g * mg; --u)
                    if (pass == 0 || s_prefix[u] == s_prefix[t]) rep = u;
                s_rep[t] = rep;
                if (rep == t) { s_upre[g][nu] = pass == 0 ? 0u : s_prefix[t]; s_urep[g][nu] = t; ++nu; }
            }
            s_nu[g] = nu;
            for (; nu < EB_MAXGROUP; ++nu) { s_upre[g][nu] = 0xffffffffu; s_urep[g][nu] = g * mg; }
        }
        for (int i = threadIdx.x; i < m * 256; i += EB_THREADS) s_hist[i >> 8][i & 255] = 0;
        __syncthreads();
        if (pass == 0) {
            // every target of a group sees the same histogram in the first pass, and the top byte of a float (sign + 7 exponent bits)
            // takes a handful of values over a mesh: plain LDS atomics would serialise 64 lanes on one address (90 us of this kernel as
            // first written).  EB_COPIES copies of the histogram, one per lane residue: four lanes per address at worst.
            for (int i = threadIdx.x; i < NG * 256 * EB_COPIES; i += EB_THREADS) (&s_h0[0][0][0])[i] = 0;
            __syncthreads();
            const int copy = threadIdx.x & (EB_COPIES - 1);
            for (int i = threadIdx.x; i < total; i += EB_THREADS) {
#pragma unroll
                for (int g = 0; g < NG; ++g) atomicAdd(&s_h0[g][eb_key(f(i, g)) >> 24][copy], 1);
            }
            __syncthreads();
            for (int i = threadIdx.x; i < NG * 256; i += EB_THREADS) {
                int c = 0;
#pragma unroll
                for (int k = 0; k < EB_COPIES; ++k) c += s_h0[i >> 8][i & 255][(k + i) & (EB_COPIES - 1)];  // (rotated: no bank conflicts)
                s_hist[(i >> 8) * mg][i & 255] = c;
            }
        } else {
            unsigned upre[NG][EB_MAXGROUP];
            int urep[NG][EB_MAXGROUP], nu[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                nu[g] = __builtin_amdgcn_readfirstlane(s_nu[g]);
#pragma unroll
                for (int u = 0; u < EB_MAXGROUP; ++u) { upre[g][u] = s_upre[g][u]; urep[g][u] = s_urep[g][u]; }
            }
            for (int i = threadIdx.x; i < total; i += EB_THREADS) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const unsigned k = eb_key(f(i, g)), hi = k >> (shift + 8), digit = (k >> shift) & 255u;
#pragma unroll
                    for (int u = 0; u < EB_MAXGROUP; ++u) {
                        if (u >= nu[g]) break;  // (uniform: usually one to three distinct prefixes)
                        if (hi == upre[g][u]) atomicAdd(&s_hist[urep[g][u]][digit], 1);
                    }
                }
            }
        }
        __syncthreads();
        // the bin that holds each target's rank: ONE WAVE per target (m <= 12 targets, 16 waves) -- a lane sums four neighbouring bins
        // (one 16-byte LDS read), the wave scans the 64 sums with DPP and the lane whose range holds the rank walks its four bins: no
        // work-group barrier inside.  (256 threads per target, four targets at a time, an inclusive scan through shuffles + the wave totals
        // through LDS was two barriers per round of four targets: six per pass, 24 per select -- the call took 52.8 us, 41.5 with this; ONE
        // thread walking the 256 bins, before that, 12 us per pass.  Also measured: the values in the REGISTERS of their owners instead of
        // a planar scratch array, no compaction of the low vertices -- 41.5 -> 38.0 us in the kernel for 128 VGPRs and a 24-byte spill, and the
        // spill's scratch set-up made the Fauna tests run ten times longer end to end: not kept.)
        {
            const int w = threadIdx.x >> 6, ln = threadIdx.x & 63;
            for (int t = w; t < m; t += EB_THREADS / 64) {
                const int4 c = reinterpret_cast<const int4*>(s_hist[s_rep[t]])[ln];
                const int s4 = c.x + c.y + c.z + c.w;
                const int incl = a3d_wave_incl_scan(s4), excl = incl - s4;
                const int r = s_rank[t];
                const unsigned pre = s_prefix[t];
                if (r >= excl && r < incl) {  // exactly one lane per target
                    int b = 4 * ln, base = excl;
                    if (r >= base + c.x) {
                        base += c.x; ++b;
                        if (r >= base + c.y) {
                            base += c.y; ++b;
                            if (r >= base + c.z) { base += c.z; ++b; }
                        }
                    }
                    s_rank[t] = r - base;
                    s_prefix[t] = pass == 0 ? (unsigned)b : ((pre << 8) | (unsigned)b);
                }
            }
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
    float* work;       // [3 N V] scratch
};

__global__ __launch_bounds__(EB_THREADS) void eb_kernel(const EbParams a) {
    __shared__ EbArg s_arg[EB_THREADS / 64][2], s_arg4[EB_THREADS / 64][4];
    __shared__ float s_sum[EB_THREADS / 64][3];
    __shared__ float s_cent[EB_MAXN][3];
    __shared__ int s_ab[EB_MAXN][2], s_foot[EB_MAXN][4];
    __shared__ int s_rank[EB_MAXSEL];
    __shared__ unsigned s_prefix[EB_MAXSEL];
    __shared__ __attribute__((aligned(16))) int s_hist[EB_MAXSEL][256];  // (rows read as int4 by the bin search)
    __shared__ int s_h0[2][256][EB_COPIES];  // first-pass histograms, EB_COPIES copies each
    __shared__ float s_q[8];  // margins / centres of the quadrants
    __shared__ int s_ok;
    const int tid = threadIdx.x, N = a.N, V = a.V, total = N * V;
    const float* __restrict__ pos = a.pos;
    float* ranked = a.work;         // [total]
    float* lowxz = a.work + total;  // [<= total][2] Fauna: x, z of the vertices below the y threshold
    __shared__ int s_nlow;
    __shared__ float s_blend[17], s_ramp[9];
    if (tid < 17) s_blend[tid] = a.blend[tid];
    if (tid < 9) s_ramp[tid] = a.ramp[tid];
    if (tid == 0) { s_ok = 1; s_nlow = 0; }
    A3D_STAMP(0, 0);
    // ---- phase A: the 16 waves spread over the instances (wpi waves each, ipr instances per round): the sums and arg-extrema of an
    // instance meet in LDS, one exchange per reduction for all instances of the round
    const int wave = tid >> 6, lane = tid & 63;
    int wpi = EB_THREADS / 64;
    while (wpi > 1 && (EB_THREADS / 64) / wpi < N) wpi >>= 1;
    const int ipr = (EB_THREADS / 64) / wpi, sub = wave % wpi, w0 = wave - sub;
    auto arg_better = [](bool want_max, const EbArg& x, const EbArg& y) {  // first index on ties, as torch.argmax / argmin
        return want_max ? (x.v > y.v || (x.v == y.v && x.i < y.i)) : (x.v < y.v || (x.v == y.v && x.i < y.i));
    };
    auto wave_arg = [&](bool want_max, EbArg x) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            EbArg y;
            y.v = __shfl_xor(x.v, o, 64); y.i = __shfl_xor(x.i, o, 64);
            if (arg_better(want_max, y, x)) x = y;
        }
        return x;
    };
    for (int n0 = 0; n0 < N; n0 += ipr) {
        const int n = n0 + wave / wpi;
        const bool have = n < N;
        const float* p = pos + (long long)(have ? n : 0) * V * 3;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        if (have)
            for (int v = sub * 64 + lane; v < V; v += wpi * 64) {
                const float x = p[3 * v], y = p[3 * v + 1], z = p[3 * v + 2];
                sx += x; sy += y; sz += z;
                ranked[(long long)n * V + v] = a.fauna ? y : x;  // the coordinate phase B ranks first, planar: a third of the bytes per pass
            }
        sx = a3d_wave_sum(sx); sy = a3d_wave_sum(sy); sz = a3d_wave_sum(sz);
        __syncthreads();  // (s_sum free)
        if (lane == 0) { s_sum[wave][0] = sx; s_sum[wave][1] = sy; s_sum[wave][2] = sz; }
        __syncthreads();
        sx = sy = sz = 0.f;
        for (int k = 0; k < wpi; ++k) { sx += s_sum[w0 + k][0]; sy += s_sum[w0 + k][1]; sz += s_sum[w0 + k][2]; }
        const float cx = sx / (float)V, cy = sy / (float)V, cz = sz / (float)V;
        if (have && sub == 0 && lane == 0) { s_cent[n][0] = cx; s_cent[n][1] = cy; s_cent[n][2] = cz; }
        EbArg hi = {-INFINITY, 0x7fffffff}, lo = {INFINITY, 0x7fffffff};
        if (have)
            for (int v = sub * 64 + lane; v < V; v += wpi * 64) {
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
        hi = wave_arg(true, hi); lo = wave_arg(false, lo);
        if (lane == 0) { s_arg[wave][0] = hi; s_arg[wave][1] = lo; }  // (s_arg: not read since the previous round's barriers)
        __syncthreads();
        if (have && sub == 0 && lane == 0) {
            for (int k = 1; k < wpi; ++k) {
                if (arg_better(true, s_arg[w0 + k][0], hi)) hi = s_arg[w0 + k][0];
                if (arg_better(false, s_arg[w0 + k][1], lo)) lo = s_arg[w0 + k][1];
            }
            s_ab[n][0] = hi.i < V ? hi.i : 0; s_ab[n][1] = lo.i < V ? lo.i : 0;
        }
    }
    __syncthreads();
    A3D_STAMP(0, 1);
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
            eb_multiselect<1>([&](int i, int) { return ranked[i]; }, total, 4, s_rank, s_prefix, s_hist, s_h0);
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
            eb_multiselect<1>([&](int i, int) { return ranked[i]; }, total, 2, s_rank, s_prefix, s_hist, s_h0);
            const float thr = eb_lerp(eb_unkey(s_prefix[0]), eb_unkey(s_prefix[1]), s_q[6]);
            A3D_STAMP(0, 2);
            // the low vertices' x and z, compacted (in any order: they are only ranked): the six quantiles read 8 bytes per LOW vertex
            for (int base = 0; base < total; base += EB_THREADS) {
                const int i = base + tid;
                const bool low = i < total && ranked[i] < thr;
                const unsigned long long lows = __ballot(low);
                int at = 0;
                if (lane == 0 && lows) at = atomicAdd(&s_nlow, __popcll(lows));
                at = __shfl(at, 0, 64) + __popcll(lows & ((1ull << lane) - 1ull));
                if (low) { lowxz[2ll * at] = pos[3ll * i]; lowxz[2ll * at + 1] = pos[3ll * i + 2]; }
            }
            __syncthreads();
            const int nlow = s_nlow;
            if (nlow == 0 && tid == 0) s_ok = 0;  // (torch.quantile of an empty tensor raises in the reference)
            if (tid == 0) {
                float w0, w1, w2;
                ranks_of(0.5f, nlow, 0, &w0); ranks_of(0.95f, nlow, 2, &w1); ranks_of(0.05f, nlow, 4, &w2);
                for (int k = 0; k < 6; ++k) s_rank[6 + k] = s_rank[k];
                s_q[5] = w0; s_q[6] = w1; s_q[7] = w2;
            }
            __syncthreads();
            // x among the low vertices in slots 0..5, z in slots 6..11: one set of four passes for both
            eb_multiselect<2>([&](int i, int g) { return lowxz[2ll * i + g]; }, nlow, 6, s_rank, s_prefix, s_hist, s_h0);
            if (tid == 0) {
                float q[2][3];
                for (int c = 0; c < 2; ++c)
                    for (int k = 0; k < 3; ++k) q[c][k] = eb_lerp(eb_unkey(s_prefix[6 * c + 2 * k]), eb_unkey(s_prefix[6 * c + 2 * k + 1]), s_q[5 + k]);
                s_q[0] = q[0][0]; s_q[1] = q[1][0];                                      // x0, z0 (medians)
                s_q[2] = (q[0][1] - q[0][2]) * 0.2f; s_q[3] = (q[1][1] - q[1][2]) * 0.2f;  // mx, mz
            }
            __syncthreads();
        }
        A3D_STAMP(0, 3);
        // ---- phase C: the foot of every quadrant of every instance (waves over instances, as phase A)
        for (int n0 = 0; n0 < N; n0 += ipr) {
            const int n = n0 + wave / wpi;
            const bool have = n < N;
            const float* p = pos + (long long)(have ? n : 0) * V * 3;
            EbArg best[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { best[k].v = INFINITY; best[k].i = 0x7fffffff; }
            if (have)
                for (int v = sub * 64 + lane; v < V; v += wpi * 64) {
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
            __syncthreads();  // (s_arg4 free)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                best[k] = wave_arg(false, best[k]);
                if (lane == 0) s_arg4[wave][k] = best[k];
            }
            __syncthreads();
            if (have && sub == 0 && lane < 4) {
                EbArg r = s_arg4[w0][lane];
                for (int k = 1; k < wpi; ++k)
                    if (arg_better(false, s_arg4[w0 + k][lane], r)) r = s_arg4[w0 + k][lane];
                s_foot[n][lane] = r.i < V ? r.i : 0;  // (an empty quadrant: vertex 0, like argmin over all-inf)
                if (r.i >= V) s_ok = 0;               // (same value from every writer)
            }
        }
        __syncthreads();
    }
    A3D_STAMP(0, 4);
    // ---- phase D: joints and bones, one thread per bone end; every joint is a closed form of the spine ends and the centroid
    __shared__ int s_attach[4];
    const int nj = a.n_body + 1, half = a.n_body / 2, nb2 = (nj + 1) / 2, K = a.n_body + 4 * a.n_leg;
    auto joint = [&](int n, int j, float* o) {  // joints_a[:-1] ++ joints_b   (skinning.py:118-126)
        const float* p = pos + (long long)n * V * 3;
        const bool head = j < nb2 - 1;
        const int i = head ? j : j - (nb2 - 1), vi = s_ab[n][head ? 0 : 1];
        const float e[3] = {0.f, p[3 * vi + 1], p[3 * vi + 2]}, mid[3] = {0.f, s_cent[n][1] + (a.n_leg > 0 ? 0.5f : 0.f), s_cent[n][2]};
        const float bl = s_blend[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = head ? e[c] * (1.f - bl) + mid[c] * bl : e[c] * bl + mid[c] * (1.f - bl);
    };
    // body bones: (i + 1, i) for the first half, then (i, i + 1) for i = n_body - 1 .. half   (skinning.py:128-141)
    auto body_end = [&](int n, int bone, int e, float* o) {
        const int i = bone < half ? bone : a.n_body - 1 - (bone - half);
        joint(n, bone < half ? (e == 0 ? i + 1 : i) : (e == 0 ? i : i + 1), o);
    };
    if (a.n_leg > 0) {
        if (tid < 2) {  // attachment joints (instance 0): nearest body bone end in z to the foot, first index on ties
            int att = a.attach[tid];
            if (att < 0) {
                const float fz = pos[3 * s_foot[0][tid] + 2];
                float bd = INFINITY;
                att = 0;
                for (int k = 0; k < a.n_body; ++k) {
                    float o[3];
                    body_end(0, k, 1, o);
                    const float d = fabsf(o[2] - fz);
                    if (d < bd) { bd = d; att = k; }
                }
            }
            s_attach[tid] = att;
        }
        __syncthreads();
        if (tid == 0) {
            s_attach[2] = a.attach[2] < 0 ? s_attach[1] : a.attach[2];
            s_attach[3] = a.attach[3] < 0 ? s_attach[0] : a.attach[3];
            a.nearest[0] = s_attach[0]; a.nearest[1] = s_attach[1];
        }
        __syncthreads();
    }
    if (tid == 0) a.ok[0] = a.n_leg > 0 ? s_ok : 1;
    for (int idx = tid; idx < N * K * 2; idx += EB_THREADS) {
        const int n = idx / (2 * K), r = idx - n * 2 * K, bone = r >> 1, e = r & 1;
        float o[3];
        if (bone < a.n_body) {
            body_end(n, bone, e, o);
        } else {  // leg bone i = (joint i + 1, joint i) of the ramp foot -> attachment   (skinning.py:196-215)
            const int l = (bone - a.n_body) / a.n_leg, i = (bone - a.n_body) - l * a.n_leg;
            const float* foot = pos + (long long)n * V * 3 + 3 * s_foot[n][l];
            float anchor[3];
            body_end(n, s_attach[l], 1, anchor);  // bones_pred[:, :, body_bone_idx, 1]
            const float rm = s_ramp[e == 0 ? i + 1 : i];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = foot[c] * (1.f - rm) + anchor[c] * rm;
        }
        float* out = a.bones + ((long long)n * K + bone) * 6 + 3 * e;
        out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
    }
    A3D_STAMP(0, 5);
}

extern "C" int a3d_estimate_bones(const a3d_estimate_bones_args* args, a3d_stream_t stream) {
    A3D_CHECK_ARG(args && args->size >= sizeof(a3d_estimate_bones_args));
    A3D_CHECK_ARG(args->pos && args->bones && args->nearest && args->ok && args->workspace && args->N > 0 && args->N <= EB_MAXN && args->V > 0);
    A3D_CHECK_ARG((long long)args->N * args->V <= (1 << 22) && args->n_body >= 2 && args->n_body % 2 == 0 && args->n_body <= 32 && args->n_leg >= 0 && args->n_leg <= 8);
    EbParams p;
    p.pos = args->pos; p.N = args->N; p.V = args->V; p.n_body = args->n_body; p.n_leg = args->n_leg; p.yplus = args->body_mode_y_plus;
    p.fauna = args->use_y_threshold; p.y_q = args->y_threshold;
    for (int i = 0; i < 17; ++i) p.blend[i] = args->blend[i];
    for (int i = 0; i < 9; ++i) p.ramp[i] = args->ramp[i];
    for (int i = 0; i < 4; ++i) p.attach[i] = args->attach[i];
    A3D_CHECK_ARG(args->n_leg == 0 || ((p.attach[0] < args->n_body && p.attach[1] < args->n_body && p.attach[2] < args->n_body && p.attach[3] < args->n_body)));
    p.bones = args->bones; p.nearest = args->nearest; p.ok = args->ok; p.work = args->workspace;
    hipLaunchKernelGGL(eb_kernel, dim3(1), dim3(EB_THREADS), 0, (hipStream_t)stream, p);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

A3D_PROFILE_TU(bones)
