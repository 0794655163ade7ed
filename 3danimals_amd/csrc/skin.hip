// Linear-blend skinning on gfx950: one thread per (image, vertex).
//
// Replaces the per-vertex part of skinning() (model/geometry/skinning.py:377, :419-431): softmax-over-bones
// weights from point-to-segment distances (skinning.py:16-22, geometry/util.py:30-53) and the weighted sum
// of the K per-bone affine maps.  The reference materialises K copies of the [B,V,3] vertex array; here a
// vertex is read once (12 B), the image's bones (K x 7 floats) and transforms (K x 12 floats) sit in LDS,
// and the softmax runs in registers over the K cached logits.  Weights are recomputed in backward instead of stored.
// HBM traffic: 12 B/vertex in (shared prior: once per image from L2), 12 B/vertex out.
#include "a3d_common.h"
#include "bones_common.h"

#define SK_THREADS 256
#define SK_MAXK 64

// A3D_STAMP (a3d_common.h; tools/skin_phases, tools/kernel_phases.py): kernel ids of this file: 0 = sk_fwd_kernel, 1 = sk_bwd_kernel.

struct SkBone {
    float ax, ay, az, dx, dy, dz, inv_len2;
};

__device__ __forceinline__ void sk_stage_bones(const float* __restrict__ bones, int K, SkBone* s_bone) {
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float* b = bones + 6 * k;
        SkBone sb;
        sb.ax = b[0]; sb.ay = b[1]; sb.az = b[2];
        sb.dx = b[3] - b[0]; sb.dy = b[4] - b[1]; sb.dz = b[5] - b[2];
        float l2 = sb.dx * sb.dx + sb.dy * sb.dy + sb.dz * sb.dz;
        sb.inv_len2 = 1.f / fmaxf(l2, 1e-6f);  // geometry/util.py:41
        s_bone[k] = sb;
    }
}

__device__ __forceinline__ void sk_stage(const float* __restrict__ bones, const float* __restrict__ T, int K, SkBone* s_bone,
                                         float* s_T) {
    sk_stage_bones(bones, K, s_bone);
    for (int i = threadIdx.x; i < K * 12; i += blockDim.x) s_T[i] = T[i];
}

// -dist(p, segment_k) / temperature   (geometry/util.py:41-52, skinning.py:21)
__device__ __forceinline__ float sk_logit(const SkBone& b, float px, float py, float pz, float neg_inv_temp) {
    float rx = px - b.ax, ry = py - b.ay, rz = pz - b.az;
    float t = (rx * b.dx + ry * b.dy + rz * b.dz) * b.inv_len2;
    t = fminf(fmaxf(t, 0.f), 1.f);
    float sx = t * b.dx - rx, sy = t * b.dy - ry, sz = t * b.dz - rz;
    // (v_sqrt_f32, 1 ulp: the argument is >= 1e-6, far from the denormals whose handling makes the correctly rounded sqrtf this file is
    // otherwise compiled with ~20 instructions -- of the ~50 a logit took)
    return __builtin_amdgcn_sqrtf(sx * sx + sy * sy + sz * sz + 1e-6f) * neg_inv_temp;
}

// KMAX = compile-time bound on K: the K logits of a vertex (a sqrt each) are computed ONCE and stay in registers for the
// max / sum / blend passes (the first version recomputed them per pass: 3 sqrt chains per bone per vertex).
// POSE (a3d_skin_pose_fwd): ``T`` is an OUTPUT -- every work-group composes the K chain transforms of its image itself (links built
// once into LDS, every bone multiplies the <= 8 links of its chain: the work of bones.hip's bn_fwd_kernel, ~1 us per work-group, all
// work-groups at once) instead of reading the result of a separate launch; the first work-group of an image also writes them out for
// the backward and for posed_bones.  ``angles`` [B,K,3], ``chain`` [K,D]; ``clear`` = the backward's angle-gradient accumulator.
// EXACT: K == KMAX -- no bone guards.  With them every logit sits in a basic block of its own (``k < K ? f(k) : -inf`` becomes a branch
// around f), and five dependent chains that could interleave run one after the other.
template <int KMAX, bool POSE, bool EXACT = false>
__global__ __launch_bounds__(SK_THREADS) void sk_fwd_kernel(const float* __restrict__ v, int v_batch, const float* __restrict__ bones,
                                                            int bones_batch, float* T, int V, int K,
                                                            float neg_inv_temp, float* __restrict__ out, float* __restrict__ weights,
                                                            float* __restrict__ clear, int n_clear, const float* __restrict__ angles,
                                                            const int* __restrict__ chain, int D, float* __restrict__ PS, int groups) {
    __shared__ SkBone s_bone[SK_MAXK];
    __shared__ float s_T[SK_MAXK * 12];
    const int b = blockIdx.y;
    A3D_STAMP(0, 0);
    A3D_STAMP_CLOCK(0, 6);
    // the backward's per-image transform gradient (accumulated there with atomics) cleared here: one memset less on the backward path
    for (int z = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; z < n_clear; z += gridDim.x * gridDim.y * blockDim.x) clear[z] = 0.f;
    __shared__ float s_L[POSE ? SK_MAXK : 1][13];
    __shared__ int s_chain[POSE ? SK_MAXK * BN_MAXD : 1];
    const float* bb = bones + (bones_batch == 1 ? 0ll : (long long)b * K * 6);
    // FOUR lanes per vertex (a quad), each with a quarter of the bones: the K logits are a sqrt + an exp each and a thread that does all
    // twenty is ~2000 dependent instructions with 1.5 waves per SIMD to hide them behind (the kernel took 9 us for 1 MB); four times
    // the waves with a quarter of the chain each, and the quad meets through DPP (max, then the four sums)
    const int sub = threadIdx.x & 3;
    // the first vertex of this thread, requested BEFORE the staging below: its round trip runs under the staging's own
    int i = blockIdx.x * groups * (SK_THREADS / 4) + (threadIdx.x >> 2);
    bool valid = i < V;
    const float* vb = v + (v_batch == 1 ? 0ll : (long long)b * V) * 3;
    float px = vb[3ll * (valid ? i : 0)], py = vb[3ll * (valid ? i : 0) + 1], pz = vb[3ll * (valid ? i : 0) + 2];
    const bool products_wg = POSE && PS && blockIdx.x == gridDim.x - 1;  // (one extra work-group per image, no vertices: see below)
    if (POSE) {
        const float* aa = angles + (long long)b * K * 3;
        for (int w = threadIdx.x; w < K * D; w += blockDim.x) s_chain[w] = chain[w];
        // the K links, THREE lanes per bone (K <= 20: inside the first wave): every lane of a triple takes the sine / cosine of one of
        // the bone's angles (one thread per bone ran six of them back to back: 1.7 of the 2.2 us this stage took), the triple meets
        // through shuffles, its first lane composes the link.  In the products work-group lane c also leaves d link / d angle_c for
        // the backward (from the same sines and cosines: no second pass over the angles).
        if ((int)threadIdx.x < 3 * K) {
            const int li = threadIdx.x / 3, lc = threadIdx.x - 3 * li;
            float sn, cs;
            sincosf(aa[3 * li + lc], &sn, &cs);
            float R[9], t[3];
            bn_rest(bb + 6 * li, R, t);
            BnTrig g;
            g.sx = __shfl(sn, 3 * li); g.cx = __shfl(cs, 3 * li);
            g.sy = __shfl(sn, 3 * li + 1); g.cy = __shfl(cs, 3 * li + 1);
            g.sz = __shfl(sn, 3 * li + 2); g.cz = __shfl(cs, 3 * li + 2);
            if (lc == 0) bn_store(s_L[li], bn_link_from(R, t, g));
            if (products_wg) bn_link_derivative_from(R, t, g, lc, PS + (long long)b * bn_products_floats(K, D) + (long long)K * D * 24 + 36 * li + 12 * lc);
        }
        sk_stage_bones(bb, K, s_bone);
    } else {
        sk_stage(bb, T + (long long)b * K * 12, K, s_bone, s_T);
    }
    __syncthreads();
    A3D_STAMP(0, 1);
    if (POSE) {
        // one EXTRA work-group per image (the last one; no vertices) leaves the prefix / suffix products of every chain position for the
        // backward (bn_chain_adjoint_ps).  Inside a vertex work-group those 2K serial product chains sat in front of a barrier all
        // 256 threads wait at, and that work-group was the slowest of its image
        if (products_wg) {
            bn_chain_products(s_L, s_chain, K, D, PS + (long long)b * bn_products_floats(K, D));
            A3D_STAMP(0, 5);
            return;
        }
        // the K chain products, FOUR lanes per bone (lane c keeps column c of the running product: 3 values; the other columns come
        // from the quad through DPP), the chain's link columns fetched up front -- 8 steps of 9 DPP + 9 multiply-adds instead of 8
        // steps of two dependent LDS round trips + 36 multiply-adds in one thread (1.9 of a work-group's ~5 us)
        if ((int)threadIdx.x < 4 * K) {
            const int k = threadIdx.x >> 2, c = threadIdx.x & 3;
            float Lc[BN_MAXD][3];
            bool on[BN_MAXD];
#pragma unroll
            for (int j = 0; j < BN_MAXD; ++j) {
                const int ii = j < D ? s_chain[k * D + j] : -1;
                on[j] = ii >= 0;
                const float* l = s_L[on[j] ? ii : 0];
                Lc[j][0] = on[j] ? l[c] : (c == 0 ? 1.f : 0.f);
                Lc[j][1] = on[j] ? l[4 + c] : (c == 1 ? 1.f : 0.f);
                Lc[j][2] = on[j] ? l[8 + c] : (c == 2 ? 1.f : 0.f);
            }
            float r0 = c == 0 ? 1.f : 0.f, r1 = c == 1 ? 1.f : 0.f, r2 = c == 2 ? 1.f : 0.f;  // column c of the identity
            auto quad = [](float x, int q) {
                const int xi = __float_as_int(x);
                return __int_as_float(q == 0 ? __builtin_amdgcn_mov_dpp(xi, 0x00, 0xF, 0xF, true)
                                             : (q == 1 ? __builtin_amdgcn_mov_dpp(xi, 0x55, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(xi, 0xAA, 0xF, 0xF, true)));
            };
#pragma unroll
            for (int j = 0; j < BN_MAXD; ++j) {
                {   // (padded positions multiply by the identity: no branch per step)
                    const float n0 = quad(r0, 0) * Lc[j][0] + quad(r0, 1) * Lc[j][1] + quad(r0, 2) * Lc[j][2] + (c == 3 ? r0 : 0.f);
                    const float n1 = quad(r1, 0) * Lc[j][0] + quad(r1, 1) * Lc[j][1] + quad(r1, 2) * Lc[j][2] + (c == 3 ? r1 : 0.f);
                    const float n2 = quad(r2, 0) * Lc[j][0] + quad(r2, 1) * Lc[j][1] + quad(r2, 2) * Lc[j][2] + (c == 3 ? r2 : 0.f);
                    r0 = n0; r1 = n1; r2 = n2;
                }
            }
            s_T[12 * k + c] = r0; s_T[12 * k + 4 + c] = r1; s_T[12 * k + 8 + c] = r2;
            if (blockIdx.x == 0) {
                float* To = T + ((long long)b * K + k) * 12;
                To[c] = r0; To[4 + c] = r1; To[8 + c] = r2;
            }
        }
        A3D_STAMP(0, 2);
    }
    // ``groups`` consecutive 64-vertex groups per work-group: 1 at the bench size; more for large meshes, where a work-group per 64
    // vertices would repeat the chain composition above thousands of times (R = 128 grid: 374 work-groups per image)
    for (int gi = 0; gi < groups; ++gi) {
    if (gi > 0) {
        i = (blockIdx.x * groups + gi) * (SK_THREADS / 4) + (threadIdx.x >> 2);
        if (i - (int)(threadIdx.x >> 2) >= V) break;  // (uniform: the whole group is past the end)
        valid = i < V;
        const float* p = vb + 3ll * (valid ? i : 0);
        px = p[0]; py = p[1]; pz = p[2];
    }
    constexpr int KQ = (KMAX + 3) / 4;
    float lg[KQ];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const int k = 4 * q + sub;
        lg[q] = (EXACT || k < K) ? sk_logit(s_bone[(EXACT || k < K) ? k : 0], px, py, pz, neg_inv_temp) : -INFINITY;
        m = fmaxf(m, lg[q]);
    }
    m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0xB1, 0xF, 0xF, true)));  // quad_perm [1,0,3,2]
    m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0x4E, 0xF, 0xF, true)));  // quad_perm [2,3,0,1]
    // (the logits of the first group only need the bones: they run beside the chain products above, and the one barrier both wait at is here)
    if (POSE && gi == 0) {
        __syncthreads();  // s_T complete (the blend below only reads it)
        A3D_STAMP(0, 3);
    }
    float s = 0.f, ox = 0.f, oy = 0.f, oz = 0.f;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const int k = 4 * q + sub;
        if (EXACT || k < K) {
            const float e = __expf(lg[q] - m);
            lg[q] = e;
            const float* t = s_T + 12 * k;
            s += e;
            ox += e * (t[0] * px + t[1] * py + t[2] * pz + t[3]);
            oy += e * (t[4] * px + t[5] * py + t[6] * pz + t[7]);
            oz += e * (t[8] * px + t[9] * py + t[10] * pz + t[11]);
        }
    }
    auto quad_sum = [](float r) {
        r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0xB1, 0xF, 0xF, true));
        r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x4E, 0xF, 0xF, true));
        return r;
    };
    s = quad_sum(s); ox = quad_sum(ox); oy = quad_sum(oy); oz = quad_sum(oz);
    if (!valid) continue;
    const float inv = 1.f / s;
    if (sub < 3) out[((long long)b * V + i) * 3 + sub] = (sub == 0 ? ox : (sub == 1 ? oy : oz)) * inv;
    if (weights) {  // [K, Bw, V]; only images that own distinct weights write
        const int Bw = (v_batch == 1 && bones_batch == 1) ? 1 : (int)gridDim.y;
        if (b < Bw) {
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const int k = 4 * q + sub;
                if (EXACT || k < K) weights[((long long)k * Bw + b) * V + i] = lg[q] * inv;
            }
        }
    }
    }
    A3D_STAMP(0, 5);
    A3D_STAMP_CLOCK(0, 7);
}

// ---- backward, one kernel, two phases per 256-vertex chunk.
// phase 1 (thread = vertex): softmax weights once (into LDS, [K][256]) and g_v = sum_k w_k R_k^T g (written per image; a shared
//          canonical mesh is summed over the batch by the caller).
// phase 2 (wave = 64 vertices of the chunk): g_T[b,k] = sum_v w_k(v) * g(v) (x) [v,1] -- a [K x V].[V x 12] product per image, on the matrix
//          pipe (fp32 MFMA 16x16x4: exact fp32 products and sums); the four waves' tiles meet in LDS once per work-group.
// Both phases are bound by the instructions they issue, not by memory (tools/skin_phases): 20 logits + the blend are ~1000 wave
// instructions per chunk and wave, and a CU that holds two work-groups takes twice as long over them.
// POSE (a3d_skin_pose_bwd): the adjoint of the chain composition (bones_common.h: bn_chain_adjoint_ps, the work of bones.hip's
// bn_bwd_kernel) runs in this launch too.  It is LINEAR in the transform gradient, so every work-group applies it to its own share of
// g_T[b] -- straight from LDS, never written out -- and adds the resulting K x 3 angle gradients to g_angles[b] (zero on entry) with
// fire-and-forget atomics: sum over work-groups of adjoint(share) = adjoint(sum).  No work-group waits for another.  (Until the middle
// of round 3 the shares met in global memory through K*12 returning atomics per work-group and the work-group that took an image's
// last ticket ran the adjoint once, alone: 14 us of parallel work + 9 us of serial tail; before that, a launch of its own.)
template <int KG, bool POSE, bool EXACT = false>  // EXACT: K == 4 * KG (no bone guards, see sk_fwd_kernel)
__global__ __launch_bounds__(SK_THREADS) void sk_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ v, int v_batch,
                                                            const float* __restrict__ bones, int bones_batch, const float* __restrict__ T,
                                                            int V, int K, float neg_inv_temp, int verts_per_wg, float* __restrict__ g_v,
                                                            float* g_T, const float* __restrict__ angles, const int* __restrict__ chain, int D,
                                                            const float* __restrict__ g_T_extra, float* __restrict__ g_angles,
                                                            const float* __restrict__ PS) {
    __shared__ SkBone s_bone[SK_MAXK];
    __shared__ float s_T[SK_MAXK * 12];
    // (rows 260 apart: the matrix operands below are read bone-major / component-major for a fixed vertex, and with a stride of 256
    // every row of a read would sit in the same bank)
    __shared__ float s_w[4 * KG][SK_THREADS + 4];
    __shared__ float s_x[6][SK_THREADS + 4];  // px py pz gx gy gz
    const int b = blockIdx.y;
    A3D_STAMP(1, 0);
    A3D_STAMP_CLOCK(1, 6);
    const float* vb = v + (v_batch == 1 ? 0ll : (long long)b * V * 3);
    const float* gb = g_out + (long long)b * V * 3;
    // this work-group's vertices [lo, hi), 256 at a time (the last trip may be partial)
    const int lo = blockIdx.x * verts_per_wg, hi = min(lo + verts_per_wg, V);
    // requested before the staging: the first chunk's vertex and gradient, and (POSE) the chain entry the adjoint at the very end hangs its
    // loads on -- round trips that would otherwise each stand alone in front of dependent work
    float px = 0.f, py = 0.f, pz = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    if (lo + (int)threadIdx.x < hi) {
        const long long i = lo + threadIdx.x;
        px = vb[3 * i]; py = vb[3 * i + 1]; pz = vb[3 * i + 2];
        gx = gb[3 * i]; gy = gb[3 * i + 1]; gz = gb[3 * i + 2];
    }
    const int my_chain = (POSE && (int)threadIdx.x < K * D) ? chain[threadIdx.x] : -2;
    sk_stage(bones + (bones_batch == 1 ? 0ll : (long long)b * K * 6), T + (long long)b * K * 12, K, s_bone, s_T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // phase 2 on the matrix pipe: g_T[b] (K x 12) = W^T (K x V) . X (V x 12), X(v) = g(v) (x) [p(v), 1] -- v_mfma_f32_16x16x4_f32, a
    // 16-bone tile of W^T times 4 vertices of X per instruction, every wave over its own 64 vertices of the chunk.  The accumulators
    // are 4 registers per tile and lane and the sum over the lanes is the instruction's own; in the vector form every lane kept KG x 12
    // partial sums and the cross-lane reduction of those 60 values (4 DPP steps each + LDS) cost twice what the products did
    // (2.0 + 1.0 us of a work-group's 10)
    constexpr int NT = (4 * KG + 15) / 16;  // 16-bone tiles
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc[NT];
#pragma unroll
    for (int tI = 0; tI < NT; ++tI) acc[tI] = f4{0.f, 0.f, 0.f, 0.f};
    const int mi = lane & 15, mk = lane >> 4;  // A operand: bone mi of the tile, vertex mk of the four; B operand: vertex mk, column mi
    const int jc = 3 + (mi >> 2), jd = mi & 3;  // column mi = 4 c + d of X: g_c * (d < 3 ? p_d : 1)
    auto matrix_phase = [&]() {
        // (every lane reads a valid location unconditionally and the masks are applied to the loaded values: a load under a condition
        // becomes a branch with a wait of its own -- four serial LDS round trips in front of every instruction of the matrix pipe)
        const float* xg = &s_x[jc < 6 ? jc : 5][wave * 64 + mk];
        const float* xp = &s_x[jd < 3 ? jd : 2][wave * 64 + mk];
        const float* wa[NT];
        bool wa_on[NT];
#pragma unroll
        for (int tI = 0; tI < NT; ++tI) {
            const int k = 16 * tI + mi;
            wa_on[tI] = k < 4 * KG && (EXACT || k < K);
            wa[tI] = &s_w[k < 4 * KG ? k : 4 * KG - 1][wave * 64 + mk];
        }
#pragma unroll 8
        for (int st = 0; st < 16; ++st) {
            const float gv = xg[4 * st], pv = xp[4 * st];
            float av[NT];
#pragma unroll
            for (int tI = 0; tI < NT; ++tI) av[tI] = wa[tI][4 * st];
            const float x = mi < 12 ? gv * (jd < 3 ? pv : 1.f) : 0.f;
#pragma unroll
            for (int tI = 0; tI < NT; ++tI) acc[tI] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_on[tI] ? av[tI] : 0.f, x, acc[tI], 0, 0, 0);
        }
    };
    for (int base = lo; base < hi; base += SK_THREADS) {
        __syncthreads();       // previous chunk's LDS fully consumed (also orders sk_stage on the first trip)
        {
            const int i = base + threadIdx.x;
            const bool ok = i < hi;
            if (base > lo) {
                px = py = pz = gx = gy = gz = 0.f;
                if (ok) {
                    px = vb[3ll * i]; py = vb[3ll * i + 1]; pz = vb[3ll * i + 2];
                    gx = gb[3ll * i]; gy = gb[3ll * i + 1]; gz = gb[3ll * i + 2];
                }
            }
            float lg[4 * KG];  // the logits (a sqrt each) once, in registers
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4 * KG; ++k) {
                lg[k] = (EXACT || k < K) ? sk_logit(s_bone[k], px, py, pz, neg_inv_temp) : -INFINITY;
                m = fmaxf(m, lg[k]);
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4 * KG; ++k) {
                lg[k] = (EXACT || k < K) ? __expf(lg[k] - m) : 0.f;
                s += lg[k];
            }
            const float inv = 1.f / s;
            float ox = 0.f, oy = 0.f, oz = 0.f;
#pragma unroll
            for (int k = 0; k < 4 * KG; ++k) {
                if (EXACT || k < K) {
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
        A3D_STAMP(1, 1);
        matrix_phase();
    }
    A3D_STAMP(1, 2);
    // the four waves' tiles meet in LDS (lane l, register r of a tile = bone 4 (l / 16) + r, column l % 16), and thread (bone, component)
    // adds the block's total to g_T: consecutive threads -> consecutive floats, i.e. the block's K*12 atomics are K*12/16 line requests
    // (line-coalesced device atomics are ~10x cheaper than scattered ones)
    __shared__ float s_red[SK_THREADS / 64][NT * 16][17];
#pragma unroll
    for (int tI = 0; tI < NT; ++tI)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[wave][16 * tI + 4 * mk + r][mi] = acc[tI][r];
    __syncthreads();
    A3D_STAMP(1, 3);
    __shared__ float s_gT[POSE ? 4 * KG * 12 : 1];  // POSE: this work-group's share of g_T[b]
    for (int t = threadIdx.x; t < K * 12; t += SK_THREADS) {
        const int k = t / 12, q = t - 12 * k;
        float val = s_red[0][k][q] + s_red[1][k][q] + s_red[2][k][q] + s_red[3][k][q];
        if (POSE) {
            // (a gradient that reaches the transforms directly -- posed_bones -- joins the first work-group's share)
            if (g_T_extra && blockIdx.x == 0) val += g_T_extra[((long long)b * K + k) * 12 + q];
            s_gT[t] = val;
        } else {
            atomicAdd(g_T + ((long long)b * K + k) * 12 + q, val);
        }
    }
    if (POSE) {
        __shared__ float s_adj[POSE ? 4 * 20 * BN_MAXD + 20 * 20 : 1];
        __syncthreads();
        A3D_STAMP(1, 4);
        const float* ps = PS + (long long)b * bn_products_floats(K, D);
        bn_chain_adjoint_ps(s_gT, ps, ps + (long long)K * D * 24, chain, K, D, g_angles + (long long)b * K * 3, s_adj, my_chain);
    }
    A3D_STAMP(1, 5);
    A3D_STAMP_CLOCK(1, 7);
}

extern "C" int a3d_skin_fwd(const float* v, int v_batch, const float* bones, int bones_batch, const float* T, int B, int V, int K,
                            float temperature, float* out, float* weights_or_null, float* g_T_to_clear_or_null, a3d_stream_t stream) {
    A3D_CHECK_ARG(v && bones && T && out);
    A3D_CHECK_ARG(B > 0 && V > 0 && K > 0 && K <= SK_MAXK && temperature > 0.f);
    A3D_CHECK_ARG((v_batch == 1 || v_batch == B) && (bones_batch == 1 || bones_batch == B));
    const int ngroups = a3d_div_up(V, SK_THREADS / 4), groups = a3d_div_up(ngroups, 128);  // four lanes per vertex; <= 128 work-groups per image
    const dim3 grid(a3d_div_up(ngroups, groups), B), block(SK_THREADS);
    hipStream_t s = (hipStream_t)stream;
    const float nit = -1.f / temperature;
    const int ncl = g_T_to_clear_or_null ? B * K * 12 : 0;
    float* Tm = const_cast<float*>(T);  // (read only without POSE)
    const float* no_angles = nullptr;
    const int* no_chain = nullptr;
    float* no_ps = nullptr;
    if (K <= 20) hipLaunchKernelGGL((sk_fwd_kernel<20, false>), grid, block, 0, s, v, v_batch, bones, bones_batch, Tm, V, K, nit, out, weights_or_null, g_T_to_clear_or_null, ncl, no_angles, no_chain, 0, no_ps, groups);
    else if (K <= 32) hipLaunchKernelGGL((sk_fwd_kernel<32, false>), grid, block, 0, s, v, v_batch, bones, bones_batch, Tm, V, K, nit, out, weights_or_null, g_T_to_clear_or_null, ncl, no_angles, no_chain, 0, no_ps, groups);
    else hipLaunchKernelGGL((sk_fwd_kernel<SK_MAXK, false>), grid, block, 0, s, v, v_batch, bones, bones_batch, Tm, V, K, nit, out, weights_or_null, g_T_to_clear_or_null, ncl, no_angles, no_chain, 0, no_ps, groups);
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
    cpb *= SK_THREADS;  // (the kernel takes vertices per work-group)
    const float* nf = nullptr;
    const int* ni = nullptr;
    float* ng = nullptr;
    if (K <= 20) hipLaunchKernelGGL((sk_bwd_kernel<5, false>), grid, block, 0, s, g_out, v, v_batch, bones, bones_batch, T, V, K, nit, cpb, g_v_or_null, g_T, nf, ni, 0, nf, ng, nf);
    else if (K <= 32) hipLaunchKernelGGL((sk_bwd_kernel<8, false>), grid, block, 0, s, g_out, v, v_batch, bones, bones_batch, T, V, K, nit, cpb, g_v_or_null, g_T, nf, ni, 0, nf, ng, nf);
    else hipLaunchKernelGGL((sk_bwd_kernel<16, false>), grid, block, 0, s, g_out, v, v_batch, bones, bones_batch, T, V, K, nit, cpb, g_v_or_null, g_T, nf, ni, 0, nf, ng, nf);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

// ---- kinematic chain + skinning in ONE launch each way (K <= 20 bones, chains of D <= 8 links: every configuration of the reference)
extern "C" int a3d_skin_pose_max_bones(void) { return 20; }
extern "C" size_t a3d_skin_pose_products_floats(int K, int D) { return bn_products_floats(K < 0 ? 0 : K, D < 0 ? 0 : D); }

extern "C" int a3d_skin_pose_fwd(const float* v, int v_batch, const float* bones, int bones_batch, const float* angles, const int32_t* chain,
                                 int B, int V, int K, int D, float temperature, float* out, float* T_out, float* chain_products_or_null,
                                 float* g_angles_to_clear_or_null, a3d_stream_t stream) {
    A3D_CHECK_ARG(v && bones && angles && chain && out && T_out);
    A3D_CHECK_ARG(B > 0 && V > 0 && K > 0 && K <= 20 && D > 0 && D <= BN_MAXD && temperature > 0.f);
    A3D_CHECK_ARG((v_batch == 1 || v_batch == B) && (bones_batch == 1 || bones_batch == B));
    A3D_CHECK_ARG(((uintptr_t)chain_products_or_null & 15) == 0);  // (its 3x4 blocks are written 16 bytes at a time; an image's share is a multiple of 16 bytes)
    // (g_angles[B,K,3] of the backward, which accumulates into it: cleared here, one memset less on the backward path)
    // 64-vertex groups (four lanes per vertex), as many per work-group as leave ~1024 work-groups in the launch: every work-group
    // composes its image's chains itself (~900 instructions in its first wave), and the launch is bound by the instructions it issues
    // -- B = 16: V = 6k one group per work-group 11.8 us, two 9.6, three 10.5; V = 24k three 23.6, six 22.6, twelve 27.0
    const int ngroups = a3d_div_up(V, SK_THREADS / 4);
    int groups = (int)a3d_div_up((long long)ngroups * B, 1024);
    if (groups < 1) groups = 1;
    const dim3 grid(a3d_div_up(ngroups, groups) + (chain_products_or_null ? 1 : 0), B), block(SK_THREADS);  // (+ the products work-group)
    const int ncl = g_angles_to_clear_or_null ? B * K * 3 : 0;
    float* no_w = nullptr;
    if (K == 20)  // (every configuration of the reference: 8 body + 4 x 3 leg bones)
        hipLaunchKernelGGL((sk_fwd_kernel<20, true, true>), grid, block, 0, (hipStream_t)stream, v, v_batch, bones, bones_batch, T_out, V, K,
                           -1.f / temperature, out, no_w, g_angles_to_clear_or_null, ncl, angles, chain, D, chain_products_or_null, groups);
    else
        hipLaunchKernelGGL((sk_fwd_kernel<20, true, false>), grid, block, 0, (hipStream_t)stream, v, v_batch, bones, bones_batch, T_out, V, K,
                           -1.f / temperature, out, no_w, g_angles_to_clear_or_null, ncl, angles, chain, D, chain_products_or_null, groups);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_skin_pose_bwd(const float* g_out, const float* v, int v_batch, const float* bones, int bones_batch, const float* T,
                                 const float* chain_products, const float* angles, const int32_t* chain, int B, int V, int K, int D,
                                 float temperature, float* g_v_or_null, const float* g_T_extra_or_null, float* g_angles,
                                 int g_angles_is_clear, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_out && v && bones && T && chain_products && angles && chain && g_angles);
    A3D_CHECK_ARG(B > 0 && V > 0 && K > 0 && K <= 20 && D > 0 && D <= BN_MAXD && temperature > 0.f);
    A3D_CHECK_ARG((v_batch == 1 || v_batch == B) && (bones_batch == 1 || bones_batch == B));
    hipStream_t s = (hipStream_t)stream;
    if (!g_angles_is_clear) A3D_HIP(hipMemsetAsync(g_angles, 0, sizeof(float) * (size_t)B * K * 3, s));
    // ~768 work-groups of whole 256-vertex chunks: every work-group pays a fixed part (bones + transforms into LDS, the K*12-value
    // cross-lane reduction, a chain adjoint of its own: 1.4 + 2.7 us) besides its chunks -- V = 24k, B = 16 (1504 chunks): one chunk
    // per work-group 41 us, two 29, three 33.5; V = 6k (384 chunks): one 16, two 21.  (Balancing the CUs instead -- 256 work-groups of
    // 1.5 chunks at V = 6k -- measured 20.5: the chunks of a work-group run one after the other and the kernel is bound by that chain.)
    const int chunks = a3d_div_up(V, SK_THREADS);
    int cpb = a3d_div_up((long long)chunks * B, 768);
    if (cpb < 1) cpb = 1;
    const int vpw = cpb * SK_THREADS;
    const dim3 grid(a3d_div_up(chunks, cpb), B), block(SK_THREADS);
    float* no_gT = nullptr;
    if (K == 20)
        hipLaunchKernelGGL((sk_bwd_kernel<5, true, true>), grid, block, 0, s, g_out, v, v_batch, bones, bones_batch, T, V, K, -1.f / temperature, vpw,
                           g_v_or_null, no_gT, angles, chain, D, g_T_extra_or_null, g_angles, chain_products);
    else
        hipLaunchKernelGGL((sk_bwd_kernel<5, true, false>), grid, block, 0, s, g_out, v, v_batch, bones, bones_batch, T, V, K, -1.f / temperature, vpw,
                           g_v_or_null, no_gT, angles, chain, D, g_T_extra_or_null, g_angles, chain_products);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

A3D_PROFILE_TU(skin)
