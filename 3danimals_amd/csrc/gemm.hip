// fp32 MFMA GEMM with the ReLU adjoint in the epilogue, for the fields' hidden layers on gfx950:
//     C[M,N] = (A[M,K] . B[K,N]) * (X[M,N] > 0)          N = 256, K % 32 == 0, row-major, fp32 in / fp32 accumulate
// This is the input-gradient GEMM of a 256-wide Linear layer over a long point list (A = gradient of the layer's output,
// B = its weight [out,in]) whose input X is the ReLU output of the previous layer: masking the result with (X > 0) IS that
// previous layer's ReLU adjoint, so the separate threshold_backward pass over [M,256] (read 2, write 1 x 210 MB at M = 2e5, the
// largest non-GEMM item of the training step) disappears.  hipBLASLt has no DRELU epilogue and PyTorch cannot fuse it.
//
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles, 157 TFLOP/s chip peak).  A work-group of WM x WN waves computes a BM x BN tile,
// BM = 32 WM; each wave owns 32 rows x BN/WN columns as TN = BN/WN/32 accumulator tiles of 32x32.  K is walked in chunks of 32
// through LDS: A chunk [BM][32+1] (row-major, +1 pad: the operand read lane -> (row = lane&31, k = lane>>5) is conflict-free),
// B chunk [32][BN].  The next chunk's global loads are issued before the current chunk's MFMAs (register double buffering), the
// LDS operand reads of k-pair kk+2 before the MFMAs of k-pair kk; several work-groups per CU hide barriers and stores.
#include "a3d_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // (a native vector: arrays of HIP's float4 struct are not promoted to registers)

constexpr int GM_N = 256, GM_BK = 32, GM_APAD = GM_BK + 1;

template <int WM, int WN, int BN, int OCC, bool MASK>
__global__ __launch_bounds__(64 * WM * WN, OCC) void gm_nn_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                  const float* __restrict__ X, int M, int K, float* __restrict__ C) {
    constexpr int BM = 32 * WM, NT = 64 * WM * WN, TN = BN / WN / 32;
    constexpr int A_F4 = BM * GM_BK / 4 / NT;  // float4 per thread per chunk (A)
    constexpr int B_F4 = GM_BK * BN / 4 / NT;  // float4 per thread per chunk (B)
    static_assert(A_F4 >= 1 && B_F4 >= 1 && BM * GM_BK / 4 % NT == 0 && GM_BK * BN / 4 % NT == 0, "tile / thread count mismatch");
    constexpr int A_ROWS_PER_PASS = NT / 8;         // 8 float4 per A row
    constexpr int B_ROWS_PER_PASS = NT / (BN / 4);  // BN/4 float4 per B row
    __shared__ float As[BM * GM_APAD];
    __shared__ float Bs[GM_BK * BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave - wm * WN;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    static_assert(!MASK || TN == 4, "the masked path keeps the signs of its 64 outputs per lane in one 64-bit word");
    unsigned long long xmask = 0ull;
    const int col0 = n0 + wn * (BN / WN) + (lane & 31);
    const long long mrow = m0 + 32 * wm + 4 * (lane >> 5);
    // rows past M are never stored; clamp the mask reads of a ragged last tile to the last full 32-row group
    const float* x_lane = MASK ? X + col0 : nullptr;  // + X-row * GM_N (rows clamped to M - 1: outputs past M are never stored)
    const int a_row = tid >> 3, a_kq = tid & 7, b_row = tid / (BN / 4), b_nq = tid - b_row * (BN / 4);
    const float* a_src = A + (m0 + a_row) * K + 4 * a_kq;
    const float* b_src = B + (long long)b_row * GM_N + n0 + 4 * b_nq;
    f32x4 ra[A_F4], rb[B_F4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < A_F4; ++i)
        ra[i] = m0 + a_row + A_ROWS_PER_PASS * i < M ? *reinterpret_cast<const f32x4*>(a_src + (long long)A_ROWS_PER_PASS * i * K) : zero4;
#pragma unroll
    for (int i = 0; i < B_F4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(b_src + (long long)B_ROWS_PER_PASS * i * GM_N);

    for (int k0 = 0; k0 < K; k0 += GM_BK) {
        __syncthreads();  // everyone is done reading the previous chunk
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            float* d = As + (a_row + A_ROWS_PER_PASS * i) * GM_APAD + 4 * a_kq;
            d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) *reinterpret_cast<f32x4*>(Bs + (b_row + B_ROWS_PER_PASS * i) * BN + 4 * b_nq) = rb[i];
        __syncthreads();
        {  // next chunk (the last iteration re-reads chunk 0, unused): in flight while the MFMAs below run
            const int kn = k0 + GM_BK < K ? k0 + GM_BK : 0;
            const float* ap = a_src + kn;
            const float* bp = b_src + (long long)kn * GM_N;
#pragma unroll
            for (int i = 0; i < A_F4; ++i)
                ra[i] = m0 + a_row + A_ROWS_PER_PASS * i < M ? *reinterpret_cast<const f32x4*>(ap + (long long)A_ROWS_PER_PASS * i * K) : zero4;
#pragma unroll
            for (int i = 0; i < B_F4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp + (long long)B_ROWS_PER_PASS * i * GM_N);
        }
        // ReLU-adjoint mask, prefetched under the MFMAs: this lane owns 64 outputs (tile j, register r = 4q + s -> row s + 8q); chunk
        // c of the K loop loads the 8 X values (j = c>>1, q in {2(c&1), 2(c&1)+1}, s = 0..3) -- two base addresses, immediate row
        // offsets -- and folds their signs into byte c of a 64-bit mask at the end of the iteration (bit 16j + r)
        float xv[MASK ? 8 : 1];
        if (MASK) {
            const int c = (k0 / GM_BK) & 7;  // (chunks past the 8th re-read the first ones, harmless)
            const float* xb = x_lane + 32 * (c >> 1);
            const long long r0 = mrow + 16 * (c & 1);
#pragma unroll
            for (int dq = 0; dq < 2; ++dq)
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) {
                    const long long rr = r0 + 8 * dq + sx;
                    xv[4 * dq + sx] = xb[(rr < M ? rr : M - 1) * GM_N];
                }
        }
        const float* a_base = As + (32 * wm + (lane & 31)) * GM_APAD + (lane >> 5);
        const float* b_base = Bs + (lane >> 5) * BN + wn * (BN / WN) + (lane & 31);
        float av_n = a_base[0], bv_n[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv_n[j] = b_base[32 * j];
#pragma unroll
        for (int kk = 0; kk < GM_BK; kk += 2) {
            const float av = av_n;
            float bv[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = bv_n[j];
            if (kk + 2 < GM_BK) {
                av_n = a_base[kk + 2];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv_n[j] = b_base[(kk + 2) * BN + 32 * j];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the reads above the MFMAs (the scheduler would sink them to save registers)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MASK) {
            unsigned byte = 0u;
#pragma unroll
            for (int i = 0; i < 8; ++i) byte |= (unsigned)(xv[i] > 0.f) << i;
            xmask |= (unsigned long long)byte << (8 * ((k0 / GM_BK) & 7));
        }
    }
    // epilogue: accumulator element (tile j, register r) of lane l is row 32 wm + (r&3) + 8 (r>>2) + 4 (l>>5), column 32 j + (l&31)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long m = mrow + (r & 3) + 8 * (r >> 2);
        if (m < M) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v = acc[j][r];
                if (MASK) v = (xmask >> (16 * j + r)) & 1ull ? v : 0.f;
                C[m * GM_N + 32 * j + col0] = v;
            }
        }
    }
}


// ---- persistent variant: the weight half lives in LDS, A goes straight to registers, no barrier in the main loop ------------------
// One work-group per CU (16 waves = 4 per SIMD) keeps W[:, n0:n0+128] (256 x 128 floats = 128 KB of the CU's 160 KB LDS) for its
// whole life and walks 32-row tiles of A.  A never touches LDS: for the 32x32x2 MFMA lane l = (row l&31, half kh = l>>5) needs one
// value per k-pair, and since the contraction order is free the pairing is chosen so that a lane's 16-byte global load is exactly its
// next four operands: group g of 8 k -> lane loads A[row][8g + 4kh .. +3]; k-pair u of the group multiplies A's k = 8g + u (kh 0) /
// 8g + 4 + u (kh 1) with the same rows of W.  Every byte of A is loaded once, by one lane.  B operands are ds_read_b32 (conflict
// free).  Tiles are dealt so that every SIMD gets 12 or 13 of the 12.5 average (its four waves share one MFMA pipe).
constexpr int G3_BN = 128, G3_WAVES = 16;

template <bool MASK>
__global__ __launch_bounds__(64 * G3_WAVES, 1) void gm_nn3_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                  const float* __restrict__ X, int M, float* __restrict__ C) {
    constexpr int K = 256, TN = 4;
    extern __shared__ float Bs[];  // [K][G3_BN], then 16 waves x 64 sign words (512 B each)
    unsigned long long* sign_w = reinterpret_cast<unsigned long long*>(Bs + 256 * G3_BN) + 64 * (threadIdx.x >> 6);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = blockIdx.x & 1, bid = blockIdx.x >> 1, nb = gridDim.x >> 1;
    const int n0 = half * G3_BN;
    // W half -> LDS: K*G3_BN/4 = 8192 float4, 1024 threads -> 8 each (rows of 32 float4)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = tid + 1024 * i, k = f >> 5, q = f & 31;
        *reinterpret_cast<f32x4*>(Bs + k * G3_BN + 4 * q) = *reinterpret_cast<const f32x4*>(B + (long long)k * GM_N + n0 + 4 * q);
    }
    __syncthreads();

    const int row = lane & 31, kh = lane >> 5;
    const int n_tiles = (M + 31) >> 5;
    const int stride = nb * G3_WAVES;
    const float* b_lane = Bs + 4 * kh * G3_BN + row;  // + (8g + u) * G3_BN + 32 j
    for (int t = (wave >> 2) * (nb * 4) + bid * 4 + (wave & 3); t < n_tiles; t += stride) {
        const long long m0 = 32ll * t;
        const long long mr = m0 + row < M ? m0 + row : M - 1;  // ragged last tile: clamp the loads, skip the stores
        const float* a_lane = A + mr * K + 4 * kh;
        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const long long mrow = m0 + 4 * kh;
        const float* x_wave = MASK ? X + n0 + 4 * row : nullptr;  // + X-row * GM_N

        f32x4 an[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const f32x4*>(a_lane + 8 * i);
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {  // 8 chunks of 32 k = 4 groups of 8
            f32x4 av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = an[i];
            {
                const int cn = c + 1 < 8 ? c + 1 : 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const f32x4*>(a_lane + 32 * cn + 8 * i);
            }
            f32x4 xw[MASK ? 2 : 1];
            if (MASK) {  // 2 of the tile's 16 row pairs per chunk: lane -> row 2 i + (l>>5), columns 4 (l&31) .. +3 (512 B per row)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    long long rr = m0 + 2 * (2 * c + q) + kh;
                    rr = rr < M ? rr : M - 1;  // rows past M are never stored
                    xw[q] = *reinterpret_cast<const f32x4*>(x_wave + rr * GM_N);
                }
            }
            const float* bc = b_lane + (32 * c) * G3_BN;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* bp = bc + (8 * g + u) * G3_BN;
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][u], bp[32 * j], acc[j], 0, 0, 0);
                }
            if (MASK) {  // sign ballots of the two row pairs -> this wave's 16 x 4 words of LDS (word [i][k]: bit l = sign of lane l's k-th value)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const unsigned long long bal = __ballot(xw[q][k4] > 0.f);
                        if (lane == 0) sign_w[(2 * c + q) * 4 + k4] = bal;
                    }
            }
        }
        if (MASK) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the wave's own LDS writes above are read below
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = mrow + (r & 3) + 8 * (r >> 2);
            // mask word for this register: X row 4 kh + (r&3) + 8 (r>>2) = pair i = 2 kh + ((r&3)>>1) + 4 (r>>2), half r&1; column row + 32 j
            // = float4 (row>>2) + 8 j, component row&3  ->  bit (r&1)*32 + (row>>2) + 8 j of word [i][row&3]
            unsigned long long wv = 0ull;
            if (MASK) wv = sign_w[(2 * kh + ((r & 3) >> 1) + 4 * (r >> 2)) * 4 + (row & 3)] >> ((r & 1) * 32 + (row >> 2));
            if (m < M) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v = acc[j][r];
                    if (MASK) v = (wv >> (8 * j)) & 1ull ? v : 0.f;
                    C[m * GM_N + n0 + 32 * j + row] = v;
                }
            }
        }
    }
}

}  // namespace

extern "C" int a3d_gemm_nn_relumask(const float* A, const float* B, const float* X, int64_t M, int N, int K, float* C, a3d_stream_t stream) {
    A3D_CHECK_ARG(M >= 0 && N == GM_N && K > 0 && K % GM_BK == 0 && M < 0x7fffffffll);
    A3D_CHECK_ARG(X == nullptr || K >= 256);  // the mask is gathered over the first 8 K chunks
    if (M == 0) return A3D_OK;
    A3D_CHECK_ARG(A && B && C);
    hipStream_t s = (hipStream_t)stream;
#ifdef A3D_EXPERIMENT
    static const int variant = getenv("A3D_GEMM_VARIANT") ? atoi(getenv("A3D_GEMM_VARIANT")) : 3;  // experiment knob (build.py --exp only); 3 = persistent
#else
    const int variant = 3;  // persistent (the product library reads no environment)
#endif
    const dim3 block256(256);
    if (variant == 3 && K == 256) {  // persistent, weight half resident in LDS
        // per device ordinal (a process may drive more than one GPU); idempotent values, so a race between threads is harmless
        static int n_cu_of[64] = {0};
        int dev = 0;
        A3D_HIP(hipGetDevice(&dev));
        A3D_CHECK_ARG(dev >= 0 && dev < 64);
        if (!n_cu_of[dev]) {
            int n = 0;
            A3D_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
            A3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gm_nn3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (256 * G3_BN * 4 + 16 * 64 * 8)));
            A3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gm_nn3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (256 * G3_BN * 4 + 16 * 64 * 8)));
            n_cu_of[dev] = n;
        }
        const int n_cu = n_cu_of[dev];
        const dim3 grid(n_cu & ~1), block(64 * G3_WAVES);
        if (X) hipLaunchKernelGGL(gm_nn3_kernel<true>, grid, block, (256 * G3_BN * 4 + 16 * 64 * 8), s, A, B, X, (int)M, C);
        else hipLaunchKernelGGL(gm_nn3_kernel<false>, grid, block, (256 * G3_BN * 4 + 16 * 64 * 8), s, A, B, X, (int)M, C);
        A3D_LAUNCH_CHECK();
        return A3D_OK;
    }
    if (variant == 1) {  // 64 x 256 tile, wave 32 x 128, 3 work-groups per CU
        const dim3 grid(a3d_div_up(M, 64), 1);
        if (X) hipLaunchKernelGGL((gm_nn_kernel<2, 2, 256, 3, true>), grid, block256, 0, s, A, B, X, (int)M, K, C);
        else hipLaunchKernelGGL((gm_nn_kernel<2, 2, 256, 3, false>), grid, block256, 0, s, A, B, X, (int)M, K, C);
    } else {  // 128 x 128 tile, wave 32 x 128, 4 work-groups per CU
        const dim3 grid(a3d_div_up(M, 128), 2);
        if (X) hipLaunchKernelGGL((gm_nn_kernel<4, 1, 128, 3, true>), grid, block256, 0, s, A, B, X, (int)M, K, C);  // +16 VGPRs for the mask
        else hipLaunchKernelGGL((gm_nn_kernel<4, 1, 128, 4, false>), grid, block256, 0, s, A, B, X, (int)M, K, C);
    }
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
