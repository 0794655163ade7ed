// out[img[p], :] += g[p, :]  -- the adjoint of broadcasting a per-image row (image feature, light parameters, camera rotation)
// to the covered pixels of that image (shade(): feat / w2c / light are [B,.] against [P,.] points, render.py:53-94).
// torch's index_add does P x C float atomics onto B rows (280 us for P=2e5, C=256).  The point list is sorted by image, so a
// block owns 128 consecutive rows, its threads keep partial sums in registers, and one atomic per (block, column) is issued.
// blockDim = 256 = RL row-lanes x CP column-lanes; a column-lane covers VEC (1 or 4) adjacent floats, so a 256-float row is read
// as 64 x 16 B by 4 row-lanes at a time, and the narrow per-image tensors (17 columns) still keep every lane busy.
#include "a3d_common.h"

#define SS_ROWS 128

// MASK: the summand is g * (y > 0) -- the ReLU adjoint -- and is also stored to g_pre (a3d_rows_add_relu_bwd)
template <int VEC, bool MASK, bool NT = false>
__global__ __launch_bounds__(256) void ss_kernel(const float* __restrict__ g, const long long* __restrict__ img, long long P, int C, int B,
                                                 int CP, float* __restrict__ out, const float* __restrict__ y, float* __restrict__ g_pre,
                                                 int rows_per_block) {
    __shared__ float s_part[256 * VEC];
    auto val = [&](long long i) -> float {
        float v = g[i];
        if (MASK) {
            v = y[i] > 0.f ? v : 0.f;
            g_pre[i] = v;
        }
        return v;
    };
    // VEC floats at element offset i (16-byte aligned when VEC == 4: C % 4 == 0 and torch allocations are 256 B aligned)
    auto add_group = [&](long long i, float* acc) {
        if constexpr (VEC == 4) {
            float4 v, t;
            if (NT) {  // streamed once: do not keep in the caches
                v = a3d_load_stream4(g + i);
                if (MASK) t = a3d_load_stream4(y + i);
            } else {
                v = *reinterpret_cast<const float4*>(g + i);
                if (MASK) t = *reinterpret_cast<const float4*>(y + i);
            }
            if (MASK) {
                v.x = t.x > 0.f ? v.x : 0.f; v.y = t.y > 0.f ? v.y : 0.f; v.z = t.z > 0.f ? v.z : 0.f; v.w = t.w > 0.f ? v.w : 0.f;
                *reinterpret_cast<float4*>(g_pre + i) = v;
            }
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += val(i + k);
        }
    };
    const int RL = 256 / CP;
    const int c = threadIdx.x % CP, rl = threadIdx.x / CP;
    const int CV = C / VEC;  // columns in units of VEC floats
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(r0 + (long long)rows_per_block, P);
    const long long first = img[r0], last = img[r1 - 1];
    for (int cbase = 0; cbase < CV; cbase += CP) {  // uniform trip count (barriers inside)
        const int cc = cbase + c;
        const bool active = cc < CV;
        // one pass per image present in the block: a single one, except in the <= B-1 blocks that straddle an image boundary of the sorted
        // list, which repeat the pass with the other images' rows skipped.  (Those blocks used to add row by row with atomics: up to
        // rows_per_block adds onto each of the C addresses, and same-address device atomics serialise at ~30 ns each.)
        for (long long bi = min(first, last); bi <= max(first, last); ++bi) {  // uniform bounds (min/max: a padded tail may wrap to image 0)
            const bool uniform = first == last;
            constexpr int NS = 4;  // independent row streams per thread (loads in flight)
            float acc[NS][VEC];
#pragma unroll
            for (int q = 0; q < NS; ++q)
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[q][k] = 0.f;
            if (active) {
                long long r = r0 + rl;
                if (uniform) {  // branch-free accumulation
                    for (; r + (long long)(NS - 1) * RL < r1; r += (long long)NS * RL) {
#pragma unroll
                        for (int q = 0; q < NS; ++q) add_group((r + (long long)q * RL) * C + (long long)cc * VEC, acc[q]);
                    }
                }
                for (; r < r1; r += RL)
                    if (uniform || img[r] == bi) add_group(r * C + (long long)cc * VEC, acc[0]);
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) s_part[threadIdx.x * VEC + k] = (acc[0][k] + acc[1][k]) + (acc[2][k] + acc[3][k]);
            __syncthreads();
            if (rl == 0 && active && (unsigned long long)bi < (unsigned long long)B) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float t = 0.f;
                    for (int q = 0; q < RL; ++q) t += s_part[(q * CP + c) * VEC + k];
                    if (uniform || t != 0.f) atomicAdd(out + bi * C + (long long)cc * VEC + k, t);
                }
            }
            __syncthreads();
        }
    }
}

// y[p,:] = max(y[p,:] + rows[img[p],:], 0) in place: a per-image addend folded into the ReLU pass that follows a GEMM.
// Pure streaming (read + write of y); rows[B,C] stays in L2.  One thread per 16 B (or per float when C % 4 != 0).
__global__ __launch_bounds__(256) void ss_add_relu4_kernel(float4* __restrict__ y, const float4* __restrict__ rows,
                                                           const long long* __restrict__ img, unsigned n, unsigned CV, int B) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned p = i / CV, c = i - p * CV;
    const long long b = img[p];
    float4 v = y[i];
    if ((unsigned long long)b < (unsigned long long)B) {
        const float4 a = rows[(unsigned)b * CV + c];
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    y[i] = v;
}

__global__ __launch_bounds__(256) void ss_add_relu1_kernel(float* __restrict__ y, const float* __restrict__ rows, const long long* __restrict__ img,
                                                           long long n, int C, int B) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long p = i / C;
    const long long b = img[p];
    const float add = (unsigned long long)b < (unsigned long long)B ? rows[b * C + (i - p * C)] : 0.f;
    y[i] = fmaxf(y[i] + add, 0.f);
}

static int ss_launch(const float* g, const int64_t* img, int64_t P, int C, int B, float* out, const float* y, float* g_pre, hipStream_t s) {
    A3D_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C, s));
    if (P == 0) return A3D_OK;
    const int vec = (C % 4 == 0 && C >= 64) ? 4 : 1;
    int CP = 1;
    while (CP < C / vec && CP < 256) CP <<= 1;
#ifdef A3D_EXPERIMENT
    static const int rows = [] {  // rows per work-group; A3D_SS_ROWS is an experiment knob (build.py --exp only), read once
        const char* e = getenv("A3D_SS_ROWS");
        return e && atoi(e) > 0 ? atoi(e) : SS_ROWS;
    }();
#else
    const int rows = SS_ROWS;
#endif
    const dim3 grid(a3d_div_up(P, rows)), block(256);
    const long long* im = (const long long*)img;
    if (y) {
        // g and y are read exactly once: non-temporal loads (5.0 -> 5.8 TB/s cold, tools/scratch/segsum_bench.py); g_pre is stored
        // normally, the two GEMMs that follow read it
        if (vec == 4) hipLaunchKernelGGL((ss_kernel<4, true, true>), grid, block, 0, s, g, im, (long long)P, C, B, CP, out, y, g_pre, rows);
        else hipLaunchKernelGGL((ss_kernel<1, true>), grid, block, 0, s, g, im, (long long)P, C, B, CP, out, y, g_pre, rows);
    } else {
        if (vec == 4) hipLaunchKernelGGL((ss_kernel<4, false>), grid, block, 0, s, g, im, (long long)P, C, B, CP, out, y, g_pre, rows);
        else hipLaunchKernelGGL((ss_kernel<1, false>), grid, block, 0, s, g, im, (long long)P, C, B, CP, out, y, g_pre, rows);
    }
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_rows_segsum(const float* g, const int64_t* img, int64_t P, int C, int B, float* out, a3d_stream_t stream) {
    A3D_CHECK_ARG(out && B > 0 && C > 0 && P >= 0);
    A3D_CHECK_ARG(P == 0 || (g && img));
    return ss_launch(g, img, P, C, B, out, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int a3d_rows_add_relu_fwd(float* y, const float* rows, const int64_t* img, int64_t P, int C, int B, a3d_stream_t stream) {
    A3D_CHECK_ARG(B > 0 && C > 0 && P >= 0);
    if (P == 0) return A3D_OK;
    A3D_CHECK_ARG(y && rows && img);
    const long long n = (long long)P * C;
    if (C % 4 == 0 && n / 4 < 0xffffffffll && (long long)B * C / 4 < 0xffffffffll)
        hipLaunchKernelGGL(ss_add_relu4_kernel, dim3(a3d_div_up(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (float4*)y, (const float4*)rows,
                           (const long long*)img, (unsigned)(n / 4), (unsigned)(C / 4), B);
    else
        hipLaunchKernelGGL(ss_add_relu1_kernel, dim3(a3d_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, y, rows, (const long long*)img, n, C, B);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_rows_add_relu_bwd(const float* g, const float* y, const int64_t* img, int64_t P, int C, int B, float* g_pre, float* g_rows,
                                     a3d_stream_t stream) {
    A3D_CHECK_ARG(g_rows && B > 0 && C > 0 && P >= 0);
    A3D_CHECK_ARG(P == 0 || (g && y && img && g_pre));
    return ss_launch(g, img, P, C, B, g_rows, y, g_pre, (hipStream_t)stream);
}
