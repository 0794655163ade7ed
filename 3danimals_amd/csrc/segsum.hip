// out[img[p], :] += g[p, :]  -- the adjoint of broadcasting a per-image row (image feature, light parameters, camera rotation)
// to the covered pixels of that image (shade(): feat / w2c / light are [B,.] against [P,.] points, render.py:53-94).
// torch's index_add does P x C float atomics onto B rows (280 us for P=2e5, C=256).  The point list is sorted by image, so a
// block owns 128 consecutive rows, its threads keep partial sums in registers, and one atomic per (block, column) is issued.
// blockDim = 256 = RL row-lanes x CP column-lanes; a column-lane covers VEC (1 or 4) adjacent floats, so a 256-float row is read
// as 64 x 16 B by 4 row-lanes at a time, and the narrow per-image tensors (17 columns) still keep every lane busy.
#include "a3d_common.h"

#define SS_ROWS 128

template <int VEC>
__global__ __launch_bounds__(256) void ss_kernel(const float* __restrict__ g, const long long* __restrict__ img, long long P, int C, int B,
                                                 int CP, float* __restrict__ out) {
    __shared__ float s_part[256 * VEC];
    const int RL = 256 / CP;
    const int c = threadIdx.x % CP, rl = threadIdx.x / CP;
    const int CV = C / VEC;  // columns in units of VEC floats
    const long long r0 = (long long)blockIdx.x * SS_ROWS;
    const long long r1 = min(r0 + (long long)SS_ROWS, P);
    const long long first = img[r0], last = img[r1 - 1];
    for (int cbase = 0; cbase < CV; cbase += CP) {  // uniform trip count (barriers inside)
        const int cc = cbase + c;
        const bool active = cc < CV;
        if (first == last) {  // the whole block belongs to one image (the list is sorted): branch-free accumulation
            float acc[2][VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[0][k] = acc[1][k] = 0.f;
            if (active) {
                long long r = r0 + rl;
                for (; r + RL < r1; r += 2 * RL) {  // two independent row streams per thread
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        acc[0][k] += g[r * C + (long long)cc * VEC + k];
                        acc[1][k] += g[(r + RL) * C + (long long)cc * VEC + k];
                    }
                }
                if (r < r1) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[0][k] += g[r * C + (long long)cc * VEC + k];
                }
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) s_part[threadIdx.x * VEC + k] = acc[0][k] + acc[1][k];
            __syncthreads();
            if (rl == 0 && active && (unsigned long long)first < (unsigned long long)B) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float t = 0.f;
                    for (int q = 0; q < RL; ++q) t += s_part[(q * CP + c) * VEC + k];
                    atomicAdd(out + first * C + (long long)cc * VEC + k, t);
                }
            }
            __syncthreads();
        } else if (active) {  // block straddles an image boundary (at most B-1 blocks do): plain per-row atomics
            for (long long r = r0 + rl; r < r1; r += RL) {
                const long long b = img[r];
                if ((unsigned long long)b < (unsigned long long)B)
                    for (int k = 0; k < VEC; ++k) atomicAdd(out + b * C + (long long)cc * VEC + k, g[r * C + (long long)cc * VEC + k]);
            }
        }
    }
}

extern "C" int a3d_rows_segsum(const float* g, const int64_t* img, int64_t P, int C, int B, float* out, a3d_stream_t stream) {
    A3D_CHECK_ARG(out && B > 0 && C > 0 && P >= 0);
    hipStream_t s = (hipStream_t)stream;
    A3D_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C, s));
    if (P == 0) return A3D_OK;
    A3D_CHECK_ARG(g && img);
    const int vec = (C % 4 == 0 && C >= 64) ? 4 : 1;
    int CP = 1;
    while (CP < C / vec && CP < 256) CP <<= 1;
    const dim3 grid(a3d_div_up(P, SS_ROWS)), block(256);
    if (vec == 4) hipLaunchKernelGGL(ss_kernel<4>, grid, block, 0, s, g, (const long long*)img, (long long)P, C, B, CP, out);
    else hipLaunchKernelGGL(ss_kernel<1>, grid, block, 0, s, g, (const long long*)img, (long long)P, C, B, CP, out);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
