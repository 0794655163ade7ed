// out[img[p], :] += g[p, :]  -- the adjoint of broadcasting a per-image row (image feature, light parameters, camera rotation)
// to the covered pixels of that image (shade(): feat / w2c / light are [B,.] against [P,.] points, render.py:53-94).
// torch's index_add does P x C float atomics onto B rows (280 us for P=2e5, C=256).  The point list is sorted by image, so a
// block walks 256 consecutive rows, keeps per-column partial sums in registers while the image id stays the same, and issues
// one atomic per (block, image, column).  Reads are full 4C-byte rows, coalesced over the columns.
#include "a3d_common.h"

#define SS_ROWS 128

__global__ __launch_bounds__(256) void ss_kernel(const float* __restrict__ g, const long long* __restrict__ img, long long P, int C, int B,
                                                 float* __restrict__ out) {
    const long long r0 = (long long)blockIdx.x * SS_ROWS;
    const long long r1 = min(r0 + (long long)SS_ROWS, P);
    const long long first = img[r0], last = img[r1 - 1];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        if (first == last) {
            // common case (the list is sorted by image): the whole block belongs to one image -> branch-free, 8 loads in flight
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
            long long r = r0;
            for (; r + 8 <= r1; r += 8) {
                const float* q = g + r * C + c;
                a0 += q[0]; a1 += q[(long long)C]; a2 += q[2ll * C]; a3 += q[3ll * C];
                a4 += q[4ll * C]; a5 += q[5ll * C]; a6 += q[6ll * C]; a7 += q[7ll * C];
            }
            for (; r < r1; ++r) a0 += g[r * C + c];
            if ((unsigned long long)first < (unsigned long long)B) atomicAdd(out + first * C + c, ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)));
            continue;
        }
        float acc = 0.f;
        long long cur = first;
        for (long long r = r0; r < r1; ++r) {
            const long long b = img[r];  // wave-uniform
            if (b != cur) {
                if ((unsigned long long)cur < (unsigned long long)B) atomicAdd(out + cur * C + c, acc);
                acc = 0.f;
                cur = b;
            }
            acc += g[r * C + c];
        }
        if ((unsigned long long)cur < (unsigned long long)B) atomicAdd(out + cur * C + c, acc);
    }
}

extern "C" int a3d_rows_segsum(const float* g, const int64_t* img, int64_t P, int C, int B, float* out, a3d_stream_t stream) {
    A3D_CHECK_ARG(out && B > 0 && C > 0 && P >= 0);
    hipStream_t s = (hipStream_t)stream;
    A3D_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C, s));
    if (P == 0) return A3D_OK;
    A3D_CHECK_ARG(g && img);
    hipLaunchKernelGGL(ss_kernel, dim3(a3d_div_up(P, SS_ROWS)), dim3(256), 0, s, g, (const long long*)img, (long long)P, C, B, out);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
