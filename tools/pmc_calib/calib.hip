// Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (tools/pmc_calibrate.sh): measurement
// infrastructure, not part of liba3d_hip.  /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE is half the bytes of a wide
// coalesced read; other widths and WRITE_SIZE are uncalibrated -- "calibrate on a known byte count in your own access pattern".
// The patterns are the ones the hot-path kernels use: 16-byte streaming stores / loads, 4-byte stores one per 64-byte line, random
// 4-byte gathers, fire-and-forget 64-bit atomicMin on random keys (the rasteriser), float atomicAdd by 16 adjacent lanes onto one
// 64-byte row (the gradient rows), 4-byte streaming stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void cal_fill16(float4* p, long long n) {  // n float4s written once
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void cal_fill4(float* p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 1.f;
}
__global__ void cal_read16(const float4* p, long long n, float* out) {  // n float4s read once (the sum keeps the loads alive)
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ void cal_store_per_line(float* p, long long lines) {  // one 4-byte store into each 64-byte line
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < lines) p[16 * i] = 1.f;
}
__global__ void cal_gather4(const float* p, const int* idx, long long n, float* out) {  // n random 4-byte loads
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    if (i < n) s = p[idx[i]];
    if (s == 12345.678f) out[0] = s;
}
__global__ void cal_atomic_min64(unsigned long long* keys, const int* idx, long long n) {  // n fire-and-forget 64-bit atomicMin on random keys
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicMin(keys + idx[i], (unsigned long long)i);
}
__global__ void cal_atomic_rows(float* rows, const int* idx, long long n_rows_hit) {  // 16 adjacent lanes add onto one 64-byte row
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long r = t >> 4;
    if (r < n_rows_hit) atomicAdd(rows + 16ll * idx[r] + (t & 15), 1.f);
}

extern "C" int cal_run(int which, void* a, const void* b, long long n, void* out, hipStream_t s) {
    const int blocks = (int)((n + 255) / 256 > 65535 * 16 ? 65535 * 16 : (n + 255) / 256);
    switch (which) {
        case 0: hipLaunchKernelGGL(cal_fill16, dim3(8192), dim3(256), 0, s, (float4*)a, n); break;
        case 1: hipLaunchKernelGGL(cal_fill4, dim3(8192), dim3(256), 0, s, (float*)a, n); break;
        case 2: hipLaunchKernelGGL(cal_read16, dim3(8192), dim3(256), 0, s, (const float4*)a, n, (float*)out); break;
        case 3: hipLaunchKernelGGL(cal_store_per_line, dim3(blocks), dim3(256), 0, s, (float*)a, n); break;
        case 4: hipLaunchKernelGGL(cal_gather4, dim3(blocks), dim3(256), 0, s, (const float*)a, (const int*)b, n, (float*)out); break;
        case 5: hipLaunchKernelGGL(cal_atomic_min64, dim3(blocks), dim3(256), 0, s, (unsigned long long*)a, (const int*)b, n); break;
        case 6: hipLaunchKernelGGL(cal_atomic_rows, dim3((int)((16 * n + 255) / 256)), dim3(256), 0, s, (float*)a, (const int*)b, n); break;
        default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
