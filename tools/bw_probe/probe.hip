// What the memory system gives a frame-sized streaming pass on this GPU: float4 fills / reads / copies of the sizes the compositor moves
// (92 MB of image per step at B = 16, 256 x 256, 4 + 18 channels) and of 1 GiB, one float4 per thread (the compositor's shape: a work-group
// per 256 pixels) and grid-stride over 8192 work-groups, plain and non-temporal stores.  Stand-alone (no torch):
//     hipcc --offload-arch=gfx950 -O3 tools/bw_probe/probe.hip -o gpurun_out/bw_probe && gpurun_out/bw_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            printf("%s: %s\n", #x, hipGetErrorString(e_));          \
            exit(1);                                                \
        }                                                           \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void fill_one(v4f* p, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    if (NT) __builtin_nontemporal_store(v, p + i);
    else p[i] = v;
}
template <bool NT>
__global__ __launch_bounds__(256) void fill_stride(v4f* p, long long n) {
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, p + i);
        else p[i] = v;
    }
}
__global__ __launch_bounds__(256) void read_stride(const v4f* p, long long n, float* out) {
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const v4f v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void copy_one(const v4f* a, v4f* b, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(a[i], b + i);
}

int main() {
    const long long big = 1ll << 30;
    v4f *a, *b;
    float* out;
    CK(hipMalloc(&a, big));
    CK(hipMalloc(&b, big));
    CK(hipMalloc(&out, 16));
    CK(hipMemset(a, 0, big));
    CK(hipMemset(b, 0, big));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long long sizes[] = {92ll << 20, 1ll << 30};
    for (long long bytes : sizes) {
        const long long n = bytes / 16;
        const int one = (int)((n + 255) / 256);
        auto timeit = [&](const char* name, auto launch, double moved) {
            for (int i = 0; i < 3; ++i) launch();
            CK(hipEventRecord(e0, s));
            const int reps = bytes > (512ll << 20) ? 10 : 50;
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%5lld MB  %-44s %8.1f us  %6.2f TB/s\n", bytes >> 20, name, 1e3 * ms / reps, moved * reps / (ms * 1e-3) / 1e12);
        };
        timeit("fill, one float4 per thread, plain", [&]() { hipLaunchKernelGGL(fill_one<false>, dim3(one), dim3(256), 0, s, a, n); }, (double)bytes);
        timeit("fill, one float4 per thread, non-temporal", [&]() { hipLaunchKernelGGL(fill_one<true>, dim3(one), dim3(256), 0, s, a, n); }, (double)bytes);
        timeit("fill, 8192 work-groups grid-stride, plain", [&]() { hipLaunchKernelGGL(fill_stride<false>, dim3(8192), dim3(256), 0, s, a, n); }, (double)bytes);
        timeit("fill, 8192 work-groups grid-stride, non-temporal", [&]() { hipLaunchKernelGGL(fill_stride<true>, dim3(8192), dim3(256), 0, s, a, n); }, (double)bytes);
        timeit("fill, 2048 work-groups grid-stride, non-temporal", [&]() { hipLaunchKernelGGL(fill_stride<true>, dim3(2048), dim3(256), 0, s, a, n); }, (double)bytes);
        timeit("read, 8192 work-groups grid-stride", [&]() { hipLaunchKernelGGL(read_stride, dim3(8192), dim3(256), 0, s, a, n, out); }, (double)bytes);
        timeit("copy, one float4 per thread (read + nt write)", [&]() { hipLaunchKernelGGL(copy_one, dim3(one), dim3(256), 0, s, a, b, n); }, 2.0 * bytes);
    }
    return 0;
}
