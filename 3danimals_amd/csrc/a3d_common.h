// Shared helpers for the liba3d_hip kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/a3d.h"

#define A3D_WAVE 64
// covered-pixel list (cover.hip; the rasteriser's resolve writes the same scratch): per-256-pixel block counts [nb] followed by the sums
// of groups of 64 consecutive blocks, one sum per 64-byte line [ceil(nb / 64) * 16 ints] (the sums are accumulated with device atomics:
// neighbours in one line serialise against each other in the L2's atomic unit)
#define A3D_COVER_GROUP 64
#define A3D_COVER_GROUP_STRIDE 16

void a3d_set_error(const char* fmt, ...);

#define A3D_CHECK_ARG(cond)                                                   \
    do {                                                                      \
        if (!(cond)) {                                                        \
            a3d_set_error("%s: invalid argument: %s", __func__, #cond);       \
            return A3D_EINVAL;                                                \
        }                                                                     \
    } while (0)

#define A3D_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            a3d_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e_));       \
            return A3D_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

#define A3D_LAUNCH_CHECK() A3D_HIP(hipGetLastError())

static inline int a3d_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// experiment knob for kernel bisection (tools/kernel_lab.py): integer value of the environment variable A3D_EXP, 0 when unset
int a3d_exp(void);

#ifdef __HIPCC__
// ---- phase stamps inside kernels (tools/kernel_phases.py; the library proper is built WITHOUT them: build.py --profile makes a second,
// instrumented liba3d_hip_prof.so).  A3D_STAMP(kernel id within the TU, slot 0..7): thread 0 of every work-group writes the 100 MHz wall
// clock to buf[work-group][slot] when the TU's active kernel id is that one; A3D_STAMP_CLOCK the shader clock (for the frequency the
// launch ran at).  A3D_PROFILE_TU(name) defines the TU's setter a3d_profile_set_<name>(buf, kid).  What this is for: every hot-path
// kernel here is bound by dependent round trips, barriers or the instructions of its longest-lived work-group, not by bytes, and the
// counters say so only in aggregate -- the stamps say which phase of which work-group.
#ifdef A3D_PROFILE
static __device__ unsigned long long* a3d_prof_buf;
static __device__ int a3d_prof_kid = -1;
#define A3D_PROF_MAX_WG 65536
#define A3D_STAMP_(kid, slot, what)                                                                                              \
    do {                                                                                                                         \
        if (a3d_prof_kid == (kid) && threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {                                 \
            const size_t wg_ = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);                   \
            if (wg_ < A3D_PROF_MAX_WG) a3d_prof_buf[wg_ * 8 + (slot)] = what;                                                    \
        }                                                                                                                        \
    } while (0)
#define A3D_STAMP(kid, slot) A3D_STAMP_(kid, slot, wall_clock64())
#define A3D_STAMP_CLOCK(kid, slot) A3D_STAMP_(kid, slot, clock64())
#define A3D_PROFILE_TU(tu)                                                                                                       \
    extern "C" int a3d_profile_set_##tu(void* buf, int kid) {                                                                    \
        if (hipMemcpyToSymbol(HIP_SYMBOL(a3d_prof_buf), &buf, sizeof(buf)) != hipSuccess) return A3D_EHIP;                       \
        return hipMemcpyToSymbol(HIP_SYMBOL(a3d_prof_kid), &kid, sizeof(kid)) == hipSuccess ? A3D_OK : A3D_EHIP;                 \
    }
#else
#define A3D_STAMP(kid, slot) \
    do {                     \
    } while (0)
#define A3D_STAMP_CLOCK(kid, slot) \
    do {                           \
    } while (0)
#define A3D_PROFILE_TU(tu)
#endif

// lanes below me in the wave that have the bit set: ballot + mbcnt (wave64)
__device__ __forceinline__ int a3d_lane_id() { return (int)__lane_id(); }

__device__ __forceinline__ int a3d_wave_prefix(unsigned long long mask) {
    // number of set bits among lanes strictly below the calling lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// 16-byte load of data that is read exactly once (streamed): non-temporal, so it does not displace what the next kernels reuse
__device__ __forceinline__ float4 a3d_load_stream4(const float* p) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f a = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(a.x, a.y, a.z, a.w);
}
__device__ __forceinline__ float a3d_load_stream(const float* p) { return __builtin_nontemporal_load(p); }

// Sum of a[lo, hi): eight independent loads in flight per step.  (A plain `for (i) s += a[i]` waits for every load before it
// issues the next one: ~1 us per element, 46 us for a 23-element run per thread in the single-work-group scans.)
__device__ __forceinline__ int a3d_run_sum(const int* __restrict__ a, int lo, int hi) {
    int s = 0, i = lo;
    for (; i + 8 <= hi; i += 8) {
        const int v0 = a[i], v1 = a[i + 1], v2 = a[i + 2], v3 = a[i + 3], v4 = a[i + 4], v5 = a[i + 5], v6 = a[i + 6], v7 = a[i + 7];
        s += ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
    }
    if (i + 4 <= hi) {
        const int v0 = a[i], v1 = a[i + 1], v2 = a[i + 2], v3 = a[i + 3];
        s += (v0 + v1) + (v2 + v3);
        i += 4;
    }
    if (i + 2 <= hi) {
        const int v0 = a[i], v1 = a[i + 1];
        s += v0 + v1;
        i += 2;
    }
    if (i < hi) s += a[i];
    return s;
}

// dst[i] = run + a[lo..i) for i in [lo, hi) (exclusive prefix of the run, starting from `run`), src optionally reset to `fill`; returns
// run + sum.  dst may alias a.  Eight loads in flight per step, as above.
template <bool RESET>
__device__ __forceinline__ int a3d_run_scan(int* a, int* dst, int lo, int hi, int run, int fill = 0) {
    int i = lo;
    for (; i + 8 <= hi; i += 8) {
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = a[i + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            dst[i + k] = run;
            if (RESET) a[i + k] = fill;
            run += v[k];
        }
    }
    if (i + 4 <= hi) {
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = a[i + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[i + k] = run;
            if (RESET) a[i + k] = fill;
            run += v[k];
        }
        i += 4;
    }
    if (i + 2 <= hi) {
        const int v0 = a[i], v1 = a[i + 1];
        dst[i] = run; dst[i + 1] = run + v0;
        if (RESET) { a[i] = fill; a[i + 1] = fill; }
        run += v0 + v1;
        i += 2;
    }
    if (i < hi) {
        const int c = a[i];
        dst[i] = run;
        if (RESET) a[i] = fill;
        run += c;
    }
    return run;
}

// inclusive prefix sum over the 64 lanes of a wave in six DPP additions (within rows of 16: row_shr 1, 2, 4, 8; across rows: row_bcast 15,
// 31) -- VALU only, no LDS crossbar (six ds_bpermute round trips in the __shfl_up form)
__device__ __forceinline__ int a3d_wave_incl_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
    return x;
}

__device__ __forceinline__ float a3d_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
#endif
