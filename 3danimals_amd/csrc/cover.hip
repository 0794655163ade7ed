// Covered-pixel list on gfx950: stream compaction of (rast.w > 0) into flat pixel indices, image-major and, inside an image,
// 8x8 tile by tile -- the point list the G-buffer kernel, the texture/DINO fields and the compositing scatter run over.
// (The reference shades every pixel of the frame, render.py:30-132; evaluating only the covered ones is output-identical
//  because uncovered pixels are composited with alpha 0, render.py:261-262.)
//
// Same three-step shape as the DMTet extraction: per-work-group ballot counts -> one-work-group scan -> ordered emit.
// One wave covers one 8x8 tile, so a work-group of 256 threads covers 4 tiles; entry k of the tile-ordered pixel space is
//   k = ((b*H/8 + ty)*W/8 + tx)*64 + iy*8 + ix        (tile = 8)      or      k = flat index      (tile = 0, row-major).
#include "a3d_common.h"

namespace {

constexpr int CV_BLOCK = 256;
constexpr int CV_SCAN_THREADS = 1024;

// (all 32-bit: B*H*W < 2^31 is checked by the entry points, and a 64-bit division costs ~150 instructions per thread)
__device__ __forceinline__ long long cv_flat(long long k64, int H, int W, int tile) {
    if (tile == 0) return k64;
    const unsigned k = (unsigned)k64;
    const unsigned in_tile = k & 63u;
    unsigned t = k >> 6;
    const unsigned tw = (unsigned)W >> 3, th = (unsigned)H >> 3;
    const unsigned tx = t % tw; t /= tw;
    const unsigned ty = t % th; t /= th;  // t = image
    return (long long)((t * (unsigned)H + (ty * 8u + (in_tile >> 3))) * (unsigned)W + tx * 8u + (in_tile & 7u));
}

__global__ __launch_bounds__(CV_BLOCK) void cv_count_kernel(const float4* __restrict__ rast, long long n, int H, int W, int tile,
                                                            int* __restrict__ block_count) {
    __shared__ int wave_n[CV_BLOCK / 64];
    const long long k = (long long)blockIdx.x * CV_BLOCK + threadIdx.x;
    const bool on = k < n && rast[cv_flat(k, H, W, tile)].w > 0.f;
    const unsigned long long m = __ballot(on);
    if ((threadIdx.x & 63) == 0) wave_n[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
}

// exclusive scan of block_count[nb] in place; total[0] = number of covered pixels
__global__ __launch_bounds__(CV_SCAN_THREADS) void cv_scan_kernel(int* __restrict__ block_count, int nb, long long* __restrict__ total) {
    __shared__ int wave_tot[CV_SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // contiguous run per thread, one barrier (see a3d_run_sum)
    const int per = (nb + CV_SCAN_THREADS - 1) / CV_SCAN_THREADS;
    const int lo = min((int)threadIdx.x * per, nb), hi = min(lo + per, nb);
    const int mine = a3d_run_sum(block_count, lo, hi);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int run = incl - mine;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
    run = a3d_run_scan<false>(block_count, block_count, lo, hi, run);
    if (threadIdx.x == CV_SCAN_THREADS - 1) total[0] = run;
}

__global__ __launch_bounds__(CV_BLOCK) void cv_emit_kernel(const float4* __restrict__ rast, long long n, int H, int W, int tile,
                                                           const int* __restrict__ block_off, long long* __restrict__ pix,
                                                           int* __restrict__ inv) {
    __shared__ int wave_n[CV_BLOCK / 64];
    const long long k = (long long)blockIdx.x * CV_BLOCK + threadIdx.x;
    const long long flat = k < n ? cv_flat(k, H, W, tile) : 0;
    const bool on = k < n && rast[flat].w > 0.f;
    const unsigned long long m = __ballot(on);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) wave_n[wave] = __popcll(m);
    __syncthreads();
    if (!on) {
        if (inv && k < n) inv[flat] = -1;
        return;
    }
    int o = block_off[blockIdx.x] + a3d_wave_prefix(m);
    for (int w = 0; w < wave; ++w) o += wave_n[w];
    pix[o] = flat;
    if (inv) inv[flat] = o;  // pixel -> entry of the list: what the fused compositor reads
}

}  // namespace

extern "C" size_t a3d_cover_scratch_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return sizeof(int) * (size_t)a3d_div_up((long long)B * H * W, CV_BLOCK);
}

static int cv_check(const float* rast, int B, int H, int W, int tile, const void* scratch) {
    A3D_CHECK_ARG(rast && scratch && B > 0 && H > 0 && W > 0 && (long long)B * H * W < 0x7fffffffll);
    A3D_CHECK_ARG(tile == 0 || (tile == 8 && H % 8 == 0 && W % 8 == 0));
    return A3D_OK;
}

extern "C" int a3d_cover_count(const float* rast, int B, int H, int W, int tile, void* scratch, int counted, int64_t* total,
                               a3d_stream_t stream) {
    if (int rc = cv_check(rast, B, H, W, tile, scratch)) return rc;
    A3D_CHECK_ARG(total);
    A3D_CHECK_ARG(!counted || (tile == 8 && ((long long)H * W) % CV_BLOCK == 0));
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)B * H * W;
    const int nb = a3d_div_up(n, CV_BLOCK);
    if (!counted) {  // (counted: a3d_rast_fwd left the block counts in scratch)
        hipLaunchKernelGGL(cv_count_kernel, dim3(nb), dim3(CV_BLOCK), 0, s, (const float4*)rast, n, H, W, tile, (int*)scratch);
        A3D_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(cv_scan_kernel, dim3(1), dim3(CV_SCAN_THREADS), 0, s, (int*)scratch, nb, (long long*)total);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_cover_emit(const float* rast, int B, int H, int W, int tile, const void* scratch, int64_t* pix, int32_t* inv_or_null,
                              a3d_stream_t stream) {
    if (int rc = cv_check(rast, B, H, W, tile, scratch)) return rc;
    A3D_CHECK_ARG(pix || inv_or_null);  // an empty list (total = 0) has no pix storage; the inverse map is still written
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(cv_emit_kernel, dim3(a3d_div_up(n, CV_BLOCK)), dim3(CV_BLOCK), 0, (hipStream_t)stream, (const float4*)rast, n, H, W,
                       tile, (const int*)scratch, (long long*)pix, inv_or_null);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
