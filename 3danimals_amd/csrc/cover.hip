// Covered-pixel list on gfx950: stream compaction of (rast.w > 0) into flat pixel indices, image-major and, inside an image,
// 8x8 tile by tile -- the point list the G-buffer kernel, the texture/DINO fields and the compositing scatter run over.
// (The reference shades every pixel of the frame, render.py:30-132; evaluating only the covered ones is output-identical
//  because uncovered pixels are composited with alpha 0, render.py:261-262.)
//
// Two steps: per-work-group ballot counts (a pass of its own, or left behind by the rasteriser's resolve) + fire-and-forget adds of
// every count into the sum of its group of 64 blocks; then the ordered emit, in which every work-group sums the group sums before
// its group and the block counts before it inside the group itself (<= nb/64 + 64 four-byte loads from L2, eight in flight).  The
// host reads the group sums back (ceil(nb/64) ints) and adds them up: the list's length.  No scan launch: the single-work-group scan
// of the first version cost a launch (~9 us of event time for 16 KB of data) between the rasteriser and everything that follows.
// One wave covers one 8x8 tile, so a work-group of 256 threads covers 4 tiles; entry k of the tile-ordered pixel space is
//   k = ((b*H/8 + ty)*W/8 + tx)*64 + iy*8 + ix        (tile = 8)      or      k = flat index      (tile = 0, row-major).
#include "a3d_common.h"
#include "cover_common.h"

namespace {

constexpr int CV_BLOCK = 256;

__global__ __launch_bounds__(CV_BLOCK) void cv_count_kernel(const float4* __restrict__ rast, long long n, int H, int W, int tile,
                                                            int* __restrict__ block_count, int* __restrict__ group_sum) {
    __shared__ int wave_n[CV_BLOCK / 64];
    const long long k = (long long)blockIdx.x * CV_BLOCK + threadIdx.x;
    const bool on = k < n && rast[cv_flat(k, H, W, tile)].w > 0.f;
    const unsigned long long m = __ballot(on);
    if ((threadIdx.x & 63) == 0) wave_n[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int c = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
        block_count[blockIdx.x] = c;
        if (c) atomicAdd(group_sum + (long long)(blockIdx.x / A3D_COVER_GROUP) * A3D_COVER_GROUP_STRIDE, c);
    }
}

__global__ __launch_bounds__(CV_BLOCK) void cv_emit_kernel(const float4* __restrict__ rast, long long n, int H, int W, int tile,
                                                           const int* __restrict__ block_count, const int* __restrict__ group_sum,
                                                           long long* __restrict__ pix, int* __restrict__ inv) {
    __shared__ int wave_n[CV_BLOCK / 64];
    __shared__ int s_off;
    const long long k = (long long)blockIdx.x * CV_BLOCK + threadIdx.x;
    const long long flat = k < n ? cv_flat(k, H, W, tile) : 0;
    const float idw = k < n ? rast[flat].w : 0.f;  // (issued before the offset's loads: the two latencies overlap)
    const int wave = threadIdx.x >> 6;
    if (wave == 0) {
        const int off = cv_block_offset_wave0(block_count, group_sum, (int)blockIdx.x);
        if (threadIdx.x == 0) s_off = off;
    }
    const bool on = idw > 0.f;
    const unsigned long long m = __ballot(on);
    if ((threadIdx.x & 63) == 0) wave_n[wave] = __popcll(m);
    __syncthreads();
    const int block_off = s_off;
    if (!on) {
        if (inv && k < n) inv[flat] = -1;
        return;
    }
    int o = block_off + a3d_wave_prefix(m);
    for (int w = 0; w < wave; ++w) o += wave_n[w];
    pix[o] = flat;
    if (inv) inv[flat] = o;  // pixel -> entry of the list: what the fused compositor reads
}

}  // namespace

extern "C" size_t a3d_cover_scratch_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t nb = (size_t)a3d_div_up((long long)B * H * W, CV_BLOCK);
    // block counts | group sums (one per 64-byte line) | (round 5) the look-back flags of a3d_rast_resolve_gbuffer_fwd: one per block, one per group
    const size_t ng = (size_t)a3d_div_up((long long)nb, A3D_COVER_GROUP);
    return sizeof(int) * (nb + ng * A3D_COVER_GROUP_STRIDE + nb + ng);
}

extern "C" int a3d_cover_blocks(int B, int H, int W) { return (B <= 0 || H <= 0 || W <= 0) ? 0 : a3d_div_up((long long)B * H * W, CV_BLOCK); }
extern "C" int a3d_cover_groups(int B, int H, int W) { return a3d_div_up(a3d_cover_blocks(B, H, W), A3D_COVER_GROUP); }
extern "C" int a3d_cover_group_stride(void) { return A3D_COVER_GROUP_STRIDE; }

static int cv_check(const float* rast, int B, int H, int W, int tile, const void* scratch) {
    A3D_CHECK_ARG(rast && scratch && B > 0 && H > 0 && W > 0 && (long long)B * H * W < 0x7fffffffll);
    A3D_CHECK_ARG(tile == 0 || (tile == 8 && H % 8 == 0 && W % 8 == 0));
    return A3D_OK;
}

extern "C" int a3d_cover_count(const float* rast, int B, int H, int W, int tile, void* scratch, a3d_stream_t stream) {
    if (int rc = cv_check(rast, B, H, W, tile, scratch)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)B * H * W;
    const int nb = a3d_div_up(n, CV_BLOCK);
    int* group_sum = (int*)scratch + nb;
    A3D_HIP(hipMemsetAsync(group_sum, 0, sizeof(int) * (size_t)a3d_div_up(nb, A3D_COVER_GROUP) * A3D_COVER_GROUP_STRIDE, s));
    hipLaunchKernelGGL(cv_count_kernel, dim3(nb), dim3(CV_BLOCK), 0, s, (const float4*)rast, n, H, W, tile, (int*)scratch, group_sum);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_cover_emit(const float* rast, int B, int H, int W, int tile, const void* scratch, int64_t* pix, int32_t* inv_or_null,
                              a3d_stream_t stream) {
    if (int rc = cv_check(rast, B, H, W, tile, scratch)) return rc;
    A3D_CHECK_ARG(pix || inv_or_null);  // an empty list (total = 0) has no pix storage; the inverse map is still written
    const long long n = (long long)B * H * W;
    const int nb = a3d_div_up(n, CV_BLOCK);
    hipLaunchKernelGGL(cv_emit_kernel, dim3(nb), dim3(CV_BLOCK), 0, (hipStream_t)stream, (const float4*)rast, n, H, W, tile, (const int*)scratch,
                       (const int*)scratch + nb, (long long*)pix, inv_or_null);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
