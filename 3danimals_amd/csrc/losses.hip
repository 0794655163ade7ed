// Reconstruction losses on gfx950 -- the image-space consumers of the renderer's outputs, fused
// (/root/reference/model/models/AnimalModel.py:260-307, compute_reconstruction_losses, background_mode 'none'; images = B*F frames):
//   mask        = mean((m*valid - mask_gt)^2)                    m = shaded alpha (antialiased coverage)
//   mask_inv_dt = mean((1 - m) * dt0)                            dt0 / dt1 = distance transforms of the target mask / its complement
//   mask_dt     = mean(m * dt1)
//   flow        = sum((flow - flow_gt)^2 * both) / max(2 * sum(both), 1) per frame pair, 0 where |flow_gt| > 0.5 on the mask
//   both        = erode3x3((m*valid > 0) * mask_gt)              avg_pool2d(3, stride 1, pad 1) > 0.99, zero padded
//   rgb         = mean(|rgb - image_gt| * both)     over 3 channels
//   dino        = mean((dino - dino_gt)^2 * both)   over D channels
// The reference runs ~15 elementwise/reduction passes over 4- and 17-channel full-resolution images forward and about twice that
// backward; here one thread per pixel reads the renderer's NHWC buffers once (20 contiguous floats) and the dataset's planar NCHW
// targets once, per-image sums are reduced wave -> block -> a second tiny kernel (fixed order: bit-reproducible), and the backward
// writes both image gradients in one pass.
#include "a3d_common.h"

namespace {

constexpr int LS_BLOCK = 256;
constexpr int LS_NL = 5;  // loss columns: mask, mask_inv_dt, rgb, dino, mask_dt

// partial[(b*nblk + blk)*4 + k]: per-block sums of the four summands
// (round 6) ``ds``: floats between two pixels of ``dino`` (D = contiguous; D + 1 = the renderer's 17-channel feature image read in place,
// its alpha channel skipped -- render.py:330-331 slices it off, here nobody copies the 16 channels out first)
template <bool VEC>  // VEC: dino is contiguous in its D channels and 16-byte aligned (the float4 form); else any pixel stride, four scalar loads
__global__ __launch_bounds__(LS_BLOCK) void ls_fwd_kernel(const float* __restrict__ shaded, const float* __restrict__ dino, int D, int ds,
                                                          const float* __restrict__ image_gt, const float* __restrict__ dino_gt,
                                                          const float* __restrict__ mask_gt, const float* __restrict__ dt0,
                                                          const float* __restrict__ dt1, long long dt_stride,
                                                          const float* __restrict__ valid, int H, int W, float* __restrict__ partial,
                                                          unsigned char* __restrict__ both_out) {
    __shared__ float red[LS_BLOCK / 64][LS_NL];
    // q = (rendered mask * valid > 0) * mask_gt of the rows above / of / below this work-group's 256 pixels (+ one pixel either side):
    // the 3x3 erosion then reads LDS instead of 27 global loads per pixel (the kernel was bound by those, not by its 177 MB)
    __shared__ float s_q[3][LS_BLOCK + 2];
    const int b = blockIdx.y, HW = H * W;
    const int i = blockIdx.x * LS_BLOCK + threadIdx.x;
    {
        const long long img0 = (long long)b * HW;
        const int i0 = blockIdx.x * LS_BLOCK;
        for (int k = threadIdx.x; k < 3 * (LS_BLOCK + 2); k += LS_BLOCK) {
            const int r = k / (LS_BLOCK + 2), c = k - r * (LS_BLOCK + 2);
            const int j = i0 + (r - 1) * W - 1 + c;
            float q = 0.f;
            if (j >= 0 && j < HW) {
                const long long p = img0 + j;
                q = (shaded[4 * p + 3] * valid[p] > 0.f ? 1.f : 0.f) * mask_gt[p];
            }
            s_q[r][c] = q;
        }
        __syncthreads();
    }
    float v[LS_NL] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float both = 0.f;  // of this thread's own pixel
    if (i < HW) {
        const long long img_px = (long long)b * HW, p = img_px + i;
        const int y = i / W, x = i - y * W;
        const float4 s = reinterpret_cast<const float4*>(shaded)[p];
        const float m = s.w, t = m * valid[p] - mask_gt[p];
        v[0] = t * t;
        v[1] = (1.f - m) * dt0[(long long)b * dt_stride + i];
        if (dt1) v[4] = m * dt1[(long long)b * dt_stride + i];
        {
            float sum = 0.f;  // 3x3 box sum, dy outer, dx inner (the order of the first version's 27-load form: bit-identical); zero padding of avg_pool2d outside the frame
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const bool in = x + dx >= 0 && x + dx < W && y + dy >= 0 && y + dy < H;
                    sum += in ? s_q[dy + 1][threadIdx.x + 1 + dx] : 0.f;
                }
            both = sum / 9.f > 0.99f ? 1.f : 0.f;
        }
        both_out[p] = both > 0.f ? 1 : 0;  // saved for the backward: 1 byte instead of 27 neighbourhood loads per pixel
        const float* g = image_gt + (long long)b * 3 * HW + i;
        v[2] = (fabsf(s.x - g[0]) + fabsf(s.y - g[HW]) + fabsf(s.z - g[2ll * HW])) * both;
    }
    if (dino) {
        if ((D & 3) == 0) {
            // wave-cooperative: the wave's 64 pixels own 64*D contiguous floats of the NHWC buffer; lane l takes float4 number
            // t*64 + l of that chunk (fully coalesced 1 KB loads) -- it belongs to pixel f / (D/4), channel group f % (D/4)
            const int D4 = D >> 2, lane = threadIdx.x & 63;
            const int px0 = blockIdx.x * LS_BLOCK + (threadIdx.x & ~63);  // first pixel of this wave inside image b
            const long long img_px = (long long)b * HW;
            const float* cbase = dino + (img_px + px0) * (VEC ? D : ds);
            float acc = 0.f;
            for (int t = 0; t < D4; ++t) {
                const int f = t * 64 + lane, pl = f / D4, cg = f - pl * D4, px = px0 + pl;
                const float both_pl = __shfl(both, pl, 64);  // pixel pl's mask lives in lane pl (all lanes take part)
                if (px < HW) {
                    float4 q;
                    if (VEC) q = reinterpret_cast<const float4*>(cbase)[f];
                    else { const float* qp = cbase + (long long)pl * ds + 4 * cg; q = make_float4(qp[0], qp[1], qp[2], qp[3]); }
                    const float* dg = dino_gt + ((long long)b * D + 4 * cg) * HW + px;
                    const float e0 = q.x - dg[0], e1 = q.y - dg[HW], e2 = q.z - dg[2ll * HW], e3 = q.w - dg[3ll * HW];
                    acc += (e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3) * both_pl;
                }
            }
            v[3] = acc;
        } else if (i < HW) {
            const long long p = (long long)b * HW + i;
            const float* dp = dino + p * ds;
            const float* dg = dino_gt + (long long)b * D * HW + i;
            float acc = 0.f;
            for (int c = 0; c < D; ++c) {
                const float e = dp[c] - dg[(long long)c * HW];
                acc += e * e;
            }
            v[3] = acc * both;
        }
    }
#pragma unroll
    for (int k = 0; k < LS_NL; ++k) v[k] = a3d_wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < LS_NL; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x < LS_NL) {
        float t = 0.f;
        for (int w = 0; w < LS_BLOCK / 64; ++w) t += red[w][threadIdx.x];
        partial[((long long)b * gridDim.x + blockIdx.x) * LS_NL + threadIdx.x] = t;
    }
}

// loss[b,k] = (sum over blocks) / normaliser_k ; one wave per (image, k)
__global__ __launch_bounds__(64) void ls_finish_kernel(const float* __restrict__ partial, int nblk, int HW, int D, float* __restrict__ loss) {
    const int b = blockIdx.x, k = blockIdx.y;
    float t = 0.f;
    for (int j = threadIdx.x; j < nblk; j += 64) t += partial[((long long)b * nblk + j) * LS_NL + k];
    t = a3d_wave_sum(t);
    if (threadIdx.x == 0) {
        const float n = (k < 2 || k == 4) ? (float)HW : (k == 2 ? 3.f * (float)HW : (float)D * (float)HW);
        loss[b * LS_NL + k] = D == 0 && k == 3 ? 0.f : t / n;
    }
}

// ``gds``: floats between two pixels of g_dino (D + 1: the gradient is laid out like the 17-channel image it belongs to, so that the
// compositor's backward reads it in place; the alpha channel's slot is not written -- it has no gradient, a3d_ca_buffer.g_channels)
template <bool VEC>
__global__ __launch_bounds__(LS_BLOCK) void ls_bwd_kernel(const float* __restrict__ g_loss, const float* __restrict__ shaded,
                                                          const float* __restrict__ dino, int D, int ds, int gds, const float* __restrict__ image_gt,
                                                          const float* __restrict__ dino_gt, const float* __restrict__ mask_gt,
                                                          const float* __restrict__ dt0, const float* __restrict__ dt1, long long dt_stride,
                                                          const float* __restrict__ valid, int H, int W, const unsigned char* __restrict__ both_in,
                                                          float* __restrict__ g_shaded, float* __restrict__ g_dino) {
    const int b = blockIdx.y, HW = H * W;
    const int i = blockIdx.x * LS_BLOCK + threadIdx.x;
    const long long img_px = (long long)b * HW;
    float both = 0.f;
    if (i < HW) {
        const long long p = img_px + i;
        const float gm = g_loss[LS_NL * b] / (float)HW, gd = g_loss[LS_NL * b + 1] / (float)HW, gr = g_loss[LS_NL * b + 2] / (3.f * (float)HW);
        const float g1 = dt1 ? g_loss[LS_NL * b + 4] / (float)HW : 0.f;
        const float4 s = reinterpret_cast<const float4*>(shaded)[p];
        both = both_in[p] ? 1.f : 0.f;
        const float* g = image_gt + (long long)b * 3 * HW + i;
        auto sgn = [](float e) { return e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f); };
        float4 o;
        o.x = gr * sgn(s.x - g[0]) * both;
        o.y = gr * sgn(s.y - g[HW]) * both;
        o.z = gr * sgn(s.z - g[2ll * HW]) * both;
        const float va = valid[p];
        o.w = gm * 2.f * (s.w * va - mask_gt[p]) * va - gd * dt0[(long long)b * dt_stride + i];
        if (dt1) o.w += g1 * dt1[(long long)b * dt_stride + i];
        reinterpret_cast<float4*>(g_shaded)[p] = o;
    }
    if (!dino) return;
    const float gq0 = g_loss[LS_NL * b + 3] / ((float)D * (float)HW) * 2.f;
    if ((D & 3) == 0) {  // wave-cooperative, fully coalesced float4 loads and stores (see ls_fwd_kernel)
        const int D4 = D >> 2, lane = threadIdx.x & 63;
        const int px0 = blockIdx.x * LS_BLOCK + (threadIdx.x & ~63);
        const float* cbase = dino + (img_px + px0) * (VEC ? D : ds);
        float* gbase = g_dino + (img_px + px0) * (VEC ? D : gds);
        for (int t = 0; t < D4; ++t) {
            const int f = t * 64 + lane, pl = f / D4, cg = f - pl * D4, px = px0 + pl;
            const float gq = gq0 * __shfl(both, pl, 64);
            if (px < HW) {
                float4 q;
                if (VEC) q = reinterpret_cast<const float4*>(cbase)[f];
                else { const float* qp = cbase + (long long)pl * ds + 4 * cg; q = make_float4(qp[0], qp[1], qp[2], qp[3]); }
                const float* dg = dino_gt + ((long long)b * D + 4 * cg) * HW + px;
                float4 o4;
                o4.x = gq * (q.x - dg[0]);
                o4.y = gq * (q.y - dg[HW]);
                o4.z = gq * (q.z - dg[2ll * HW]);
                o4.w = gq * (q.w - dg[3ll * HW]);
                if (VEC) {  // streamed: 67 MB that only the compositor's gather reads back, sparsely
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    v4f nt; nt.x = o4.x; nt.y = o4.y; nt.z = o4.z; nt.w = o4.w;
                    __builtin_nontemporal_store(nt, reinterpret_cast<v4f*>(gbase) + f);
                } else {
                    float* gp = gbase + (long long)pl * gds + 4 * cg;
                    __builtin_nontemporal_store(o4.x, gp); __builtin_nontemporal_store(o4.y, gp + 1);
                    __builtin_nontemporal_store(o4.z, gp + 2); __builtin_nontemporal_store(o4.w, gp + 3);
                }
            }
        }
    } else if (i < HW) {
        const long long p = img_px + i;
        const float gq = gq0 * both;
        const float* dp = dino + p * ds;
        const float* dg = dino_gt + (long long)b * D * HW + i;
        float* go = g_dino + p * gds;
        for (int c = 0; c < D; ++c) go[c] = gq * (dp[c] - dg[(long long)c * HW]);
    }
}

// ---- flow loss between consecutive frames (AnimalModel.py:285-298).  pair = (sequence, frame f < F-1): the eroded common mask of
// frame f (saved by ls_fwd_kernel) gates the squared error; a pair whose target flow exceeds 0.5 anywhere on the mask is dropped.
// partial[(pair*nblk + blk)*3 + {0: sum err, 1: mask count, 2: large-flow hits}]
__global__ __launch_bounds__(LS_BLOCK) void fl_fwd_kernel(const float* __restrict__ flow, int pix_stride, const float* __restrict__ flow_gt,
                                                          const unsigned char* __restrict__ both, int F, int HW, float* __restrict__ partial) {
    __shared__ float red[LS_BLOCK / 64][3];
    const int pair = blockIdx.y, b = pair / (F - 1), f = pair - b * (F - 1);
    const int i = blockIdx.x * LS_BLOCK + threadIdx.x;
    float v[3] = {0.f, 0.f, 0.f};
    if (i < HW) {
        const long long p = ((long long)b * F + f) * HW + i;  // pixel of frame (b, f) in the renderer's buffers
        if (both[p]) {
            const float* g = flow_gt + (long long)pair * 2 * HW + i;
            const float e0 = flow[p * pix_stride] - g[0], e1 = flow[p * pix_stride + 1] - g[HW];
            v[0] = e0 * e0 + e1 * e1;
            v[1] = 1.f;
            v[2] = (fabsf(g[0]) > 0.5f ? 1.f : 0.f) + (fabsf(g[HW]) > 0.5f ? 1.f : 0.f);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = a3d_wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
        for (int w = 0; w < LS_BLOCK / 64; ++w) t += red[w][threadIdx.x];
        partial[((long long)pair * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = t;
    }
}

// loss[pair] and scale[pair] = (large ? 0 : 1 / max(2 * count, 1)) for the backward; one wave per pair
__global__ __launch_bounds__(64) void fl_finish_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ loss, float* __restrict__ scale) {
    const int pair = blockIdx.x;
    float t[3] = {0.f, 0.f, 0.f};
    for (int j = threadIdx.x; j < nblk; j += 64)
        for (int k = 0; k < 3; ++k) t[k] += partial[((long long)pair * nblk + j) * 3 + k];
    for (int k = 0; k < 3; ++k) t[k] = a3d_wave_sum(t[k]);
    if (threadIdx.x == 0) {
        const float sc = t[2] > 0.f ? 0.f : 1.f / fmaxf(2.f * t[1], 1.f);
        scale[pair] = sc;
        loss[pair] = t[0] * sc;
    }
}

// g_flow over ALL B*F frames (the last frame of a sequence has no pair: zeros)
__global__ __launch_bounds__(LS_BLOCK) void fl_bwd_kernel(const float* __restrict__ g_loss, const float* __restrict__ scale,
                                                          const float* __restrict__ flow, int pix_stride, const float* __restrict__ flow_gt,
                                                          const unsigned char* __restrict__ both, int F, int HW, float* __restrict__ g_flow) {
    const int n = blockIdx.y, b = n / F, f = n - b * F;
    const int i = blockIdx.x * LS_BLOCK + threadIdx.x;
    if (i >= HW) return;
    const long long p = (long long)n * HW + i;
    float g0 = 0.f, g1 = 0.f;
    if (f < F - 1 && both[p]) {
        const int pair = b * (F - 1) + f;
        const float k = 2.f * g_loss[pair] * scale[pair];
        const float* g = flow_gt + (long long)pair * 2 * HW + i;
        g0 = k * (flow[p * pix_stride] - g[0]);
        g1 = k * (flow[p * pix_stride + 1] - g[HW]);
    }
    g_flow[2 * p] = g0;
    g_flow[2 * p + 1] = g1;
}

}  // namespace

extern "C" size_t a3d_recon_losses_scratch_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return sizeof(float) * LS_NL * (size_t)B * a3d_div_up((long long)H * W, LS_BLOCK);
}

extern "C" size_t a3d_recon_losses_mask_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * H * W;
}

extern "C" int a3d_recon_losses_columns(void) { return LS_NL; }

extern "C" int a3d_recon_losses_fwd(const float* shaded, const float* dino, int D, int dino_stride, const float* image_gt, const float* dino_gt,
                                    const float* mask_gt, const float* dt0, const float* dt1_or_null, int64_t dt_stride, const float* valid, int B,
                                    int H, int W, void* scratch, uint8_t* both, float* loss, a3d_stream_t stream) {
    A3D_CHECK_ARG(shaded && image_gt && mask_gt && dt0 && valid && scratch && both && loss && B > 0 && H > 0 && W > 0 && D >= 0);
    A3D_CHECK_ARG(D == 0 || (dino && dino_gt && dino_stride >= D));
    hipStream_t s = (hipStream_t)stream;
    const int nblk = a3d_div_up((long long)H * W, LS_BLOCK);
    const bool vec = D > 0 && dino_stride == D && (((uintptr_t)dino & 15) == 0);
    hipLaunchKernelGGL(vec ? ls_fwd_kernel<true> : ls_fwd_kernel<false>, dim3(nblk, B), dim3(LS_BLOCK), 0, s, shaded, D ? dino : nullptr, D, dino_stride, image_gt, dino_gt, mask_gt, dt0, dt1_or_null,
                       (long long)dt_stride, valid, H, W, (float*)scratch, both);
    A3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(ls_finish_kernel, dim3(B, LS_NL), dim3(64), 0, s, (const float*)scratch, nblk, H * W, D, loss);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_recon_losses_bwd(const float* g_loss, const float* shaded, const float* dino, int D, int dino_stride, int g_dino_stride, const float* image_gt,
                                    const float* dino_gt, const float* mask_gt, const float* dt0, const float* dt1_or_null, int64_t dt_stride,
                                    const float* valid, int B, int H, int W, const uint8_t* both, float* g_shaded, float* g_dino,
                                    a3d_stream_t stream) {
    A3D_CHECK_ARG(g_loss && shaded && image_gt && mask_gt && dt0 && valid && both && g_shaded && B > 0 && H > 0 && W > 0 && D >= 0);
    A3D_CHECK_ARG(D == 0 || (dino && dino_gt && g_dino && dino_stride >= D && g_dino_stride >= D));
    const bool vec = D > 0 && dino_stride == D && g_dino_stride == D && ((((uintptr_t)dino | (uintptr_t)g_dino) & 15) == 0);
    hipLaunchKernelGGL(vec ? ls_bwd_kernel<true> : ls_bwd_kernel<false>, dim3(a3d_div_up((long long)H * W, LS_BLOCK), B), dim3(LS_BLOCK), 0, (hipStream_t)stream, g_loss, shaded,
                       D ? dino : nullptr, D, dino_stride, g_dino_stride, image_gt, dino_gt, mask_gt, dt0, dt1_or_null, (long long)dt_stride, valid, H, W, both, g_shaded, g_dino);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" size_t a3d_flow_loss_scratch_bytes(int B, int F, int H, int W) {
    if (B <= 0 || F < 2 || H <= 0 || W <= 0) return 16;
    return sizeof(float) * 3 * (size_t)B * (F - 1) * a3d_div_up((long long)H * W, LS_BLOCK);
}

extern "C" int a3d_flow_loss_fwd(const float* flow, int pix_stride, const float* flow_gt, const uint8_t* both, int B, int F, int H, int W,
                                 void* scratch, float* loss, float* scale, a3d_stream_t stream) {
    A3D_CHECK_ARG(flow && flow_gt && both && scratch && loss && scale && B > 0 && F >= 2 && H > 0 && W > 0 && pix_stride >= 2);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = a3d_div_up((long long)H * W, LS_BLOCK), pairs = B * (F - 1);
    hipLaunchKernelGGL(fl_fwd_kernel, dim3(nblk, pairs), dim3(LS_BLOCK), 0, s, flow, pix_stride, flow_gt, both, F, H * W, (float*)scratch);
    A3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(fl_finish_kernel, dim3(pairs), dim3(64), 0, s, (const float*)scratch, nblk, loss, scale);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_flow_loss_bwd(const float* g_loss, const float* scale, const float* flow, int pix_stride, const float* flow_gt,
                                 const uint8_t* both, int B, int F, int H, int W, float* g_flow, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_loss && scale && flow && flow_gt && both && g_flow && B > 0 && F >= 2 && H > 0 && W > 0 && pix_stride >= 2);
    hipLaunchKernelGGL(fl_bwd_kernel, dim3(a3d_div_up((long long)H * W, LS_BLOCK), B * F), dim3(LS_BLOCK), 0, (hipStream_t)stream, g_loss, scale, flow,
                       pix_stride, flow_gt, both, F, H * W, g_flow);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
