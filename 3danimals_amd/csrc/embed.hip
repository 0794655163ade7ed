// Harmonic (positional) embedding of the field inputs on gfx950 -- /root/reference/model/networks/HarmonicEmbedding.py:33-44 as
// CoordMLP uses it (networks/MLPs.py:76-83): out[p] = [x, sin(x_c * f_k), cos(x_c * f_k)] with the (c, k) pairs c-major,
// optionally |x_0| first (the symmetrize option, MLPs.py:73-74) and optionally a trailing column of ones, which lets the caller
// fold the first Linear's bias into its weight: K = 3 + 6n + 1 = 64 for n = 10, and the bias gradient falls out of the weight-
// gradient GEMM.  The reference runs abs / cat / mul / sin / cos / cat / cat (7 launches over [P, <=63]) forward and ~10 backward.
//   fwd: one thread per output element (coalesced stores; x rows come from L1/L2)
//   bwd: one thread per (point, coordinate): g_x = g[x part] + sum_k f_k (g_sin cos(x f_k) - g_cos sin(x f_k)), recomputing sin/cos
#include "a3d_common.h"

namespace {

// one thread per (point, slot): slots 0..3n-1 are the (coordinate, frequency) pairs -- one sincosf, two stores -- and the last
// 3 (+1) slots copy x (and write the ones column)
__global__ __launch_bounds__(256) void he_fwd_kernel(const float* __restrict__ x, const float* __restrict__ freq, int n, int symmetrize,
                                                     int ones, long long total, int C, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int slots = 3 * n + 3 + ones;
    const long long p = i / slots;
    const int q = (int)(i - p * slots);
    float* o = out + p * C;
    if (q < 3 * n) {
        const int c = q / n, k = q - c * n;
        float xv = x[3 * p + c];
        if (symmetrize && c == 0) xv = fabsf(xv);
        float sn, cs;
        sincosf(xv * freq[k], &sn, &cs);
        o[3 + q] = sn;
        o[3 + 3 * n + q] = cs;
    } else if (q < 3 * n + 3) {
        const int j = q - 3 * n;
        const float v = x[3 * p + j];
        o[j] = (symmetrize && j == 0) ? fabsf(v) : v;
    } else {
        o[C - 1] = 1.f;
    }
}

__global__ __launch_bounds__(256) void he_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ freq,
                                                     int n, int symmetrize, long long P, int C, float* __restrict__ g_x) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * P) return;
    const long long p = i / 3;
    const int c = (int)(i - 3 * p);
    const float raw = x[i];
    const float xv = (symmetrize && c == 0) ? fabsf(raw) : raw;
    const float* gp = g + p * C;
    float acc = gp[c];
    const float* gs = gp + 3 + c * n;
    const float* gc = gs + 3 * n;
    for (int k = 0; k < n; ++k) {
        const float f = freq[k];
        float sn, cs;
        sincosf(xv * f, &sn, &cs);
        acc += f * (gs[k] * cs - gc[k] * sn);
    }
    if (symmetrize && c == 0) acc *= raw > 0.f ? 1.f : (raw < 0.f ? -1.f : 0.f);  // d|x|/dx, 0 at 0 like torch.abs
    g_x[i] = acc;
}

}  // namespace

extern "C" int a3d_harmonic_embed_fwd(const float* x, const float* freq, int n, int symmetrize, int ones, int64_t P, float* out,
                                      a3d_stream_t stream) {
    A3D_CHECK_ARG(P >= 0 && n > 0 && (ones == 0 || ones == 1));
    if (P == 0) return A3D_OK;
    A3D_CHECK_ARG(x && freq && out);
    const int C = 3 + 6 * n + ones;
    const long long total = (long long)P * (3 * n + 3 + ones);
    hipLaunchKernelGGL(he_fwd_kernel, dim3(a3d_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, x, freq, n, symmetrize, ones, total, C, out);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_harmonic_embed_bwd(const float* g_out, const float* x, const float* freq, int n, int symmetrize, int ones, int64_t P,
                                      float* g_x, a3d_stream_t stream) {
    A3D_CHECK_ARG(P >= 0 && n > 0 && (ones == 0 || ones == 1));
    if (P == 0) return A3D_OK;
    A3D_CHECK_ARG(g_out && x && freq && g_x);
    hipLaunchKernelGGL(he_bwd_kernel, dim3(a3d_div_up(3 * (long long)P, 256)), dim3(256), 0, (hipStream_t)stream, g_out, x, freq, n, symmetrize,
                       (long long)P, 3 + 6 * n + ones, g_x);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
