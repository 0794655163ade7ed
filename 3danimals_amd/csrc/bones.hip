// Per-bone world transforms from the kinematic chain on gfx950: one work-group per instance, links built once in LDS.
//
// Replaces the inner loops of skinning() (model/geometry/skinning.py:389-417): for every bone the reference walks its
// kinematic chain leaf -> root composing Rest_i . Rot(euler_i) . Rest_i^-1 one 4x4 torch op at a time (~24k aten calls
// per forward at K=20, B=16).  Here the K links of an instance are built once and bone k multiplies the <= 8 links of its chain:
//     L_i = [ R_i Rot_i R_i^T | t_i - R_i Rot_i R_i^T t_i ],   M_k = L_root ... L_parent(k) L_k        (3x4 affine)
// with R_i the rest frame from the bone direction (columns right, up, forward; right ~ +x; skinning.py:251-270), t_i the
// bone's start joint and Rot_i = Rx Ry Rz (PyTorch3D 'XYZ', skinning.py:285-340).
// Backward: g_L_j = P_j^T g_M S_j^T with prefix/suffix products of the chain, pushed through the conjugation and the Euler
// factors onto the three angles (summed per link in a fixed order: bit-reproducible).  Bones carry no gradient.
// A few hundred threads in total; the point is 2 launches instead of ~80 tiny ones on a host-bound stretch of the step.
#include "bones_common.h"

// ---- one work-group per instance: the K link matrices are built ONCE (one thread each) into LDS, then every bone walks its
// chain over LDS copies (<= 8 products of 3x4 affines).  The first version rebuilt every link of every chain per thread
// (8 x sincos/normalise chains in series, 13 us / 44 us for 320 threads of work); this form is a few hundred flops deep.
#define BN_THREADS 256

__global__ __launch_bounds__(BN_THREADS) void bn_fwd_kernel(const float* __restrict__ bones, int bones_batch, const float* __restrict__ angles,
                                                            const int* __restrict__ chain, int K, int D, float* __restrict__ M) {
    __shared__ float s_L[BN_MAXK][13];  // 13: odd stride, conflict-free row access
    __shared__ int s_chain[BN_MAXK * BN_MAXD];  // a serial walk over global memory costs ~0.2 us per dependent load
    const int n = blockIdx.x;
    const float* bb = bones + (bones_batch == 1 ? 0ll : (long long)n * K * 6);
    const float* aa = angles + (long long)n * K * 3;
    for (int w = threadIdx.x; w < K * D; w += blockDim.x) s_chain[w] = chain[w];
    for (int i = threadIdx.x; i < K; i += blockDim.x) bn_store(s_L[i], bn_link(bb + 6 * i, aa + 3 * i));
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        A34 acc = bn_identity();
        for (int j = 0; j < D; ++j) {
            const int i = s_chain[k * D + j];
            if (i >= 0) acc = bn_mul(acc, bn_load(s_L[i]));
        }
        bn_store(M + ((long long)n * K + k) * 12, acc);
    }
}

// backward: one work-group per instance runs bn_chain_adjoint (bones_common.h)
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_kernel(const float* __restrict__ g_M, const float* __restrict__ bones, int bones_batch,
                                                            const float* __restrict__ angles, const int* __restrict__ chain, int K, int D,
                                                            float* __restrict__ g_angles) {
    extern __shared__ float s_dyn[];
    const int n = blockIdx.x;
    bn_chain_adjoint<false>(g_M + (long long)n * K * 12, nullptr, bones + (bones_batch == 1 ? 0ll : (long long)n * K * 6),
                            angles + (long long)n * K * 3, chain, K, D, g_angles + (long long)n * K * 3, s_dyn);
}

extern "C" int a3d_bone_transforms_fwd(const float* bones, int bones_batch, const float* angles, const int32_t* chain, int N, int K, int D,
                                       float* M, a3d_stream_t stream) {
    A3D_CHECK_ARG(bones && angles && chain && M && N > 0 && K > 0 && K <= BN_MAXK && D > 0 && D <= BN_MAXD);
    A3D_CHECK_ARG(bones_batch == 1 || bones_batch == N);
    hipLaunchKernelGGL(bn_fwd_kernel, dim3(N), dim3(BN_THREADS), 0, (hipStream_t)stream, bones, bones_batch, angles, chain, K, D, M);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_bone_transforms_bwd(const float* g_M, const float* bones, int bones_batch, const float* angles, const int32_t* chain, int N,
                                       int K, int D, float* g_angles, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_M && bones && angles && chain && g_angles && N > 0 && K > 0 && K <= BN_MAXK && D > 0 && D <= BN_MAXD);
    A3D_CHECK_ARG(bones_batch == 1 || bones_batch == N);
    // every g_angles element is written by its owner thread: no memset, no atomics
    if (bn_bwd_lds(K, D) > 64 * 1024)  // K > ~48 with the deepest chains: opt in to the large-LDS launch (cheap, per call: no cached state)
        A3D_HIP(hipFuncSetAttribute((const void*)bn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bn_bwd_lds(K, D)));
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(N), dim3(BN_THREADS), bn_bwd_lds(K, D), (hipStream_t)stream, g_M, bones, bones_batch, angles, chain, K,
                       D, g_angles);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
