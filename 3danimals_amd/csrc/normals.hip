// Area-weighted vertex normals on gfx950 -- replaces auto_normals (model/render/mesh.py:276-304).
//
// fwd : one thread per (image, face): cross product of two edge vectors (un-normalised == area weighted,
//       mesh.py:285) scattered onto the three corners with float atomics; then one thread per (image, vertex):
//       zero sums -> (0,0,1) (mesh.py:296-298) and safe_normalize (render/util.py:28-32).
// bwd : per vertex the normalisation Jacobian, per face a 3-way GATHER of those (no atomics on the way in),
//       the cross-product adjoint, and atomics onto the vertex positions.
// The reference materialises three [B,F,3] gathers, a [B,F,3] cross product and three index.repeat(B,1,3)
// int64 tensors per call; here: 12 B/face of indices (shared over the batch) + 36 B/face gathers + 24 B/vertex.
#include "a3d_common.h"

__global__ __launch_bounds__(256) void nr_face_fwd_kernel(const float* __restrict__ v, const int* __restrict__ tri, int V, int F,
                                                          float* __restrict__ acc) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const long long vb = (long long)blockIdx.y * V;
    const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    const float* p0 = v + (vb + i0) * 3;
    const float* p1 = v + (vb + i1) * 3;
    const float* p2 = v + (vb + i2) * 3;
    const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
    const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
    const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    float* a0 = acc + (vb + i0) * 3;
    float* a1 = acc + (vb + i1) * 3;
    float* a2 = acc + (vb + i2) * 3;
    atomicAdd(a0, nx); atomicAdd(a0 + 1, ny); atomicAdd(a0 + 2, nz);
    atomicAdd(a1, nx); atomicAdd(a1 + 1, ny); atomicAdd(a1 + 2, nz);
    atomicAdd(a2, nx); atomicAdd(a2 + 1, ny); atomicAdd(a2 + 2, nz);
}

__global__ __launch_bounds__(256) void nr_vert_fwd_kernel(const float* __restrict__ acc, long long n, float* __restrict__ nrm) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = acc[3 * i], y = acc[3 * i + 1], z = acc[3 * i + 2];
    float d = x * x + y * y + z * z;
    if (!(d > 1e-20f)) { x = 0.f; y = 0.f; z = 1.f; d = 1.f; }
    const float inv = 1.f / sqrtf(fmaxf(d, 1e-20f));
    nrm[3 * i] = x * inv; nrm[3 * i + 1] = y * inv; nrm[3 * i + 2] = z * inv;
}

// d(normalize(acc))/d(acc) applied to g_nrm; zero where the default normal was substituted
__global__ __launch_bounds__(256) void nr_vert_bwd_kernel(const float* __restrict__ g_nrm, const float* __restrict__ acc, long long n,
                                                          float* __restrict__ g_acc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = acc[3 * i], y = acc[3 * i + 1], z = acc[3 * i + 2];
    const float d = x * x + y * y + z * z;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (d > 1e-20f) {
        const float inv = 1.f / sqrtf(d);
        const float nx = x * inv, ny = y * inv, nz = z * inv;
        const float gx = g_nrm[3 * i], gy = g_nrm[3 * i + 1], gz = g_nrm[3 * i + 2];
        const float dot = nx * gx + ny * gy + nz * gz;
        ox = (gx - nx * dot) * inv; oy = (gy - ny * dot) * inv; oz = (gz - nz * dot) * inv;
    }
    g_acc[3 * i] = ox; g_acc[3 * i + 1] = oy; g_acc[3 * i + 2] = oz;
}

__global__ __launch_bounds__(256) void nr_face_bwd_kernel(const float* __restrict__ g_acc, const float* __restrict__ v,
                                                          const int* __restrict__ tri, int V, int F, float* __restrict__ g_v) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const long long vb = (long long)blockIdx.y * V;
    const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    const float* g0 = g_acc + (vb + i0) * 3;
    const float* g1 = g_acc + (vb + i1) * 3;
    const float* g2 = g_acc + (vb + i2) * 3;
    const float gx = g0[0] + g1[0] + g2[0], gy = g0[1] + g1[1] + g2[1], gz = g0[2] + g1[2] + g2[2];
    if (gx == 0.f && gy == 0.f && gz == 0.f) return;
    const float* p0 = v + (vb + i0) * 3;
    const float* p1 = v + (vb + i1) * 3;
    const float* p2 = v + (vb + i2) * 3;
    const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
    const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
    // n = a x b :  g_a = b x g_n,  g_b = g_n x a
    const float gax = by * gz - bz * gy, gay = bz * gx - bx * gz, gaz = bx * gy - by * gx;
    const float gbx = gy * az - gz * ay, gby = gz * ax - gx * az, gbz = gx * ay - gy * ax;
    float* o0 = g_v + (vb + i0) * 3;
    float* o1 = g_v + (vb + i1) * 3;
    float* o2 = g_v + (vb + i2) * 3;
    atomicAdd(o1, gax); atomicAdd(o1 + 1, gay); atomicAdd(o1 + 2, gaz);
    atomicAdd(o2, gbx); atomicAdd(o2 + 1, gby); atomicAdd(o2 + 2, gbz);
    atomicAdd(o0, -(gax + gbx)); atomicAdd(o0 + 1, -(gay + gby)); atomicAdd(o0 + 2, -(gaz + gbz));
}

extern "C" int a3d_normals_fwd(const float* v, const int32_t* tri, int B, int V, int F, float* acc, float* nrm, a3d_stream_t stream) {
    A3D_CHECK_ARG(v && acc && nrm && B > 0 && V > 0 && F >= 0);
    A3D_CHECK_ARG(F == 0 || tri);
    hipStream_t s = (hipStream_t)stream;
    A3D_HIP(hipMemsetAsync(acc, 0, sizeof(float) * 3 * (size_t)B * V, s));
    if (F > 0) {
        hipLaunchKernelGGL(nr_face_fwd_kernel, dim3(a3d_div_up(F, 256), B), dim3(256), 0, s, v, tri, V, F, acc);
        A3D_LAUNCH_CHECK();
    }
    const long long n = (long long)B * V;
    hipLaunchKernelGGL(nr_vert_fwd_kernel, dim3(a3d_div_up(n, 256)), dim3(256), 0, s, acc, n, nrm);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_normals_bwd(const float* g_nrm, const float* acc, const float* v, const int32_t* tri, int B, int V, int F,
                               float* g_acc_scratch, float* g_v, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_nrm && acc && v && g_acc_scratch && g_v && B > 0 && V > 0 && F >= 0);
    A3D_CHECK_ARG(F == 0 || tri);
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)B * V;
    A3D_HIP(hipMemsetAsync(g_v, 0, sizeof(float) * 3 * (size_t)n, s));
    hipLaunchKernelGGL(nr_vert_bwd_kernel, dim3(a3d_div_up(n, 256)), dim3(256), 0, s, g_nrm, acc, n, g_acc_scratch);
    A3D_LAUNCH_CHECK();
    if (F > 0) {
        hipLaunchKernelGGL(nr_face_bwd_kernel, dim3(a3d_div_up(F, 256), B), dim3(256), 0, s, g_acc_scratch, v, tri, V, F, g_v);
        A3D_LAUNCH_CHECK();
    }
    return A3D_OK;
}
