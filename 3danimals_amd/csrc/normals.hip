// Area-weighted vertex normals on gfx950 -- replaces auto_normals (model/render/mesh.py:276-304).
//
// Gather formulation, no float atomics, bit-reproducible:
//   adjacency (once per triangle list, shared by the rest and the posed mesh and by every image of the batch):
//       CSR vertex -> incident (corner, face) entries, each list sorted by corner-major key c*F+f, i.e. the order in which the
//       reference's three scatter_add_ passes (mesh.py:291-293) visit them.
//   fwd : one thread per (image, vertex) walks its list, recomputes the face's cross product (un-normalised == area weighted,
//         mesh.py:285) and sums in list order; zero sums -> (0,0,1) (mesh.py:296-298), safe_normalize (render/util.py:28-32).
//   bwd : per vertex the normalisation Jacobian (prepass), then one thread per (image, vertex) gathers the cross-product
//         adjoint of each incident face for ITS corner -- every g_v element is written once.
// The reference materialises three [B,F,3] gathers, a [B,F,3] cross product and three index.repeat(B,1,3) int64 tensors per
// call; here 16 B/entry of indices (L2 resident, shared over the batch) + 36 B/entry position gathers + 24 B/vertex.
#include "a3d_common.h"
#include "topo_common.h"

namespace {

__global__ __launch_bounds__(256) void nr_adj_count_kernel(const int* __restrict__ tri, int n3, int* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) atomicAdd(cnt + tri[i], 1);
}

__global__ __launch_bounds__(256) void nr_adj_fill_kernel(const int* __restrict__ tri, int F, const int* __restrict__ off,
                                                          int* __restrict__ cursor, int* __restrict__ adj) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * F) return;
    const int f = i / 3, c = i - 3 * f, v = tri[i];
    adj[off[v] + atomicAdd(cursor + v, 1)] = c * F + f;
}

struct NrFace { int i0, i1, i2, c; };

__device__ __forceinline__ NrFace nr_decode(int key, int F, const int* __restrict__ tri) {
    NrFace r;
    r.c = key >= 2 * F ? 2 : (key >= F ? 1 : 0);
    const int f = key - r.c * F;
    r.i0 = tri[3 * f]; r.i1 = tri[3 * f + 1]; r.i2 = tri[3 * f + 2];
    return r;
}

__global__ __launch_bounds__(256) void nr_fwd_kernel(const float* __restrict__ v, const int* __restrict__ tri, const int* __restrict__ off,
                                                     const int* __restrict__ adj, int V, int F, float* __restrict__ acc,
                                                     float* __restrict__ nrm) {
    const int vi = blockIdx.x * blockDim.x + threadIdx.x;
    if (vi >= V) return;
    const long long vb = (long long)blockIdx.y * V;
    const float* vp = v + vb * 3;
    float x = 0.f, y = 0.f, z = 0.f;
    const int hi = off[vi + 1];
    for (int e = off[vi]; e < hi; ++e) {
        const NrFace t = nr_decode(adj[e], F, tri);
        const float* p0 = vp + 3ll * t.i0;
        const float* p1 = vp + 3ll * t.i1;
        const float* p2 = vp + 3ll * t.i2;
        const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
        const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
        x += ay * bz - az * by; y += az * bx - ax * bz; z += ax * by - ay * bx;
    }
    const long long o = (vb + vi) * 3;
    acc[o] = x; acc[o + 1] = y; acc[o + 2] = z;
    float d = x * x + y * y + z * z;
    if (!(d > 1e-20f)) { x = 0.f; y = 0.f; z = 1.f; d = 1.f; }
    const float len = sqrtf(fmaxf(d, 1e-20f));  // x / sqrt(clamp(dot, 1e-20)) as render/util.py:28-32 writes it
    nrm[o] = x / len; nrm[o + 1] = y / len; nrm[o + 2] = z / len;
}

// d(normalize(acc))/d(acc) applied to g_nrm; zero where the default normal was substituted
__global__ __launch_bounds__(256) void nr_vert_bwd_kernel(const float* __restrict__ g_nrm, int g_stride, const float* __restrict__ acc,
                                                          long long n, float* __restrict__ g_acc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = acc[3 * i], y = acc[3 * i + 1], z = acc[3 * i + 2];
    const float d = x * x + y * y + z * z;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (d > 1e-20f) {
        const float inv = 1.f / sqrtf(d);
        const float nx = x * inv, ny = y * inv, nz = z * inv;
        const float* gr = g_nrm + (long long)g_stride * i;
        const float gx = gr[0], gy = gr[1], gz = gr[2];
        const float dot = nx * gx + ny * gy + nz * gz;
        ox = (gx - nx * dot) * inv; oy = (gy - ny * dot) * inv; oz = (gz - nz * dot) * inv;
    }
    g_acc[3 * i] = ox; g_acc[3 * i + 1] = oy; g_acc[3 * i + 2] = oz;
}

__global__ __launch_bounds__(256) void nr_bwd_kernel(const float* __restrict__ g_acc, const float* __restrict__ v, const int* __restrict__ tri,
                                                     const int* __restrict__ off, const int* __restrict__ adj, int V, int F,
                                                     float* __restrict__ g_v) {
    const int vi = blockIdx.x * blockDim.x + threadIdx.x;
    if (vi >= V) return;
    const long long vb = (long long)blockIdx.y * V;
    const float* vp = v + vb * 3;
    const float* gp = g_acc + vb * 3;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    const int hi = off[vi + 1];
    for (int e = off[vi]; e < hi; ++e) {
        const NrFace t = nr_decode(adj[e], F, tri);
        const float* g0 = gp + 3ll * t.i0;
        const float* g1 = gp + 3ll * t.i1;
        const float* g2 = gp + 3ll * t.i2;
        const float gx = g0[0] + g1[0] + g2[0], gy = g0[1] + g1[1] + g2[1], gz = g0[2] + g1[2] + g2[2];
        const float* p0 = vp + 3ll * t.i0;
        const float* p1 = vp + 3ll * t.i1;
        const float* p2 = vp + 3ll * t.i2;
        const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
        const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
        // n = a x b :  g_a = b x g_n  (-> corner 1),  g_b = g_n x a  (-> corner 2),  corner 0 gets -(g_a + g_b)
        const float gax = by * gz - bz * gy, gay = bz * gx - bx * gz, gaz = bx * gy - by * gx;
        const float gbx = gy * az - gz * ay, gby = gz * ax - gx * az, gbz = gx * ay - gy * ax;
        if (t.c == 1) { ox += gax; oy += gay; oz += gaz; }
        else if (t.c == 2) { ox += gbx; oy += gby; oz += gbz; }
        else { ox -= gax + gbx; oy -= gay + gby; oz -= gaz + gbz; }
    }
    const long long o = (vb + vi) * 3;
    g_v[o] = ox; g_v[o + 1] = oy; g_v[o + 2] = oz;
}

}  // namespace

extern "C" int a3d_normals_adjacency(const int32_t* tri, int V, int F, int32_t* off, int32_t* adj, int32_t* cursor, a3d_stream_t stream) {
    A3D_CHECK_ARG(off && cursor && V > 0 && F >= 0 && (long long)3 * F < 0x7fffffffll);
    A3D_CHECK_ARG(F == 0 || (tri && adj));
    hipStream_t s = (hipStream_t)stream;
    A3D_HIP(hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t)V, s));
    if (F > 0) {
        hipLaunchKernelGGL(nr_adj_count_kernel, dim3(a3d_div_up(3 * F, 256)), dim3(256), 0, s, tri, 3 * F, cursor);
        A3D_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(nr_adj_scan_kernel, dim3(1), dim3(NR_SCAN_THREADS), 0, s, cursor, V, off);
    A3D_LAUNCH_CHECK();
    if (F > 0) {
        hipLaunchKernelGGL(nr_adj_fill_kernel, dim3(a3d_div_up(3 * F, 256)), dim3(256), 0, s, tri, F, off, cursor, adj);
        A3D_LAUNCH_CHECK();
        hipLaunchKernelGGL(nr_adj_sort_kernel, dim3(a3d_div_up(V, 256)), dim3(256), 0, s, off, V, adj);
        A3D_LAUNCH_CHECK();
    }
    return A3D_OK;
}

extern "C" int a3d_normals_fwd(const float* v, const int32_t* tri, const int32_t* off, const int32_t* adj, int B, int V, int F, float* acc,
                               float* nrm, a3d_stream_t stream) {
    A3D_CHECK_ARG(v && off && acc && nrm && B > 0 && V > 0 && F >= 0);
    A3D_CHECK_ARG(F == 0 || (tri && adj));
    hipLaunchKernelGGL(nr_fwd_kernel, dim3(a3d_div_up(V, 256), B), dim3(256), 0, (hipStream_t)stream, v, tri, off, adj, V, F, acc, nrm);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_normals_bwd(const float* g_nrm, int g_nrm_stride, const float* acc, const float* v, const int32_t* tri, const int32_t* off,
                               const int32_t* adj, int B, int V, int F, float* g_acc_scratch, float* g_v, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_nrm && g_nrm_stride >= 3 && acc && v && off && g_acc_scratch && g_v && B > 0 && V > 0 && F >= 0);
    A3D_CHECK_ARG(F == 0 || (tri && adj));
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)B * V;
    hipLaunchKernelGGL(nr_vert_bwd_kernel, dim3(a3d_div_up(n, 256)), dim3(256), 0, s, g_nrm, g_nrm_stride, acc, n, g_acc_scratch);
    A3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(nr_bwd_kernel, dim3(a3d_div_up(V, 256), B), dim3(256), 0, s, g_acc_scratch, v, tri, off, adj, V, F, g_v);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
