// Area-weighted vertex normals on gfx950 -- replaces auto_normals (model/render/mesh.py:276-304).
//
// Gather formulation, no float atomics, bit-reproducible:
//   adjacency (once per triangle list, shared by the rest and the posed mesh and by every image of the batch):
//       CSR vertex -> incident (corner, face) entries with corner-major keys c*F+f; the kernels below visit a list in ascending key
//       order -- the order in which the reference's three scatter_add_ passes (mesh.py:291-293) visit them -- whether it is stored
//       sorted (a3d_normals_adjacency, a3d_mesh_topology) or not (the lists the DMTet emit + a3d_mesh_topology_finalize build).
//   fwd : one thread per (image, vertex) walks its list, recomputes the face's cross product (un-normalised == area weighted,
//         mesh.py:285) and sums in list order; zero sums -> (0,0,1) (mesh.py:296-298), safe_normalize (render/util.py:28-32).
//   bwd : per vertex the normalisation Jacobian (prepass), then one thread per (image, vertex) gathers the cross-product
//         adjoint of each incident face for ITS corner -- every g_v element is written once.
// The reference materialises three [B,F,3] gathers, a [B,F,3] cross product and three index.repeat(B,1,3) int64 tensors per
// call; here 16 B/entry of indices (L2 resident, shared over the batch) + 36 B/entry position gathers + 24 B/vertex.
#include "normals_common.h"

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

// (v2 / acc2 / nrm2: a SECOND vertex array over the same triangle list -- images B1, B1+1, ... of the launch -- so that e.g. the one
// canonical mesh does not cost a launch of its own beside the B posed ones: a3d_normals_fwd_pair)
__global__ __launch_bounds__(256) void nr_fwd_kernel(const float* __restrict__ v, const int* __restrict__ tri, const int* __restrict__ off,
                                                     const int* __restrict__ adj, int V, int F, float* __restrict__ acc,
                                                     float* __restrict__ nrm, int B1, const float* __restrict__ v2,
                                                     float* __restrict__ acc2, float* __restrict__ nrm2, int stride) {
    const int vi = blockIdx.x * blockDim.x + threadIdx.x;
    if (vi >= V) return;
    long long vb = (long long)blockIdx.y * V;
    if ((int)blockIdx.y >= B1) {
        v = v2; acc = acc2; nrm = nrm2;
        vb = (long long)((int)blockIdx.y - B1) * V;
    }
    nr_fwd_vertex<NR_SLOTS>(v + vb * 3, tri, off, adj, stride, F, vi, acc, nrm, (vb + vi) * 3);
}

// d(normalize(acc))/d(acc) applied to g_nrm; zero where the default normal was substituted
__device__ __forceinline__ void nr_vert_adjoint(float x, float y, float z, float gx, float gy, float gz, float o[3]) {
    const float d = x * x + y * y + z * z;
    o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
    if (d > 1e-20f) {
        const float inv = 1.f / sqrtf(d);
        const float nx = x * inv, ny = y * inv, nz = z * inv;
        const float dot = nx * gx + ny * gy + nz * gz;
        o[0] = (gx - nx * dot) * inv; o[1] = (gy - ny * dot) * inv; o[2] = (gz - nz * dot) * inv;
    }
}

__global__ __launch_bounds__(256) void nr_vert_bwd_kernel(const float* __restrict__ g_nrm, int g_stride, const float* __restrict__ acc,
                                                          long long n, float* __restrict__ g_acc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* gr = g_nrm + (long long)g_stride * i;
    float o[3];
    nr_vert_adjoint(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2], gr[0], gr[1], gr[2], o);
    g_acc[3 * i] = o[0]; g_acc[3 * i + 1] = o[1]; g_acc[3 * i + 2] = o[2];
}

__device__ __forceinline__ void nr_bwd_entry(int c, const float g0[3], const float g1[3], const float g2[3], const float p0[3], const float p1[3],
                                             const float p2[3], float& ox, float& oy, float& oz) {
    const float gx = g0[0] + g1[0] + g2[0], gy = g0[1] + g1[1] + g2[1], gz = g0[2] + g1[2] + g2[2];
    const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
    const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
    // n = a x b :  g_a = b x g_n  (-> corner 1),  g_b = g_n x a  (-> corner 2),  corner 0 gets -(g_a + g_b)
    const float gax = by * gz - bz * gy, gay = bz * gx - bx * gz, gaz = bx * gy - by * gx;
    const float gbx = gy * az - gz * ay, gby = gz * ax - gx * az, gbz = gx * ay - gy * ax;
    if (c == 1) { ox += gax; oy += gay; oz += gaz; }
    else if (c == 2) { ox += gbx; oy += gby; oz += gbz; }
    else { ox -= gax + gbx; oy -= gay + gby; oz -= gaz + gbz; }
}

// FUSED: no prepass -- the normalisation adjoint of every gathered corner is recomputed from (acc, g_nrm) on the spot (same formula,
// same bits): 24 B gathered per corner instead of 12 and ~24 adjoints per thread instead of one, for one launch less on a stretch
// where a launch costs more than the arithmetic (the prepass: 4.9 us of kernel + a gap for 93k threads of work)
template <bool FUSED>
__global__ __launch_bounds__(256) void nr_bwd_kernel(const float* __restrict__ g_acc, const float* __restrict__ v, const int* __restrict__ tri,
                                                     const int* __restrict__ off, const int* __restrict__ adj, int V, int F,
                                                     float* __restrict__ g_v, int stride, const float* __restrict__ g_nrm, int g_stride,
                                                     const float* __restrict__ acc) {
    const int vi = blockIdx.x * blockDim.x + threadIdx.x;
    if (vi >= V) return;
    const long long vb = (long long)blockIdx.y * V;
    const float* vp = v + vb * 3;
    const float* gp = FUSED ? nullptr : g_acc + vb * 3;
    const float* ap = FUSED ? acc + vb * 3 : nullptr;
    const float* np = FUSED ? g_nrm + vb * g_stride : nullptr;
    auto corner = [&](int i, float o[3]) {
        if (FUSED) {
            const float* gr = np + (long long)g_stride * i;
            nr_vert_adjoint(ap[3ll * i], ap[3ll * i + 1], ap[3ll * i + 2], gr[0], gr[1], gr[2], o);
        } else {
            o[0] = gp[3ll * i]; o[1] = gp[3ll * i + 1]; o[2] = gp[3ll * i + 2];
        }
    };
    float ox = 0.f, oy = 0.f, oz = 0.f;
    int lo, cnt;
    vf_list(off, stride, vi, lo, cnt);
    if (cnt > 0) {
    int keys[NR_SLOTS];
    nr_first_keys(adj, lo, cnt, keys);
    NrFace t[NR_SLOTS];
#pragma unroll
    for (int k = 0; k < NR_SLOTS; ++k) t[k] = nr_decode(k < cnt ? keys[k] : keys[0], F, tri);
    // two half-batches of four entries: 18 gathered floats per entry (positions + adjoints of the three corners) in flight at once
#pragma unroll
    for (int h = 0; h < NR_SLOTS; h += 4) {
        float p[4][9], g[4][9];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i0 = t[h + k].i0, i1 = t[h + k].i1, i2 = t[h + k].i2;
#pragma unroll
            for (int q = 0; q < 3; ++q) { p[k][q] = vp[3ll * i0 + q]; p[k][3 + q] = vp[3ll * i1 + q]; p[k][6 + q] = vp[3ll * i2 + q]; }
            corner(i0, g[k]); corner(i1, g[k] + 3); corner(i2, g[k] + 6);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (h + k < cnt) nr_bwd_entry(t[h + k].c, g[k], g[k] + 3, g[k] + 6, p[k], p[k] + 3, p[k] + 6, ox, oy, oz);
    }
    int last = keys[NR_SLOTS - 1];
    for (int e = NR_SLOTS; e < cnt; ++e) {
        last = nr_next_key_mem(adj, lo, cnt, last);
        const NrFace tt = nr_decode(last, F, tri);
        float p[9], g[9];
#pragma unroll
        for (int q = 0; q < 3; ++q) { p[q] = vp[3ll * tt.i0 + q]; p[3 + q] = vp[3ll * tt.i1 + q]; p[6 + q] = vp[3ll * tt.i2 + q]; }
        corner(tt.i0, g); corner(tt.i1, g + 3); corner(tt.i2, g + 6);
        nr_bwd_entry(tt.c, g, g + 3, g + 6, p, p + 3, p + 6, ox, oy, oz);
    }
    }
    const long long o = (vb + vi) * 3;
    g_v[o] = ox; g_v[o + 1] = oy; g_v[o + 2] = oz;
}

// ---- faces first (round 4).  The kernel above re-derives the adjoint of every incident face at every vertex: valence x (3 positions + 3
// normal adjoints) = ~48 gathers per vertex, i.e. ~24 per face, and on a mesh whose numbering is not spatial (a surface extracted from a
// BCC lattice or from a scrambled grid file) every one of them is a cache line of its own: 11 -> 21 us.  Here every (image, face) does that
// work ONCE -- 9 gathers: positions, acc and g_nrm of its three corners; the normalisation adjoint of the pre-pass folded in, three per
// thread -- and leaves the three corner contributions (36 B) in a face-major scratch; every (image, vertex) then sums its <= 8 entries in
// ascending key order: the SAME float values in the SAME order as the gather kernel adds them, so the result is bit-identical.
__global__ __launch_bounds__(256) void nr_face_bwd_kernel(const float* __restrict__ g_nrm, int g_stride, const float* __restrict__ acc,
                                                          const float* __restrict__ v, const int* __restrict__ tri, int V, int F,
                                                          float* __restrict__ fadj) {
    A3D_STAMP(0, 0);  // (A3D_STAMP kernel ids of this file: 0 = nr_face_bwd_kernel, 1 = nr_sum_bwd_kernel)
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const long long vb = (long long)blockIdx.y * V;
    const float* vp = v + vb * 3;
    const float* ap = acc + vb * 3;
    const float* np = g_nrm + vb * g_stride;
    const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    float p0[3], p1[3], p2[3], g0[3], g1[3], g2[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { p0[q] = vp[3ll * i0 + q]; p1[q] = vp[3ll * i1 + q]; p2[q] = vp[3ll * i2 + q]; }
    {
        const float *a0 = ap + 3ll * i0, *a1 = ap + 3ll * i1, *a2 = ap + 3ll * i2;
        const float *n0 = np + (long long)g_stride * i0, *n1 = np + (long long)g_stride * i1, *n2 = np + (long long)g_stride * i2;
        const float x0 = a0[0], y0 = a0[1], z0 = a0[2], x1 = a1[0], y1 = a1[1], z1 = a1[2], x2 = a2[0], y2 = a2[1], z2 = a2[2];
        const float gx0 = n0[0], gy0 = n0[1], gz0 = n0[2], gx1 = n1[0], gy1 = n1[1], gz1 = n1[2], gx2 = n2[0], gy2 = n2[1], gz2 = n2[2];
        nr_vert_adjoint(x0, y0, z0, gx0, gy0, gz0, g0);
        nr_vert_adjoint(x1, y1, z1, gx1, gy1, gz1, g1);
        nr_vert_adjoint(x2, y2, z2, gx2, gy2, gz2, g2);
    }
    float* o = fadj + ((long long)blockIdx.y * F + f) * 9;
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // corner c's share, by the gather kernel's own function (same operations, same order)
        float ox = 0.f, oy = 0.f, oz = 0.f;
        nr_bwd_entry(c, g0, g1, g2, p0, p1, p2, ox, oy, oz);
        o[3 * c] = ox; o[3 * c + 1] = oy; o[3 * c + 2] = oz;
    }
    A3D_STAMP(0, 5);
}

__global__ __launch_bounds__(256) void nr_sum_bwd_kernel(const float* __restrict__ fadj, const int* __restrict__ off, const int* __restrict__ adj,
                                                         int V, int F, float* __restrict__ g_v, int stride) {
    A3D_STAMP(1, 0);
    const int vi = blockIdx.x * blockDim.x + threadIdx.x;
    if (vi >= V) return;
    const float* fa = fadj + (long long)blockIdx.y * F * 9;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    int lo, cnt;
    vf_list(off, stride, vi, lo, cnt);
    if (cnt > 0) {
        int keys[NR_SLOTS];
        nr_first_keys(adj, lo, cnt, keys);
        float e[NR_SLOTS][3];
#pragma unroll
        for (int k = 0; k < NR_SLOTS; ++k) {  // (all entries in flight; unused slots re-read the first)
            const int key = k < cnt ? keys[k] : keys[0];
            const int c = key >= 2 * F ? 2 : (key >= F ? 1 : 0);
            const float* s = fa + 9ll * (key - c * F) + 3 * c;
            e[k][0] = s[0]; e[k][1] = s[1]; e[k][2] = s[2];
        }
#pragma unroll
        for (int k = 0; k < NR_SLOTS; ++k)
            if (k < cnt) { ox += e[k][0]; oy += e[k][1]; oz += e[k][2]; }
        int last = keys[NR_SLOTS - 1];
        for (int q = NR_SLOTS; q < cnt; ++q) {
            last = nr_next_key_mem(adj, lo, cnt, last);
            const int c = last >= 2 * F ? 2 : (last >= F ? 1 : 0);
            const float* s = fa + 9ll * (last - c * F) + 3 * c;
            ox += s[0]; oy += s[1]; oz += s[2];
        }
    }
    const long long o = ((long long)blockIdx.y * V + vi) * 3;
    g_v[o] = ox; g_v[o + 1] = oy; g_v[o + 2] = oz;
    A3D_STAMP(1, 5);
}

// ---- experiment (round 6; experiment builds only, A3D_EXP=45): the face kernel keeps ONE vector per (image, face) -- the
// adjoint of the face's un-normalised normal, g0 + g1 + g2 as nr_bwd_entry sums it (12 B instead of 36) -- and the vertex kernel rebuilds
// its corner's term from the three positions of the face (the gather form's arithmetic, so the same bits).  Scratch traffic falls from
// 72 to 24 B per (image, face); the vertex kernel grows a round trip (key -> index row -> positions).
__global__ __launch_bounds__(256) void nr_face12_bwd_kernel(const float* __restrict__ g_nrm, int g_stride, const float* __restrict__ acc,
                                                            const int* __restrict__ tri, int V, int F, float* __restrict__ fadj) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const long long vb = (long long)blockIdx.y * V;
    const float* ap = acc + vb * 3;
    const float* np = g_nrm + vb * g_stride;
    const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    const float *a0 = ap + 3ll * i0, *a1 = ap + 3ll * i1, *a2 = ap + 3ll * i2;
    const float *n0 = np + (long long)g_stride * i0, *n1 = np + (long long)g_stride * i1, *n2 = np + (long long)g_stride * i2;
    const float x0 = a0[0], y0 = a0[1], z0 = a0[2], x1 = a1[0], y1 = a1[1], z1 = a1[2], x2 = a2[0], y2 = a2[1], z2 = a2[2];
    const float gx0 = n0[0], gy0 = n0[1], gz0 = n0[2], gx1 = n1[0], gy1 = n1[1], gz1 = n1[2], gx2 = n2[0], gy2 = n2[1], gz2 = n2[2];
    float g0[3], g1[3], g2[3];
    nr_vert_adjoint(x0, y0, z0, gx0, gy0, gz0, g0);
    nr_vert_adjoint(x1, y1, z1, gx1, gy1, gz1, g1);
    nr_vert_adjoint(x2, y2, z2, gx2, gy2, gz2, g2);
    float* o = fadj + ((long long)blockIdx.y * F + f) * 3;
    o[0] = g0[0] + g1[0] + g2[0]; o[1] = g0[1] + g1[1] + g2[1]; o[2] = g0[2] + g1[2] + g2[2];
}

__global__ __launch_bounds__(256) void nr_sum12_bwd_kernel(const float* __restrict__ fadj, const float* __restrict__ v, const int* __restrict__ tri,
                                                           const int* __restrict__ off, const int* __restrict__ adj, int V, int F,
                                                           float* __restrict__ g_v, int stride) {
    const int vi = blockIdx.x * blockDim.x + threadIdx.x;
    if (vi >= V) return;
    const float* fa = fadj + (long long)blockIdx.y * F * 3;
    const float* vp = v + (long long)blockIdx.y * V * 3;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    int lo, cnt;
    vf_list(off, stride, vi, lo, cnt);
    const float zero[3] = {0.f, 0.f, 0.f};
    if (cnt > 0) {
        int keys[NR_SLOTS];
        nr_first_keys(adj, lo, cnt, keys);
        NrFace t[NR_SLOTS];
        float g[NR_SLOTS][3];
#pragma unroll
        for (int k = 0; k < NR_SLOTS; ++k) {
            const int key = k < cnt ? keys[k] : keys[0];
            t[k] = nr_decode(key, F, tri);
            const float* s = fa + 3ll * (key - t[k].c * F);
            g[k][0] = s[0]; g[k][1] = s[1]; g[k][2] = s[2];
        }
#pragma unroll
        for (int h = 0; h < NR_SLOTS; h += 4) {
            float p[4][9];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i0 = t[h + k].i0, i1 = t[h + k].i1, i2 = t[h + k].i2;
#pragma unroll
                for (int q = 0; q < 3; ++q) { p[k][q] = vp[3ll * i0 + q]; p[k][3 + q] = vp[3ll * i1 + q]; p[k][6 + q] = vp[3ll * i2 + q]; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)  // (g0 + 0 + 0 inside nr_bwd_entry: the stored sum, exactly)
                if (h + k < cnt) nr_bwd_entry(t[h + k].c, g[h + k], zero, zero, p[k], p[k] + 3, p[k] + 6, ox, oy, oz);
        }
        int last = keys[NR_SLOTS - 1];
        for (int e = NR_SLOTS; e < cnt; ++e) {
            last = nr_next_key_mem(adj, lo, cnt, last);
            const NrFace tt = nr_decode(last, F, tri);
            const float* s = fa + 3ll * (last - tt.c * F);
            const float gg[3] = {s[0], s[1], s[2]};
            float p[9];
#pragma unroll
            for (int q = 0; q < 3; ++q) { p[q] = vp[3ll * tt.i0 + q]; p[3 + q] = vp[3ll * tt.i1 + q]; p[6 + q] = vp[3ll * tt.i2 + q]; }
            nr_bwd_entry(tt.c, gg, zero, zero, p, p + 3, p + 6, ox, oy, oz);
        }
    }
    const long long o = ((long long)blockIdx.y * V + vi) * 3;
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
                               float* nrm, int lists_stride, a3d_stream_t stream) {
    A3D_CHECK_ARG(v && off && acc && nrm && B > 0 && V > 0 && F >= 0 && lists_stride >= 0);
    A3D_CHECK_ARG(F == 0 || (tri && adj));
    hipLaunchKernelGGL(nr_fwd_kernel, dim3(a3d_div_up(V, 256), B), dim3(256), 0, (hipStream_t)stream, v, tri, off, adj, V, F, acc, nrm, B,
                       (const float*)nullptr, (float*)nullptr, (float*)nullptr, lists_stride);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_normals_fwd_pair(const float* v_a, int B_a, const float* v_b, int B_b, const int32_t* tri, const int32_t* off,
                                    const int32_t* adj, int V, int F, float* acc_a, float* nrm_a, float* acc_b, float* nrm_b,
                                    int lists_stride, a3d_stream_t stream) {
    A3D_CHECK_ARG(v_a && v_b && off && acc_a && nrm_a && acc_b && nrm_b && B_a > 0 && B_b > 0 && B_a + B_b <= 65535 && V > 0 && F >= 0 && lists_stride >= 0);
    A3D_CHECK_ARG(F == 0 || (tri && adj));
    hipLaunchKernelGGL(nr_fwd_kernel, dim3(a3d_div_up(V, 256), B_a + B_b), dim3(256), 0, (hipStream_t)stream, v_a, tri, off, adj, V, F, acc_a,
                       nrm_a, B_a, v_b, acc_b, nrm_b, lists_stride);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_normals_bwd(const float* g_nrm, int g_nrm_stride, const float* acc, const float* v, const int32_t* tri, const int32_t* off,
                               const int32_t* adj, int B, int V, int F, float* g_acc_scratch, float* g_v, int lists_stride,
                               float* face_scratch_or_null, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_nrm && g_nrm_stride >= 3 && acc && v && off && g_v && B > 0 && V > 0 && F >= 0 && lists_stride >= 0 && B <= 65535);
    A3D_CHECK_ARG(g_acc_scratch || face_scratch_or_null);
    A3D_CHECK_ARG(F == 0 || (tri && adj));
    hipStream_t s = (hipStream_t)stream;
    if (face_scratch_or_null && F > 0 && a3d_exp() == 45) {  // (experiment builds only) one 12-byte vector per face in the same scratch
        hipLaunchKernelGGL(nr_face12_bwd_kernel, dim3(a3d_div_up(F, 256), B), dim3(256), 0, s, g_nrm, g_nrm_stride, acc, tri, V, F, face_scratch_or_null);
        A3D_LAUNCH_CHECK();
        hipLaunchKernelGGL(nr_sum12_bwd_kernel, dim3(a3d_div_up(V, 256), B), dim3(256), 0, s, (const float*)face_scratch_or_null, v, tri, off, adj, V, F,
                           g_v, lists_stride);
        A3D_LAUNCH_CHECK();
        return A3D_OK;
    }
    if (face_scratch_or_null && F > 0) {  // faces first: every face once, then a sum per vertex (same bits as the gather form below)
        hipLaunchKernelGGL(nr_face_bwd_kernel, dim3(a3d_div_up(F, 256), B), dim3(256), 0, s, g_nrm, g_nrm_stride, acc, v, tri, V, F, face_scratch_or_null);
        A3D_LAUNCH_CHECK();
        hipLaunchKernelGGL(nr_sum_bwd_kernel, dim3(a3d_div_up(V, 256), B), dim3(256), 0, s, (const float*)face_scratch_or_null, off, adj, V, F, g_v,
                           lists_stride);
        A3D_LAUNCH_CHECK();
        return A3D_OK;
    }
    A3D_CHECK_ARG(g_acc_scratch);
    const long long n = (long long)B * V;
    // (the one-launch form measured SLOWER inside the step: 24.8 against 17.8 us for the call -- its 24 normalisation adjoints per thread
    // with correctly rounded sqrt and division are ~1200 more instructions in a kernel that was latency bound but not idle; it stays
    // behind the knob A3D_EXP=41)
    if (a3d_exp() != 41) {
        hipLaunchKernelGGL(nr_vert_bwd_kernel, dim3(a3d_div_up(n, 256)), dim3(256), 0, s, g_nrm, g_nrm_stride, acc, n, g_acc_scratch);
        A3D_LAUNCH_CHECK();
        hipLaunchKernelGGL(nr_bwd_kernel<false>, dim3(a3d_div_up(V, 256), B), dim3(256), 0, s, g_acc_scratch, v, tri, off, adj, V, F, g_v, lists_stride,
                           (const float*)nullptr, 0, (const float*)nullptr);
    } else {
        hipLaunchKernelGGL(nr_bwd_kernel<true>, dim3(a3d_div_up(V, 256), B), dim3(256), 0, s, (const float*)nullptr, v, tri, off, adj, V, F, g_v,
                           lists_stride, g_nrm, g_nrm_stride, acc);
    }
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

A3D_PROFILE_TU(normals)
