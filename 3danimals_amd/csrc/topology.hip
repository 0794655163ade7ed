// Mesh topology of one triangle list in ONE entry point on gfx950: the vertex -> (corner, face) CSR lists the normals and the
// G-buffer backward gather over (a3d_normals_adjacency) and the edge -> opposite-vertex table of the silhouette antialiasing
// (a3d_aa_topology), built together once per DMTet call (the rest mesh, the posed meshes and every image of the batch share it):
//   init            : cursor[V] = 0, hash keys = empty, values = none                                   (one launch, not 3 memsets)
//   count + insert  : thread = corner 3f+i: atomicAdd(cursor[tri]) and hash insert of the edge opposite to it
//   scan            : single work-group exclusive scan -> off[V+1]
//   fill + lookup   : thread = corner: adj[off[v] + cursor[v]++] = c*F + f and opp[3f+i] from the finished hash
//   sort            : per-vertex lists into corner-major order (the reference's scatter_add_ order, mesh.py:291-293); the same launch
//                     re-arms cursor and hash, so a caller that keeps the scratch skips the init launch of the next call
// 4-5 launches instead of 9 (5 + 4) on a latency-bound stretch: ~24 KB of indices in, 12F + 4V + 12F bytes out.
// Replaces what /root/reference/model/render/mesh.py:276-304 (index.repeat / scatter_add_) and nvdiffrast's antialias topology
// hash (constructed lazily inside dr.antialias, render.py:264-267) do per call.
#include "a3d_common.h"
#include "topo_common.h"

namespace {

__global__ __launch_bounds__(256) void tp_init_kernel(int* __restrict__ cursor, int V, unsigned long long* __restrict__ keys, int* __restrict__ vals,
                                                      unsigned n) {
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        keys[i] = AA_EMPTY_KEY;
#pragma unroll
        for (int k = 0; k < AA_VALS; ++k) vals[AA_VALS * i + k] = AA_NONE;
    }
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)V; i += stride) cursor[i] = 0;
}

__global__ __launch_bounds__(256) void tp_count_insert_kernel(const int* __restrict__ tri, int F, int V, int* __restrict__ cnt, unsigned mask,
                                                              unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * F) return;
    const int v = tri[idx];
    if ((unsigned)v < (unsigned)V) atomicAdd(cnt + v, 1);
    aa_insert_edge(tri, idx, mask, keys, vals);
}

__global__ __launch_bounds__(256) void tp_fill_lookup_kernel(const int* __restrict__ tri, int F, int V, const int* __restrict__ off,
                                                             int* __restrict__ cursor, int* __restrict__ adj, unsigned mask,
                                                             const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                                             int* __restrict__ opp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * F) return;
    const int f = idx / 3, c = idx - 3 * f, v = tri[idx];
    if ((unsigned)v < (unsigned)V) adj[off[v] + atomicAdd(cursor + v, 1)] = c * F + f;
    opp[idx] = aa_lookup_edge(tri, idx, mask, keys, vals);
}

// the sort, and afterwards the scratch left the way the next call wants to find it: fill cursors zero, hash slots empty (neither is
// read by this launch), which saves the next call its init launch (scratch_is_clean)
__global__ __launch_bounds__(256) void tp_sort_rearm_kernel(const int* __restrict__ off, int V, int* __restrict__ adj, int* __restrict__ cursor,
                                                            unsigned long long* __restrict__ keys, int* __restrict__ vals, unsigned n) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (t < (unsigned)V) {
        nr_adj_sort_vertex(off, (int)t, adj);
        cursor[t] = 0;
    }
    for (unsigned i = t; i < n; i += stride) {
        keys[i] = AA_EMPTY_KEY;
#pragma unroll
        for (int k = 0; k < AA_VALS; ++k) vals[AA_VALS * i + k] = AA_NONE;
    }
}

// ---- second half of the fused DMTet path: the emit launch (dmtet.hip) already wrote the int32 triangle list and counted the valences.
// ONE launch (the stand-alone entry point above needs four, plus an init on first use) finishes the vertex -> (corner, face) lists:
//   every work-group scans the V valence counts into LDS itself (V+1 ints; the first one also writes off[] out) -- no scan launch;
//   thread = corner: slot in its vertex's list from an atomicAdd on the HIGH half of the count word (the low half stays readable for
//   work-groups that scan later), entry written;
//   the lists stay UNSORTED: the normals kernels take a vertex's whole list into registers and order the keys there (normals.hip) --
//   no sort launch;
//   the count array of the NEXT extraction (the two alternate) is zeroed by grid-stride stores -- no init launch.
// There is no edge hash and no opposite-vertex table on this path: the silhouette analysis looks the few opposite vertices it needs up
// in these lists (topo_common.h: aa_opposite_from_lists, two batched round trips per lookup).
// (Measured on the way: hash inserts inside the emit launch -- three serial CAS chains per face thread: emit 17 -> 38 us; hash insert
//  in this launch + a sort/lookup launch: 12 + 9 us of kernel time, as much as the four launches they replace.)
__global__ __launch_bounds__(256) void tp_finalize_kernel(const int* __restrict__ tri, int F, int V, int* cnt, int* __restrict__ off,
                                                          int* __restrict__ adj, int* __restrict__ cnt_next, int v_next) {
    extern __shared__ int s_off[];  // [V + 1]
    __shared__ int s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = blockIdx.x * blockDim.x + tid;
    const int my_v = idx < 3 * F ? tri[idx] : -1;  // (issued first: overlaps with the scan)
    {   // exclusive scan of (cnt & 0xFFFF) over the vertices: a contiguous run per thread, eight unconditional loads in flight
        const int per = (V + 255) / 256;
        const int lo = min(tid * per, V), hi = min(lo + per, V);
        int mine = 0;
        for (int i = lo; i < hi; i += 8) {
            int c[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = cnt[min(i + k, V - 1)] & 0xFFFF;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (i + k < hi) {
                    s_off[i + k] = mine;  // run-local exclusive prefix; the run's base is added below
                    mine += c[k];
                }
            }
        }
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int base = incl - mine;
        for (int w = 0; w < wave; ++w) base += s_wave[w];
        for (int i = lo; i < hi; ++i) s_off[i] += base;
        if (tid == 255) s_off[V] = base + mine;
        __syncthreads();
        if (blockIdx.x == 0)
            for (int i = tid; i <= V; i += 256) off[i] = s_off[i];
    }
    if ((unsigned)my_v < (unsigned)V) {
        const int f = idx / 3, c = idx - 3 * f;
        const int slot = atomicAdd(cnt + my_v, 0x10000) >> 16;
        adj[s_off[my_v] + slot] = c * F + f;
    }
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + tid; i < (unsigned)v_next; i += stride) cnt_next[i] = 0;
}

// ---- the same for LARGE meshes (V above TP_LDS_SCAN_MAX): every work-group scanning all V counts is O(V * F / 256) loads (R = 128 grid,
// V = 2.4e4, 558 work-groups: 49 us); there the counts are scanned once by the single-work-group scan (the counts it resets become the
// fill cursor) and a plain fill launch follows -- two launches of ~8 us.
__global__ __launch_bounds__(256) void tp_fill_kernel(const int* __restrict__ tri, int F, int V, const int* __restrict__ off, int* __restrict__ cursor,
                                                      int* __restrict__ adj, int* __restrict__ cnt_next, int v_next) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 3 * F) {
        const int f = idx / 3, c = idx - 3 * f, v = tri[idx];
        if ((unsigned)v < (unsigned)V) adj[off[v] + atomicAdd(cursor + v, 1)] = c * F + f;
    }
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)v_next; i += stride) cnt_next[i] = 0;
}

}  // namespace

#define TP_LDS_SCAN_MAX 12288

extern "C" int a3d_mesh_topology_finalize_max_vertices(void) { return 0x3fffffff; }  // (no limit: large meshes take the two-launch form)

extern "C" int a3d_mesh_topology_finalize(const int32_t* tri, int V, int F, int32_t* count, int32_t* off, int32_t* adj, int32_t* count_next,
                                          int v_next, a3d_stream_t stream) {
    A3D_CHECK_ARG(tri && count && off && adj && count_next && V > 0 && F > 0 && v_next >= 0);
    A3D_CHECK_ARG((long long)3 * F < 0x7fffffffll);
    hipStream_t s = (hipStream_t)stream;
    if (V > TP_LDS_SCAN_MAX) {
        hipLaunchKernelGGL(nr_adj_scan_kernel, dim3(1), dim3(NR_SCAN_THREADS), 0, s, count, V, off);  // (count -> 0: it is the fill cursor now)
        A3D_LAUNCH_CHECK();
        hipLaunchKernelGGL(tp_fill_kernel, dim3(a3d_div_up(3ll * F, 256)), dim3(256), 0, s, tri, F, V, off, count, adj, count_next, v_next);
        A3D_LAUNCH_CHECK();
        return A3D_OK;
    }
    const size_t lds = sizeof(int) * ((size_t)V + 1);
    if (lds > 48 * 1024) A3D_HIP(hipFuncSetAttribute((const void*)tp_finalize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(tp_finalize_kernel, dim3(a3d_div_up(3ll * F, 256)), dim3(256), lds, s, tri, F, V, count, off, adj, count_next, v_next);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_mesh_topology(const int32_t* tri, int V, int F, int32_t* off, int32_t* adj, int32_t* cursor, void* hash, int32_t* opp,
                                 int scratch_is_clean, a3d_stream_t stream) {
    A3D_CHECK_ARG(off && cursor && V > 0 && F >= 0 && (long long)3 * F < 0x7fffffffll);
    A3D_CHECK_ARG(F == 0 || (tri && adj && hash && opp));
    hipStream_t s = (hipStream_t)stream;
    const unsigned n = F > 0 ? aa_slots(F) : 0;
    unsigned long long* keys = (unsigned long long*)hash;
    int* vals = (int*)(keys + n);
    const long long work = (long long)n > V ? (long long)n : V;
    int blocks = a3d_div_up(work, 256);
    if (blocks > 1024) blocks = 1024;
    if (!scratch_is_clean) {  // (clean: cursor[V] and the hash of a3d_aa_hash_bytes(F) bytes as a previous call of the same F left them)
        hipLaunchKernelGGL(tp_init_kernel, dim3(blocks), dim3(256), 0, s, cursor, V, keys, vals, n);
        A3D_LAUNCH_CHECK();
    }
    if (F > 0) {
        hipLaunchKernelGGL(tp_count_insert_kernel, dim3(a3d_div_up(3ll * F, 256)), dim3(256), 0, s, tri, F, V, cursor, n - 1, keys, vals);
        A3D_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(nr_adj_scan_kernel, dim3(1), dim3(NR_SCAN_THREADS), 0, s, cursor, V, off);
    A3D_LAUNCH_CHECK();
    if (F > 0) {
        hipLaunchKernelGGL(tp_fill_lookup_kernel, dim3(a3d_div_up(3ll * F, 256)), dim3(256), 0, s, tri, F, V, off, cursor, adj, n - 1, keys, vals, opp);
        A3D_LAUNCH_CHECK();
        // (enough work-groups that clearing the n hash slots is one or two rounds of stores, not ten)
        hipLaunchKernelGGL(tp_sort_rearm_kernel, dim3(a3d_div_up((long long)V > (long long)n / 2 ? (long long)V : (long long)n / 2, 256)), dim3(256), 0, s, off, V,
                           adj, cursor, keys, vals, n);
        A3D_LAUNCH_CHECK();
    }
    return A3D_OK;
}
