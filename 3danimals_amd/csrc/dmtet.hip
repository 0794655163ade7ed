// DMTet marching tetrahedra on gfx950 -- sort-free formulation.
//
// The reference (model/geometry/dmtet.py:104-155) gathers the six edges of every surface-crossing tet,
// sorts them globally inside torch.unique(dim=0) and numbers the surface vertices by rank.  Because the
// grid's unique edge list is static and already lexicographically sorted (dmtet.py:283-288), that rank is
// an exclusive prefix sum of the "sign crossing" flag over the static list.  So:
//   count : wave-ballot popcounts of the crossing flag per 1024-edge block and of the 1-/2-triangle case
//           per 1024-tet block, then one work-group scan per block-sum array  (-> V, n1, n2)
//   emit  : edges re-evaluate the flag, ballot/mbcnt gives the in-wave rank, LDS the cross-wave offset;
//           tets look their 3-4 surface vertices up through tet2edge -> edge2vert and write int64 faces.
// Integer/byte work, HBM-bound: 8 B/edge + 16 B/tet (count) and 8 B/edge + 40 B/tet (emit) of streaming
// reads plus L2-resident sdf gathers.  This TU is compiled with -ffp-contract=off: the vertex placement
// must round exactly like the reference's separate torch kernels (mul, mul, add).
#include "a3d_common.h"

#define DM_THREADS 256
#define DM_SLABS 4
#define DM_BLOCK_ITEMS (DM_THREADS * DM_SLABS)

// reference dmtet.py:26-45
__constant__ signed char c_tri_table[16][6] = {
    {-1, -1, -1, -1, -1, -1}, {1, 0, 2, -1, -1, -1}, {4, 0, 3, -1, -1, -1}, {1, 4, 2, 1, 3, 4},
    {3, 1, 5, -1, -1, -1},    {2, 3, 0, 2, 5, 3},    {1, 4, 0, 1, 5, 4},    {4, 2, 5, -1, -1, -1},
    {4, 5, 2, -1, -1, -1},    {4, 1, 0, 4, 5, 1},    {3, 2, 0, 3, 5, 2},    {1, 3, 5, -1, -1, -1},
    {4, 1, 2, 4, 3, 1},       {3, 0, 4, -1, -1, -1}, {2, 0, 1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1}};
// number of triangles per case packed 2 bits each: [0,1,1,2,1,2,2,1,1,2,2,1,2,1,1,0]
#define DM_NTRI(c) ((0x16696994u >> (2 * (c))) & 3u)

static_assert(((0x16696994u >> 0) & 3) == 0 && ((0x16696994u >> 6) & 3) == 2 && ((0x16696994u >> 30) & 3) == 0, "ntri pack");

__device__ __forceinline__ int dm_tet_case(const float* __restrict__ sdf, int4 t) {
    return (sdf[t.x] > 0.f ? 1 : 0) | (sdf[t.y] > 0.f ? 2 : 0) | (sdf[t.z] > 0.f ? 4 : 0) | (sdf[t.w] > 0.f ? 8 : 0);
}

__device__ __forceinline__ bool dm_edge_cross(const float* __restrict__ sdf, int2 e) {
    return (sdf[e.x] > 0.f) != (sdf[e.y] > 0.f);  // exactly one endpoint inside (dmtet.py:118); 0 counts as outside
}

// ------------------------------------------------------------------------------------------------ count
__global__ __launch_bounds__(DM_THREADS) void dm_count_kernel(const float* __restrict__ sdf, const int2* __restrict__ edges,
                                                                const int4* __restrict__ tets, int Ne, int Nt, int nbe,
                                                                int* __restrict__ blk_e, int* __restrict__ blk_t1,
                                                                int* __restrict__ blk_t2) {
    __shared__ int s_cnt[2][DM_THREADS / A3D_WAVE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int c0 = 0, c1 = 0;
    if ((int)blockIdx.x < nbe) {
        const long long base = (long long)blockIdx.x * DM_BLOCK_ITEMS;
#pragma unroll
        for (int k = 0; k < DM_SLABS; ++k) {
            long long i = base + k * DM_THREADS + tid;
            bool f = i < Ne && dm_edge_cross(sdf, edges[i]);
            c0 += __popcll(__ballot(f));  // wave-uniform
        }
    } else {
        const long long base = (long long)(blockIdx.x - nbe) * DM_BLOCK_ITEMS;
#pragma unroll
        for (int k = 0; k < DM_SLABS; ++k) {
            long long i = base + k * DM_THREADS + tid;
            unsigned n = 0;
            if (i < Nt) n = DM_NTRI(dm_tet_case(sdf, tets[i]));
            c0 += __popcll(__ballot(n == 1u));
            c1 += __popcll(__ballot(n == 2u));
        }
    }
    if (lane == 0) { s_cnt[0][wave] = c0; s_cnt[1][wave] = c1; }
    __syncthreads();
    if (tid == 0) {
        int a = 0, b = 0;
        for (int w = 0; w < DM_THREADS / A3D_WAVE; ++w) { a += s_cnt[0][w]; b += s_cnt[1][w]; }
        if ((int)blockIdx.x < nbe) blk_e[blockIdx.x] = a;
        else { blk_t1[blockIdx.x - nbe] = a; blk_t2[blockIdx.x - nbe] = b; }
    }
}

// three work-groups, one per block-sum array: in-place exclusive scan, totals to counts[0..2]
__global__ __launch_bounds__(1024) void dm_scan_kernel(int* __restrict__ blk_e, int* __restrict__ blk_t1, int* __restrict__ blk_t2,
                                                       int nbe, int nbt, int* __restrict__ counts) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int which = blockIdx.x;
    int* arr = which == 0 ? blk_e : (which == 1 ? blk_t1 : blk_t2);
    const int n = which == 0 ? nbe : nbt;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int i = base + tid;
        int v = i < n ? arr[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_wave[w];
        int carry = s_carry;
        if (i < n) arr[i] = carry + woff + incl - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + woff + incl;
        __syncthreads();
    }
    if (tid == 0) {
        counts[which] = s_carry;
        if (which == 0) counts[3] = 0;
    }
}

// ------------------------------------------------------------------------------------------------ emit
__global__ __launch_bounds__(DM_THREADS) void dm_emit_edges_kernel(const float* __restrict__ pos, const float* __restrict__ sdf,
                                                                     const int2* __restrict__ edges, int Ne,
                                                                     const int* __restrict__ blk_e, int* __restrict__ edge2vert,
                                                                     float* __restrict__ verts, int* __restrict__ vert_edge) {
    __shared__ int s_cnt[DM_THREADS / A3D_WAVE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int running = blk_e[blockIdx.x];
    const long long base = (long long)blockIdx.x * DM_BLOCK_ITEMS;
    for (int k = 0; k < DM_SLABS; ++k) {
        long long i = base + k * DM_THREADS + tid;
        int2 e = make_int2(0, 0);
        float sa = 0.f, sb = 0.f;
        bool f = false;
        if (i < Ne) {
            e = edges[i];
            sa = sdf[e.x];
            sb = sdf[e.y];
            f = (sa > 0.f) != (sb > 0.f);
        }
        unsigned long long m = __ballot(f);
        if (lane == 0) s_cnt[wave] = __popcll(m);
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < DM_THREADS / A3D_WAVE; ++w) {
            int c = s_cnt[w];
            if (w < wave) woff += c;
            total += c;
        }
        if (i < Ne) {
            int vid = -1;
            if (f) {
                vid = running + woff + a3d_wave_prefix(m);
                // reference dmtet.py:124-131: w = flip([s_a, -s_b]) / (s_a + (-s_b)); v = p_a*w_a + p_b*w_b
                float nsb = -sb;
                float den = sa + nsb;
                float wa = nsb / den, wb = sa / den;
                const float* pa = pos + 3ll * e.x;
                const float* pb = pos + 3ll * e.y;
                float* o = verts + 3ll * vid;
                o[0] = pa[0] * wa + pb[0] * wb;
                o[1] = pa[1] * wa + pb[1] * wb;
                o[2] = pa[2] * wa + pb[2] * wb;
                vert_edge[vid] = (int)i;
            }
            edge2vert[i] = vid;
        }
        running += total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(DM_THREADS) void dm_emit_tets_kernel(const float* __restrict__ sdf, const int4* __restrict__ tets,
                                                                    const int* __restrict__ tet2edge, int Nt,
                                                                    const int* __restrict__ blk_t1, const int* __restrict__ blk_t2,
                                                                    const int* __restrict__ edge2vert, int n1,
                                                                    long long* __restrict__ faces, long long* __restrict__ uv_idx) {
    __shared__ int s_cnt[2][DM_THREADS / A3D_WAVE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int run1 = blk_t1[blockIdx.x], run2 = blk_t2[blockIdx.x];
    const long long base = (long long)blockIdx.x * DM_BLOCK_ITEMS;
    for (int k = 0; k < DM_SLABS; ++k) {
        long long t = base + k * DM_THREADS + tid;
        int cs = 0;
        unsigned n = 0;
        if (t < Nt) {
            cs = dm_tet_case(sdf, tets[t]);
            n = DM_NTRI(cs);
        }
        unsigned long long m1 = __ballot(n == 1u), m2 = __ballot(n == 2u);
        if (lane == 0) { s_cnt[0][wave] = __popcll(m1); s_cnt[1][wave] = __popcll(m2); }
        __syncthreads();
        int w1 = 0, w2 = 0, t1 = 0, t2 = 0;
#pragma unroll
        for (int w = 0; w < DM_THREADS / A3D_WAVE; ++w) {
            int a = s_cnt[0][w], b = s_cnt[1][w];
            if (w < wave) { w1 += a; w2 += b; }
            t1 += a; t2 += b;
        }
        if (n != 0u) {
            const int* te = tet2edge + 6ll * t;
            int ev[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) ev[j] = te[j];
            long long slot = (n == 1u) ? (long long)(run1 + w1 + a3d_wave_prefix(m1))
                                       : (long long)n1 + 2ll * (run2 + w2 + a3d_wave_prefix(m2));
            const signed char* row = c_tri_table[cs];
            for (unsigned q = 0; q < n; ++q) {
                long long* fo = faces + 3 * (slot + q);
                long long* uo = uv_idx + 3 * (slot + q);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    int slot_e = row[3 * q + j];
                    int eid = ev[0];
#pragma unroll
                    for (int s = 1; s < 6; ++s) eid = (slot_e == s) ? ev[s] : eid;  // register select, no scratch
                    fo[j] = (long long)edge2vert[eid];
                }
                // reference dmtet.py:91-96 with face_gidx = 2t + q
                uo[0] = 4ll * t;
                uo[1] = 4ll * t + q + 1;
                uo[2] = 4ll * t + q + 2;
            }
        }
        run1 += t1;
        run2 += t2;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ backward
__global__ __launch_bounds__(256) void dm_bwd_kernel(const float* __restrict__ g_verts, const float* __restrict__ pos,
                                                     const float* __restrict__ sdf, const int2* __restrict__ edges,
                                                     const int* __restrict__ vert_edge, int V, float* __restrict__ g_pos,
                                                     float* __restrict__ g_sdf) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    int2 e = edges[vert_edge[v]];
    float sa = sdf[e.x], sb = sdf[e.y];
    float den = sa - sb, inv = 1.f / den, inv2 = inv * inv;
    float gx = g_verts[3ll * v], gy = g_verts[3ll * v + 1], gz = g_verts[3ll * v + 2];
    const float* pa = pos + 3ll * e.x;
    const float* pb = pos + 3ll * e.y;
    float g_wa = gx * pa[0] + gy * pa[1] + gz * pa[2];
    float g_wb = gx * pb[0] + gy * pb[1] + gz * pb[2];
    // w_a = -s_b/den, w_b = s_a/den
    atomicAdd(g_sdf + e.x, (g_wa - g_wb) * sb * inv2);
    atomicAdd(g_sdf + e.y, (g_wb - g_wa) * sa * inv2);
    if (g_pos) {
        float wa = -sb * inv, wb = sa * inv;
        atomicAdd(g_pos + 3ll * e.x + 0, gx * wa); atomicAdd(g_pos + 3ll * e.x + 1, gy * wa); atomicAdd(g_pos + 3ll * e.x + 2, gz * wa);
        atomicAdd(g_pos + 3ll * e.y + 0, gx * wb); atomicAdd(g_pos + 3ll * e.y + 1, gy * wb); atomicAdd(g_pos + 3ll * e.y + 2, gz * wb);
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" size_t a3d_dmtet_scratch_bytes(int Ne, int Nt) {
    return sizeof(int) * ((size_t)a3d_div_up(Ne, DM_BLOCK_ITEMS) + 2 * (size_t)a3d_div_up(Nt, DM_BLOCK_ITEMS) + 4);
}

static void dm_split_scratch(void* scratch, int Ne, int Nt, int** e, int** t1, int** t2, int* nbe, int* nbt) {
    *nbe = a3d_div_up(Ne, DM_BLOCK_ITEMS);
    *nbt = a3d_div_up(Nt, DM_BLOCK_ITEMS);
    *e = (int*)scratch;
    *t1 = *e + *nbe;
    *t2 = *t1 + *nbt;
}

extern "C" int a3d_dmtet_count(const float* sdf, const int32_t* edges, const int32_t* tets, int Ne, int Nt, void* block_scan,
                               int32_t* counts, a3d_stream_t stream) {
    A3D_CHECK_ARG(sdf && edges && tets && block_scan && counts);
    A3D_CHECK_ARG(Ne > 0 && Nt > 0);
    int *be, *b1, *b2, nbe, nbt;
    dm_split_scratch(block_scan, Ne, Nt, &be, &b1, &b2, &nbe, &nbt);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(dm_count_kernel, dim3(nbe + nbt), dim3(DM_THREADS), 0, s, sdf, (const int2*)edges, (const int4*)tets, Ne, Nt,
                       nbe, be, b1, b2);
    A3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(dm_scan_kernel, dim3(3), dim3(1024), 0, s, be, b1, b2, nbe, nbt, counts);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_dmtet_emit(const float* pos, const float* sdf, const int32_t* edges, const int32_t* tets, const int32_t* tet2edge,
                              int Ne, int Nt, const void* block_scan, int V, int n1, int n2, int32_t* edge2vert, float* verts,
                              int32_t* vert_edge, int64_t* faces, int64_t* uv_idx, a3d_stream_t stream) {
    A3D_CHECK_ARG(pos && sdf && edges && tets && tet2edge && block_scan && edge2vert);
    A3D_CHECK_ARG(Ne > 0 && Nt > 0 && V >= 0 && n1 >= 0 && n2 >= 0);
    A3D_CHECK_ARG(V == 0 || (verts && vert_edge));
    A3D_CHECK_ARG((n1 + n2) == 0 || (faces && uv_idx));
    int *be, *b1, *b2, nbe, nbt;
    dm_split_scratch((void*)block_scan, Ne, Nt, &be, &b1, &b2, &nbe, &nbt);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(dm_emit_edges_kernel, dim3(nbe), dim3(DM_THREADS), 0, s, pos, sdf, (const int2*)edges, Ne, be, edge2vert,
                       verts, vert_edge);
    A3D_LAUNCH_CHECK();
    if (n1 + n2 > 0) {
        hipLaunchKernelGGL(dm_emit_tets_kernel, dim3(nbt), dim3(DM_THREADS), 0, s, sdf, (const int4*)tets, tet2edge, Nt, b1, b2,
                           edge2vert, n1, (long long*)faces, (long long*)uv_idx);
        A3D_LAUNCH_CHECK();
    }
    return A3D_OK;
}

extern "C" int a3d_dmtet_bwd(const float* g_verts, const float* pos, const float* sdf, const int32_t* edges, const int32_t* vert_edge,
                             int V, int Nv, float* g_pos_or_null, float* g_sdf, a3d_stream_t stream) {
    A3D_CHECK_ARG(pos && sdf && edges && g_sdf && Nv > 0 && V >= 0);
    hipStream_t s = (hipStream_t)stream;
    A3D_HIP(hipMemsetAsync(g_sdf, 0, sizeof(float) * (size_t)Nv, s));
    if (g_pos_or_null) A3D_HIP(hipMemsetAsync(g_pos_or_null, 0, sizeof(float) * 3 * (size_t)Nv, s));
    if (V > 0) {
        A3D_CHECK_ARG(g_verts && vert_edge);
        hipLaunchKernelGGL(dm_bwd_kernel, dim3(a3d_div_up(V, 256)), dim3(256), 0, s, g_verts, pos, sdf, (const int2*)edges, vert_edge, V,
                           g_pos_or_null, g_sdf);
        A3D_LAUNCH_CHECK();
    }
    return A3D_OK;
}
