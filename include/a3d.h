/* a3d.h -- flat C ABI of liba3d_hip.so: the MI355X (gfx950) reconstruct-and-render hot path of 3DAnimals.
 *
 * Every entry point replaces an operator the reference reaches through Python (there is no FFI in the
 * reference for this path -- it is pure PyTorch plus the third-party nvdiffrast extension); the
 * reference-side interface each one stands in for is cited as /root/reference file:line.  The Python
 * binding a maintainer adds is a ctypes stub, see INTEGRATION.md and 3danimals_amd/_lib.py.
 *
 * Conventions
 *  - all pointers are DEVICE pointers owned by the caller (PyTorch); row-major, densely packed;
 *    float = IEEE fp32, indices int32 on the device side (the Python layer keeps the reference's
 *    int64 index tensors and converts once per topology, as the reference does at the dr.* boundary,
 *    render.py:182,292); DMTet face buffers are emitted as int64 because that is what callers index with.
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is
 *    enqueued on it; no entry point synchronises, allocates or frees device memory.
 *  - return 0 on success, a negative A3D_E* code otherwise; a3d_last_error() gives a thread-local message.
 *  - "zeroed by callee" = the function enqueues the hipMemsetAsync itself.
 *  - Bx arguments named *_batch are 1 (shared, broadcast over the batch) or B.
 */
#ifndef A3D_H
#define A3D_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A3D_OK 0
#define A3D_EINVAL (-1)  /* bad argument */
#define A3D_EHIP (-2)    /* a HIP runtime call failed */

typedef void* a3d_stream_t;

int a3d_version(void); /* 404 = this header */
const char* a3d_last_error(void);

/* Box fingerprint for the benchmark line (no reference counterpart: the reference's meter, /root/reference/model/utils/meters.py:119, reports
 * images/s only): a streaming fill of `n_floats` floats (one non-temporal 16-byte store per thread, 256 threads per work-group) and a
 * grid-stride streaming read of the same range; bench.py times both with HIP events on `stream` and prints the GB/s beside every
 * kernel's figure.  `dst` / `src` 16-byte aligned, n_floats a multiple of 4; `sink` = one float nobody reads. */
int a3d_bw_probe_fill(float* dst, int64_t n_floats, a3d_stream_t stream);
int a3d_bw_probe_read(const float* src, int64_t n_floats, float* sink, a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DMTet marching tetrahedra -- replaces DMTet.__call__ + map_uv index part,
 * /root/reference/model/geometry/dmtet.py:104-155, :86-96.
 * Static per grid (caller precomputes once): edges[Ne,2] = lexicographically sorted unique (min,max)
 * tet edges (== dmtet.py:283-288), tet2edge[Nt,6] = row in `edges` of each tet edge slot (dmtet.py:46).
 * Two phases with one 16-byte read-back in between (output sizes are data dependent; the reference
 * synchronises at the same place, dmtet.py:110):
 *   a3d_dmtet_count : counts[0]=V crossing edges, counts[1]=n1 one-triangle tets, counts[2]=n2 two-triangle
 *                     tets (F = n1 + 2 n2), counts[3] = flagged grid vertices (vertex_scratch; else 0), counts[4], counts[5] = the
 *                     numbers of non-empty edge / tet blocks listed for the emit launch (-1, -1 without word groups);
 *                     scratch = a3d_dmtet_scratch_bytes(Ne,Nt) bytes, 8-byte aligned; it carries the block scans, the crossing /
 *                     case bit planes and those lists from count to emit.
 *   a3d_dmtet_emit  : verts[V,3] (vertex v = v-th crossing edge in `edges` order, placed at the SDF zero
 *                     crossing with the reference's operation order), vert_edge[V] (edge row, for backward),
 *                     faces[F,3] int64 (1-triangle tets first, then 2-triangle tets, dmtet.py:140-143),
 *                     uv_idx[F,3] int64 (dmtet.py:91-96).
 *   a3d_dmtet_bwd   : g_sdf[Nv] (zeroed by callee) and optionally g_pos[Nv,3] (zeroed by callee) from g_verts.
 */
size_t a3d_dmtet_scratch_bytes(int Ne, int Nt);
int a3d_dmtet_count(const float* sdf, const int32_t* edges, const int32_t* tets, int Ne, int Nt, void* scratch,
                    int32_t* counts /*[6]*/, void* vertex_scratch_or_null, int vertex_scratch_is_clean, int Nv,
                    const uint32_t* edge_groups_or_null, const uint32_t* tet_groups_or_null, int group_slots /* 8 or 16 */,
                    int32_t* words_to_clear_or_null, int n_words_to_clear, a3d_stream_t stream);
/* words_to_clear: n 4-byte words zeroed by the last launch of the call (the valence counters a3d_dmtet_emit's topo_count wants zero). */
/* edge_groups / tet_groups (both or none; static per grid like `edges`): the culled count pass.  Row w of edge_groups
 * [ceil(Ne / a3d_dmtet_block_items()) * a3d_dmtet_block_items() / 64 rows x group_slots (8, or 16 for files whose words touch more groups)] lists the distinct values of
 * (vertex index >> a3d_dmtet_word_group_bits()) over the 64 consecutive rows edges[64 w .. 64 w + 63] (unused slots repeat one of
 * them; rows past the list and rows with more distinct values than slots hold 0xffffffff in every slot); tet_groups the same over
 * `tets`.  With them the pass takes a sign-plane pre-pass and reads a word's index rows only if its vertex groups do not all lie on one
 * side of the surface -- on a grid numbered along its rows ~90 % of the 8 B/edge + 16 B/tet stream is never read.  Same outputs bit for
 * bit.  The reference has no counterpart: it evaluates every tet on every call (dmtet.py:107-118). */
int a3d_dmtet_word_group_slots(void);
int a3d_dmtet_word_group_bits(void);
int a3d_dmtet_block_items(void);
/* The culled count pass for a grid in ANY numbering (a file from an external mesher: the reference's data/tets/{128,256}_tets.npz,
 * /root/reference/model/geometry/dmtet.py:214-226, data/tets/generate_tets.py:31-47).  The output order of DMTet is defined by the file's
 * numbering (vertex = rank of its crossing edge in the sorted `edges`, faces in `tets` order: dmtet.py:115-121,140-143), but on such a
 * grid 64 consecutive rows are anywhere in space and the word groups above cull nothing.  So the caller ranks the vertices along a
 * space-filling curve once (static per grid) and hands over the rows rewritten in ranks, each with the row it came from; the pass
 * evaluates signs and culls in THAT order and sets the crossing / case bits at the rows of the ORIGINAL order.  Same scratch contents
 * as a3d_dmtet_count leaves (no block lists: counts[4] = counts[5] = -1), same counts, same a3d_dmtet_emit afterwards, same output bits. */
typedef struct a3d_dmtet_order {
    uint32_t size;                 /* sizeof(a3d_dmtet_order) of the caller's header (fields are only ever appended) */
    int32_t group_slots;           /* slots per row of edge_groups / tet_groups: 8 or 16 */
    const int32_t* vertex_of_rank; /* [Nv]   grid vertex at rank r (a permutation) */
    const int32_t* edges_ranked;   /* [Ne,2] the rows of `edges` with both entries replaced by ranks, in any row order (best: sorted) */
    const int32_t* edge_of_row;    /* [Ne]   row of `edges` that edges_ranked[i] came from */
    const int32_t* tets_ranked;    /* [Nt,4] the rows of `tets` in ranks, corner order KEPT (the case index depends on it), any row order */
    const int32_t* tet_of_row;     /* [Nt]   row of `tets` that tets_ranked[i] came from */
    const uint32_t* edge_groups;   /* word groups (see a3d_dmtet_count) of edges_ranked, group_slots per row */
    const uint32_t* tet_groups;    /* ... of tets_ranked */
} a3d_dmtet_order;
int a3d_dmtet_count_ordered(const float* sdf, int Nv, int Ne, int Nt, const a3d_dmtet_order* order, void* scratch /*16-byte aligned*/,
                            int32_t* counts /*[6]*/, void* vertex_scratch_or_null, int vertex_scratch_is_clean,
                            int32_t* words_to_clear_or_null, int n_words_to_clear, a3d_stream_t stream);
/* (Nv = number of grid vertices, or 0 if unknown.  Grids of >= 2^20 vertices take a pre-pass that leaves one sign bit per vertex in
 * scratch; the count pass then looks signs up there -- a handful of cache lines per wave instead of one per 32 vertices -- and streams
 * its index rows with four rows per lane in flight.  Same bit planes and counts either way.) */
struct a3d_dmtet_emit_opts; /* the optional groups below, by name (defined after the prose that describes them) */
int a3d_dmtet_emit(const float* pos, const float* sdf, const int32_t* edges, const int32_t* tet2edge, int Ne, int Nt,
                   const void* scratch, int V, int n1, int n2, float* verts, int32_t* vert_edge, int64_t* faces, int64_t* uv_idx,
                   const struct a3d_dmtet_emit_opts* opts_or_null, a3d_stream_t stream);
/* The emit launch for SPARSE planes -- after a3d_dmtet_count_ordered, where the surface items of a randomly numbered grid are spread
 * evenly over the planes (one or two per 1024-row block at the "256" class) and a work-group per block is all latency: thread = 64-row
 * word of a plane instead.  Same outputs as a3d_dmtet_emit, bit for bit; `scratch` must come from a3d_dmtet_count_ordered (whose scan
 * launch also leaves the in-block prefixes of the tet planes, which this launch reads).  The optional groups of a3d_dmtet_emit travel in
 * a struct (same meaning, see there): */
typedef struct a3d_dmtet_emit_opts {
    uint32_t size;                /* sizeof(a3d_dmtet_emit_opts) of the caller's header (fields are only ever appended) */
    int32_t Nv;                   /* grid vertices (needed with vertex_scratch / g_sdf_to_clear) */
    void* vertex_scratch;         /* the flagged-vertex plane of the count call, or NULL */
    int64_t* surf_idx;            /* [n_surf] sorted list of the flagged vertices (with vertex_scratch) */
    float* g_sdf_to_clear;        /* [Nv] zeroed by the launch, or NULL */
    int32_t* tri32;               /* [F,3] int32 copy of faces, or NULL (with topo_count) */
    int32_t* topo_count;          /* [>= V] valence counters, zero on entry */
    int32_t* topo_adj;            /* [V * topo_stride] vertex -> face lists, or NULL */
    const int32_t* device_counts; /* speculative launch: `counts` of the count call, still on the device; V, n1 (= F), n_surf are capacities */
    int32_t n_surf;
    int32_t topo_stride;
    int32_t use_block_lists;      /* a3d_dmtet_emit only: 1 = cover only the non-empty blocks the culled count pass listed ... */
    int32_t n_edge_blocks_listed; /* ... = counts[4] of that a3d_dmtet_count call (speculative launch: a capacity) */
    int32_t n_tet_blocks_listed;  /* ... = counts[5] */
    int32_t surf_bucket;          /* (404) with surf_pts: the row padding of that block (> 0; a speculative n_surf must be a multiple of it) */
    float* surf_pts;              /* (404) [round_up(n_surf, surf_bucket), 3] = pos[surf_idx[i]], zero rows behind -- or NULL (see below) */
} a3d_dmtet_emit_opts;
int a3d_dmtet_emit_sparse(const float* pos, const float* sdf, const int32_t* edges, const int32_t* tet2edge, int Ne, int Nt,
                          const void* scratch, int V, int n1, int n2, float* verts, int32_t* vert_edge, int64_t* faces, int64_t* uv_idx,
                          const a3d_dmtet_emit_opts* opts_or_null, a3d_stream_t stream);
/* n_*_blocks_listed = counts[4], counts[5] of the a3d_dmtet_count call that filled `scratch`: with the culled count pass (word groups)
 * the blocks that hold a crossing edge / a surface tet are listed there and the emit launch covers those alone (~5 % of the grid's
 * blocks); -1, -1 (what the plain count pass reports) = every block.
 * device_counts != NULL: a SPECULATIVE call, made before the host has read `counts` -- the GPU does not idle across the read-back.
 * V, n1 (read as F), n_surf and the two listed counts then are CAPACITIES (of the output buffers / of the launch, e.g. the previous
 * extraction's numbers + 25 %), n2 is ignored, and the kernel takes the true numbers from device_counts (= `counts` of the count call on
 * the same stream).  If any of them exceeds its capacity the launch leaves every buffer untouched: the caller compares the counts it
 * reads back with the capacities and, in that case, calls again with exact sizes. */
/* tri32 / topo_count (both or none): the emit launch also writes the int32 copy of faces that the render kernels read and counts the
 * valences of the surface vertices (topo_count[>= V], zero on entry) -- the first step of the mesh topology, which
 * a3d_mesh_topology_finalize completes in one launch (the stand-alone a3d_mesh_topology needs four).
 * topo_adj[V * topo_stride] (with tri32 / topo_count): the emit launch completes the topology ITSELF -- face f's corner c appends the key
 * c*F + f to the list of its vertex at adj[v * topo_stride + (old count)], topo_count[v] ends as the list length (lists_stride =
 * topo_stride layout, see a3d_normals_*).  topo_stride must be >= the largest possible valence: 2 x the largest number of tets around
 * one edge of the grid (a static property; 12 on the Kuhn grids); entries past it would be dropped. */
/* Optional, for callers that evaluate the SDF network with a graph only where the surface's gradient can reach (DMTetGeometry.
 * _get_mesh_surface_backward): with vertex_scratch (a3d_dmtet_vertex_scratch_bytes(Nv) bytes, 16-byte aligned) a3d_dmtet_count also
 * flags the grid vertices at the ends of crossing edges and returns their number in counts[3] -- the same read-back as V, n1, n2 --
 * and a3d_dmtet_emit writes them as a sorted int64 list surf_idx[n_surf] (extra work-groups of the same launch) and clears the flags
 * again: a caller that hands the same vertex_scratch to the next count on the same stream passes vertex_scratch_is_clean = 1 and
 * saves its memset.  Replaces a mask + torch.nonzero and its own host synchronisation. */
size_t a3d_dmtet_vertex_scratch_bytes(int Nv);
/* (404) surf_pts / surf_bucket: the emit launch also leaves the grid POSITIONS of the listed vertices as a zero-padded block (the rows
 * the SDF network is re-evaluated on, dmtet.py:228-250 through DMTetGeometry._get_mesh_surface_backward) -- the forward one of the two
 * a3d_dmtet_gather_rows launches below is then not needed. */
/* (403) ... and the rows of that list gathered into a zero-padded block: out[rows,C] = src[idx[i],:] for i < n, zeros behind (the grid
 * positions the field is re-evaluated at, forward; the SDF gradient at those vertices, backward -- dmtet.py:228-250 evaluates the field
 * on every grid vertex with a graph; see DMTetGeometry._get_mesh_surface_backward). */
int a3d_dmtet_gather_rows(const float* src, const int64_t* idx, int64_t n, int64_t rows, int C, float* out, a3d_stream_t stream);
int a3d_dmtet_bwd(const float* g_verts, const float* pos, const float* sdf, const int32_t* edges, const int32_t* vert_edge,
                  int V, int Nv, float* g_pos_or_null, float* g_sdf, int g_sdf_is_clear,
                  a3d_stream_t stream); /* (g_sdf_to_clear of a3d_dmtet_emit + g_sdf_is_clear = 1: the emit launch cleared the buffer) */

/* ------------------------------------------------------------------------------------------------
 * Linear-blend skinning -- replaces the per-vertex part of skinning(),
 * /root/reference/model/geometry/skinning.py:377 (weights, :16-22 + geometry/util.py:30-53) and :419-431.
 * T[B,K,12] = per-image, per-bone world transform (rows of the 3x4 affine), composed on the host side from
 * the kinematic chain.  out[b,v] = sum_k softmax_k(-dist(v, bone_k)/temperature) * (T[b,k] . [v,1]).
 * weights_or_null[K,max(Bv,Bb),V] optionally receives the softmax weights (aux['vertices_to_bones']).
 * Backward: g_v[B,V,3] (fully written; per image even when v is shared -- the caller sums over B; may be null)
 * through the affine maps only (weights are detached, skinning.py:377) and g_T[B,K,12] (zeroed by callee).
 */
int a3d_skin_fwd(const float* v, int v_batch, const float* bones /*[Bb,K,2,3]*/, int bones_batch, const float* T, int B, int V,
                 int K, float temperature, float* out /*[B,V,3]*/, float* weights_or_null, float* g_T_to_clear_or_null,
                 a3d_stream_t stream);
int a3d_skin_bwd(const float* g_out, const float* v, int v_batch, const float* bones, int bones_batch, const float* T, int B,
                 int V, int K, float temperature, float* g_v_or_null, float* g_T, int g_T_is_clear, a3d_stream_t stream);
/* (g_T_to_clear: the backward's g_T buffer, cleared by the forward launch; the backward then takes g_T_is_clear = 1 and skips its memset.
 * The same pair exists for a3d_shade_fwd / _bwd (g_par) and a3d_gbuffer_fwd / _bwd (g_rows).) */

/* ------------------------------------------------------------------------------------------------
 * (403) estimate_bones on the device -- the heuristic skeleton of /root/reference/model/geometry/skinning.py:50-248 as ONE
 * launch: pos[N,V,3] (N = B x F instances, <= 32) -> bones[N, n_body + 4 n_leg, 2, 3].  Spine: the extreme-z vertices (body_mode_y_plus:
 * among those not far below the centroid, 'z_minmax_y+'), snapped to x = 0, joined through the lifted centroid; blend[ceil((n_body+1)/2)] =
 * linspace(0, 1, .) of the caller.  Legs (n_leg > 0): the lowest vertex of each top-view quadrant -- margins from the 5 % / 95 % quantiles of
 * x over ALL values of the call, or (use_y_threshold: Fauna) quadrants centred on the medians of x and z among the vertices below the
 * y_threshold quantile of y -- joined to body joint attach[l] by n_leg bones, ramp[n_leg + 1] = linspace(0, 1, .).  attach[0], attach[1] < 0:
 * found here (nearest body bone end in z, instance 0) and attach[2], attach[3] < 0: copies of attach[1], attach[0] (skinning.py:187-216).
 * nearest[2] = the attachment joints of legs 0 / 1 as used (what the caller's kinematic chain needs: its ONE read-back), ok[1] = 0 when a
 * quadrant held no vertex (the reference drops into pdb there, :183; the foot is then vertex 0). */
typedef struct a3d_estimate_bones_args {
    uint32_t size;
    int32_t N;
    const float* pos;
    float* bones;
    int32_t* nearest;
    int32_t* ok;
    int32_t V;
    int32_t n_body;
    int32_t n_leg;
    int32_t body_mode_y_plus;
    int32_t use_y_threshold;
    float y_threshold;
    int32_t attach[4];
    float blend[17];
    float ramp[9];
    float* workspace; /* [3 N V] floats of scratch: the select passes read a planar copy of the coordinate they rank (and, Fauna, the
                         compacted x / z of the low vertices) instead of striding through pos again */
} a3d_estimate_bones_args;
int a3d_estimate_bones(const a3d_estimate_bones_args* args, a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Per-bone world transforms from the kinematic chain -- replaces the chain-composition loops of skinning(),
 * /root/reference/model/geometry/skinning.py:389-417 (+ _estimate_bone_rotation :251-270, euler_angles_to_matrix :315-340,
 * _prepare/_invert_transform_mtx :343-366).  bones[bones_batch,K,2,3] (no gradient), angles[N,K,3] radians (Euler 'XYZ'),
 * chain[K,D] int32: for bone k the bones of its chain root -> ... -> k, front-padded with -1 (D <= 8).
 * M[N,K,12] = rows of the 3x4 affine  L_root ... L_k,  L_i = rotation by angles_i about bone i's start joint in its rest frame.
 * Backward: g_angles[N,K,3] (zeroed by callee) from g_M.
 */
int a3d_bone_transforms_fwd(const float* bones, int bones_batch, const float* angles, const int32_t* chain, int N, int K, int D, float* M,
                            a3d_stream_t stream);
int a3d_bone_transforms_bwd(const float* g_M, const float* bones, int bones_batch, const float* angles, const int32_t* chain, int N, int K,
                            int D, float* g_angles, a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Kinematic chain + skinning in ONE launch each way -- the whole of skinning(), /root/reference/model/geometry/skinning.py:369-439
 * (chain composition :389-417 + the per-vertex blend :377, :419-431), for K <= a3d_skin_pose_max_bones() bones and chains of D <= 8
 * links (every configuration of the reference: 20 bones, depth <= 8).  Forward: every work-group of the skinning launch composes the K
 * transforms of its image in LDS; T_out[B,K,12] receives them (backward, posed_bones), chain_products (when a backward will follow) the
 * prefix / suffix products of every chain position and the derivative of every link by its three angles, which the backward needs
 * (one extra work-group per image computes them beside the vertex work).  Backward: g_v as a3d_skin_bwd; the chain adjoint is
 * linear in the transform gradient, so every work-group applies it to its own share of g_T[b] (kept in LDS; + g_T_extra[b] once: a
 * gradient that reached the transforms directly, e.g. through posed_bones; may be null) and ADDS the result to g_angles[B,K,3] with
 * atomics: g_angles must be zero on entry -- the forward clears it when handed the buffer (then g_angles_is_clear = 1), otherwise the
 * backward memsets it.
 */
int a3d_skin_pose_max_bones(void);
size_t a3d_skin_pose_products_floats(int K, int D); /* floats PER IMAGE of chain_products: K*D*24 (prefix / suffix products) + K*36 (d link / d angle) */
int a3d_skin_pose_fwd(const float* v, int v_batch, const float* bones, int bones_batch, const float* angles, const int32_t* chain, int B,
                      int V, int K, int D, float temperature, float* out, float* T_out, float* chain_products_or_null /*[B, a3d_skin_pose_products_floats(K, D)]*/,
                      float* g_angles_to_clear_or_null, a3d_stream_t stream);
int a3d_skin_pose_bwd(const float* g_out, const float* v, int v_batch, const float* bones, int bones_batch, const float* T,
                      const float* chain_products, const float* angles, const int32_t* chain, int B, int V, int K, int D, float temperature,
                      float* g_v_or_null, const float* g_T_extra_or_null, float* g_angles, int g_angles_is_clear, a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Area-weighted vertex normals -- replaces auto_normals, /root/reference/model/render/mesh.py:276-304.
 * a3d_normals_adjacency: once per triangle list, CSR vertex -> incident corners: off[V+1], adj[3F] (entry = corner*F + face, each
 *   list sorted, i.e. in the order the reference's three scatter_add_ passes visit them, mesh.py:291-293); cursor[V] = scratch.
 * fwd: acc[B,V,3] receives the un-normalised sums (saved for backward), nrm[B,V,3] the result (zero sums -> (0,0,1), then
 *   safe_normalize, mesh.py:296-299).  No float atomics: results are bit-reproducible.
 * bwd: g_nrm is read with a row stride (3 = contiguous [B,V,3]; 16 = the v_nrm columns of a3d_gbuffer_bwd's gradient rows).
 */
int a3d_normals_adjacency(const int32_t* tri /*[F,3]*/, int V, int F, int32_t* off, int32_t* adj, int32_t* cursor, a3d_stream_t stream);
int a3d_normals_fwd(const float* v /*[B,V,3]*/, const int32_t* tri /*[F,3]*/, const int32_t* off, const int32_t* adj, int B, int V, int F,
                    float* acc, float* nrm, int lists_stride, a3d_stream_t stream);
int a3d_normals_bwd(const float* g_nrm /*B*V rows of 3, g_nrm_stride floats apart*/, int g_nrm_stride, const float* acc, const float* v,
                    const int32_t* tri, const int32_t* off, const int32_t* adj, int B, int V, int F, float* g_acc_scratch /*[B,V,3] or NULL with face_scratch*/,
                    float* g_v /*[B,V,3]*/, int lists_stride, float* face_scratch_or_null /*[B,F,9]: faces first -- every face's adjoint once, then a
                    sum per vertex in the same key order: same bits, ~half the gathers, for meshes whose numbering is not spatial*/, a3d_stream_t stream);
/* Two vertex arrays over ONE triangle list in one launch (e.g. the canonical mesh beside the B posed meshes of an iteration: a launch
 * of its own for one image is pure latency); results identical to two a3d_normals_fwd calls. */
int a3d_normals_fwd_pair(const float* v_a /*[B_a,V,3]*/, int B_a, const float* v_b /*[B_b,V,3]*/, int B_b, const int32_t* tri,
                         const int32_t* off, const int32_t* adj, int V, int F, float* acc_a, float* nrm_a, float* acc_b, float* nrm_b,
                         int lists_stride, a3d_stream_t stream);
/* lists_stride: the layout of the vertex -> (corner, face) lists (off, adj), wherever this header takes them.
 *   0 : CSR -- list v = adj[off[v] .. off[v+1]), off[V+1] (a3d_normals_adjacency, a3d_mesh_topology, a3d_mesh_topology_finalize);
 *   S > 0 : fixed stride -- list v = adj[v*S .. v*S + off[v]), off[V] = the list lengths (a3d_dmtet_emit with topo_adj: the emit launch
 *   writes the lists itself, no scan and no further launch).
 * The lists may be stored in any order: the kernels take a vertex's list into registers, order the keys there and issue all gathers
 * at once, so the sums run in ascending key order -- the reference's scatter_add_ order, the same bits -- either way. */

/* ------------------------------------------------------------------------------------------------
 * Per-point shading arithmetic -- replaces the elementwise part of shade(), /root/reference/model/render/render.py:71-93:
 * prepare_shading_normal with perturbed_nrm=None (renderutils/ops.py:194-227 -> bsdf.py:28-51), camera-space normal
 * (render.py:73-74) and DirectionalLight.shade (light.py:186-190), on the covered-pixel list.
 * gb[P,12] as written by a3d_gbuffer_fwd; par rows of w2c rotation (9) | view position (3) | light direction, ambient, diffuse (5) --
 * ncol 12 without a light, 17 with.  With img (point -> image, int64 [P], non-decreasing) par is [B,ncol], one row per image, and
 * the backward reduces g_par[B,ncol] itself (zeroed by callee); with img null par and g_par hold one row per point [P,ncol].
 * kd[P,3] with row stride kd_stride floats.
 * fwd: nrm[P,3]; with a light also shading[P] and shaded[P,3] = shading*kd.   bwd: g_nrm / g_shading / g_shaded may be NULL
 * (= zero); writes g_gb[P,12] (canonical-position columns zero), g_par, g_kd[P,3] (contiguous).
 */
int a3d_shade_fwd(const float* gb, const float* par, int ncol, const int64_t* img_or_null, const float* kd, int kd_stride, int64_t P,
                  int two_sided, float* nrm, float* shading, float* shaded, float* g_par_to_clear_or_null, int B, a3d_stream_t stream);
int a3d_shade_bwd(const float* g_nrm, const float* g_shading, const float* g_shaded, const float* gb, const float* par, int ncol,
                  const int64_t* img_or_null, int B, const float* kd, int kd_stride, int64_t P, int two_sided, float* g_gb, float* g_par,
                  float* g_kd, int g_par_is_clear, a3d_stream_t stream);
/* (round 6) The per-image quantities of the shading where the caller keeps them, instead of a [B,17] table assembled for the call: the
 * camera rotation as the top-left 3x3 of the world-to-camera matrices read in place (rot_row_stride 4; 3 = packed rows), the view
 * positions, the light rows (direction 3 | ambient | diffuse; NULL = no light) -- each with its own image stride in floats (0 = one row
 * shared by every image).  The reference builds these per pixel: w2c[:, :3, :3] (render.py:73-74), view_pos (render.py:139-146),
 * DirectionalLight.forward (light.py:176-184).  The same struct, writable, describes where their gradients are accumulated. */
typedef struct a3d_shade_params {
    uint32_t size;           /* sizeof(a3d_shade_params) of the caller's header */
    int32_t rot_row_stride;  /* floats between two rows of the rotation: 3 or 4 */
    const float* rot;
    const float* view;
    const float* light;      /* NULL: no light */
    int64_t rot_image_stride;
    int64_t view_image_stride;
    int64_t light_image_stride;
} a3d_shade_params;
/* a3d_shade_bwd for the fused render (compositor -> shading adjoint -> G-buffer scatter without torch ops in between): g_shaded[P,3] =
 * the compositor's gradient of the shaded colour; a point's image = pix[p] / pixels_per_image (the covered-pixel list itself: no
 * point -> image array); the colour gradient is written as rows of the texture field's OUTPUT gradient g_tex[tex_rows, tex_cols] (columns
 * 0..2 = d/d kd, the other columns -- ks, the unused normal channels, render.py:66-71 -- and the rows past P -- the padding rows of the
 * field's input -- zero: fully written); g_gb[P,12]; g_par: the rows of par's layout, ACCUMULATED (the caller clears them: the
 * compositor's forward does, a3d_ca_shade.clear). */
int a3d_shade_bwd_rows(const float* g_shaded, const float* gb, const a3d_shade_params* par, const a3d_shade_params* g_par, const int64_t* pix,
                       int64_t pixels_per_image, const float* kd, int kd_stride, int64_t P, int two_sided, float* g_gb, float* g_tex,
                       int tex_cols, int64_t tex_rows, a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Covered-pixel list: flat indices (b*H + y)*W + x of the pixels with rast.w > 0 (triangle_id + 1, as dr.rasterize returns it),
 * image-major and inside an image 8x8-tile by tile (tile = 8; H, W multiples of 8) or row-major (tile = 0).  The list the fused
 * G-buffer / shading path runs over instead of the reference's dense [B,H,W] frame (/root/reference/model/render/render.py:
 * 139-221 shades every pixel; uncovered ones are composited with alpha 0, :261-262).
 * scratch (a3d_cover_scratch_bytes) = int block_count[a3d_cover_blocks] (covered pixels per 256 list positions) followed by
 * int group_sum[a3d_cover_groups * a3d_cover_group_stride] (sums of 64 consecutive blocks, every stride-th word; the rest zero).  a3d_cover_count fills both -- or a3d_rast_fwd's resolve already did
 * (cover_scratch, tile = 8 and H*W a multiple of 256).  The caller reads the group sums back and adds them up: the length of the list.
 * emit writes pix (every work-group derives its own offset from the group sums and block counts before it: no scan launch) and, when
 * given, the inverse map inv[B*H*W] (entry of the list per pixel, -1 = uncovered) that a3d_composite_aa_* reads.
 */
size_t a3d_cover_scratch_bytes(int B, int H, int W);
int a3d_cover_blocks(int B, int H, int W);
int a3d_cover_groups(int B, int H, int W);
int a3d_cover_group_stride(void); /* ints between two group sums (each sits in its own 64-byte line) */
int a3d_cover_count(const float* rast /*[B,H,W,4]*/, int B, int H, int W, int tile, void* scratch, a3d_stream_t stream);
int a3d_cover_emit(const float* rast, int B, int H, int W, int tile, const void* scratch, int64_t* pix /*[total]*/, int32_t* inv_or_null,
                   a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Rasterise -- replaces dr.DepthPeeler(...).rasterize_next_layer() layer 0 / dr.rasterize,
 * /root/reference/model/render/render.py:292-294, :351 (nvdiffrast, third party).
 * clip[clip_batch,V,4]; rast[B,H,W,4] = (u, v, z/w, triangle_id+1), empty = 0.  Per-fragment arithmetic is
 * specified operation by operation in oracle/raster_ref.c.  Backward: gradient of (u,v) w.r.t. clip x,y,w
 * (z/w and the id carry none); g_clip[clip_batch,V,4] zeroed by callee.
 * scratch_is_clean != 0: the caller hands back the scratch buffer of a previous a3d_rast_fwd call of the same size that completed
 * on the same stream (every key is all-ones again: the resolve re-arms what it consumed) -- the 8 B/pixel clear launch is skipped.
 * prev_rast != NULL: depth peeling, DepthPeeler.rasterize_next_layer() for layer n > 0 -- prev_rast[B,H,W,4] is the previous layer;
 * per pixel the nearest fragment strictly behind the previous layer's (depth, id) is returned, empty pixels stay empty.
 * cover_scratch != NULL (H, W multiples of 8, H*W a multiple of 256): the resolve also leaves the covered-pixel list's block counts and
 * group sums (a3d_cover_scratch_bytes) there: a3d_cover_emit can follow without a3d_cover_count.
 * aa_screen / aa_count != NULL: extra work-groups of the triangle launch fill a3d_aa_analyze's `screen` [clip_batch,V,2] and zero its
 * `count` [a3d_aa_shards()], for a3d_aa_analyze(prepared = 1) on the same clip.
 * topo_off / topo_adj / topo_opp (all or none): more extra work-groups fill the opposite-vertex table opp[F,3] of this triangle list from
 * its vertex -> face lists (a3d_aa_topology_from_lists' work, beside the triangle work instead of in a launch of its own) -- for lists
 * that came from a3d_mesh_topology_finalize, which builds no table.
 * normals_v_a != NULL (F > 0): a3d_normals_fwd_pair's work as still more extra work-groups of the triangle launch -- the vertex normals
 * of v_a[B_a,V,3] (the mesh being rasterised: auto_normals, mesh.py:276-304, read by the G-buffer pass of the same frame) and, with
 * B_b > 0, of a second array v_b[B_b,V,3], both over `tri` with the vertex -> face lists normals_off / normals_adj; acc / normals as
 * a3d_normals_fwd writes them, bit for bit.  The gathers of the normals run beside the memory-side atomics of the triangle work
 * instead of in a launch of their own.
 */
size_t a3d_rast_scratch_bytes(int B, int H, int W); /* 64-bit (depth, id) key per pixel */
/* The optional groups described above (round 4: by name, in a struct, instead of 19 positional NULLs; every field may be NULL / 0): */
typedef struct a3d_rast_opts {
    uint32_t size;                 /* sizeof(a3d_rast_opts) of the caller's header (fields are only ever appended) */
    int32_t lists_stride;          /* layout of topo_off / topo_adj and normals_off / normals_adj (see a3d_normals_*: 0 = CSR) */
    const float* prev_rast;        /* depth peeling: the previous layer [B,H,W,4] */
    void* cover_scratch;           /* the covered-pixel list's block counts and group sums are left here */
    float* aa_screen;              /* [clip_batch,V,2] with aa_count: what a3d_aa_analyze(prepared = 1) needs first */
    int32_t* aa_count;             /* [a3d_aa_shards()] */
    const int32_t* topo_off;       /* with topo_adj and topo_opp: the opposite-vertex table from the vertex -> face lists */
    const int32_t* topo_adj;
    int32_t* topo_opp;             /* [F,3] */
    const float* normals_v_a;      /* [normals_B_a,V,3]: the vertex normals of these meshes ride in the triangle launch */
    const float* normals_v_b;      /* [normals_B_b,V,3] or NULL: a second vertex array over the same triangle list */
    const int32_t* normals_off;    /* the vertex -> face lists of `tri` */
    const int32_t* normals_adj;
    float* normals_acc_a;          /* outputs as a3d_normals_fwd writes them */
    float* normals_a;
    float* normals_acc_b;
    float* normals_b;
    int32_t normals_B_a;
    int32_t normals_B_b;
    /* (402) the BINNED path -- the triangle launch appends every triangle to the list of each 8x8 tile its pixel box touches, a second
     * launch (one work-group per 256-pixel block) runs the fragment tests per tile with the depth test in LDS and writes the texels: no
     * memory-side atomics on the frame, no key buffer, no resolve pass; the same triangle ids bit for bit.  Taken when bins != NULL,
     * cover_scratch != NULL, prev_rast == NULL and a3d_rast_bins_bytes(B, H, W, bin_cap) != 0; `scratch` may then be NULL.  A tile whose
     * list overflows bin_cap is rasterised exactly all the same (its block takes every triangle); the third word of the group-sum area
     * of cover_scratch counts such blocks and the second holds the largest tile count above bin_cap / 2 (0 = none), for the caller to
     * size the lists by.  bins_clean: the tile counts are zero (as the previous call on the same stream left them). */
    void* bins;                    /* a3d_rast_bins_bytes(B, H, W, bin_cap) bytes */
    int32_t bin_cap;               /* entries per tile list (>= 16) */
    int32_t bins_clean;
    /* (402) defer_resolve: only the triangle launch runs (cover_scratch required; its look-back flags are zeroed); `rast` is written by the
     * caller's next call -- a3d_rast_resolve_gbuffer_fwd (resolve + covered-pixel list + G-buffer rows in one launch) or a3d_rast_resolve. */
    int32_t defer_resolve;
    int32_t reserved;
} a3d_rast_opts;
size_t a3d_rast_bins_bytes(int B, int H, int W, int bin_cap); /* 0: the frame cannot take the binned path */
/* The second half of a3d_rast_fwd(defer_resolve = 1) on the same `scratch` (keys) and `cover_scratch`:
 * a3d_rast_resolve = the resolve launch alone (texels + block counts + group sums);
 * a3d_rast_resolve_gbuffer_fwd = resolve AND a3d_cover_gbuffer_fwd in ONE launch: every 256-pixel block counts its covered pixels from the
 *   keys, publishes the count and looks the earlier blocks' counts up itself (two-level decoupled look-back over agent-scope flags in
 *   cover_scratch), so the texels are never read back and no launch boundary separates the two.  p_cap = rows allocated for pix / out /
 *   extra_out (the caller does not know P yet): entries past it are dropped; the caller reads P from the group sums as usual and, when
 *   P > p_cap, re-runs a3d_cover_gbuffer_fwd (texels, block counts and sums are complete either way).  Word 3 of the group-sum area is
 *   a status word: non-zero = a look-back ran out of its spin budget (never observed): the list offsets and sums of that call are not to be
 *   trusted, the texels are -- the caller counts from them (a3d_cover_count) and calls a3d_cover_gbuffer_fwd.
 * Replaces the same reference lines as a3d_rast_fwd + a3d_cover_gbuffer_fwd (render.py:292-294, 139-221). */
int a3d_rast_resolve(const float* clip, int clip_batch, const int32_t* tri, int B, int V, int F, int H, int W, float* rast, void* scratch,
                     void* cover_scratch, a3d_stream_t stream);
struct a3d_gb_aux; /* (403; defined with a3d_gbuffer_fwd below) */
int a3d_rast_resolve_gbuffer_fwd(const float* clip, int clip_batch, const int32_t* tri, int B, int V, int F, int H, int W, float* rast,
                                 void* scratch, void* cover_scratch, int64_t p_cap, int64_t* pix, int32_t* inv_or_null, const float* v_pos,
                                 const float* v_nrm, const float* prior, int prior_batch, float* out, const float* extra_or_null, int E,
                                 float* extra_out_or_null, float* g_rows_to_clear_or_null, const struct a3d_gb_aux* aux_or_null,
                                 a3d_stream_t stream);
int a3d_rast_fwd(const float* clip, int clip_batch, const int32_t* tri, int B, int V, int F, int H, int W, float* rast,
                 void* scratch, int scratch_is_clean, const a3d_rast_opts* opts_or_null, a3d_stream_t stream);
/* (403) Probe of the property a3d_rast_resolve_gbuffer_fwd's look-back rests on -- work-groups are dispatched in the order of their
 * linear index -- on this box: n_workgroups work-groups, each waiting (seat held, the resolve's own spin budget) for the flag of the one
 * `stride` before it.  scratch[n_workgroups + 1] ints (cleared by the callee); afterwards scratch[n_workgroups] == 2 <=> every wait
 * ended and the last work-group ran (bit 0 set: a wait timed out -- do not defer the resolve on this device). */
int a3d_dispatch_order_probe(int n_workgroups, int stride, int32_t* scratch, a3d_stream_t stream);
int a3d_rast_bwd(const float* g_rast, const float* rast, const float* clip, int clip_batch, const int32_t* tri, int B, int V,
                 int F, int H, int W, float* g_clip, a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Interpolate -- replaces dr.interpolate(attr, rast, tri) with rast_db=None,
 * /root/reference/model/render/render.py:23-24 (call sites :182-209).
 * attr[attr_batch,V,C]; out[B,H,W,C] = u*A0 + v*A1 + (1-u-v)*A2, 0 where empty.
 * Backward: g_attr[attr_batch,V,C] (zeroed by callee; may be null), g_rast[B,H,W,4] (fully written: du, dv, 0, 0).
 */
int a3d_interp_fwd(const float* attr, int attr_batch, int C, const float* rast, const int32_t* tri, int B, int V, int F, int H,
                   int W, float* out, a3d_stream_t stream);
int a3d_interp_bwd(const float* g_out, const float* attr, int attr_batch, int C, const float* rast, const int32_t* tri, int B,
                   int V, int F, int H, int W, float* g_attr_or_null, float* g_rast, a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Both topology tables of one triangle list in one entry point (5 launches instead of 9): the vertex -> (corner, face) CSR of
 * a3d_normals_adjacency (off[V+1], adj[3F], cursor[V] scratch) and the opposite-vertex table opp[F,3] of a3d_aa_topology
 * (hash = a3d_aa_hash_bytes(F) bytes of scratch).  Same outputs, bit for bit, as the two separate calls.  The last launch leaves
 * cursor and hash re-armed: a caller that hands the same cursor / hash (same hash size, cursor at least V long) to the next call on
 * the same stream passes scratch_is_clean = 1 and saves the init launch.  Once per DMTet call:
 * replaces the per-call index.repeat / scatter_add_ bookkeeping of /root/reference/model/render/mesh.py:276-304 and the
 * topology hash nvdiffrast builds inside dr.antialias (render.py:264-267).
 */
int a3d_mesh_topology(const int32_t* tri, int V, int F, int32_t* off, int32_t* adj, int32_t* cursor, void* hash, int32_t* opp,
                      int scratch_is_clean, a3d_stream_t stream);
/* Second half of the topology for a triangle list that came out of a3d_dmtet_emit with tri32 / topo_count (see there): ONE launch turns
 * the valence counts into off[V+1] (every work-group scans them in LDS) and fills adj[3F]; above 12k vertices, where that scan would be
 * repeated by hundreds of work-groups, a single-work-group scan launch + a fill launch.  The lists of adj are NOT sorted (lists_sorted = 0 for a3d_normals_*: same bits as with a3d_mesh_topology's sorted lists).
 * No opposite-vertex table is built: a3d_aa_analyze finds the few opposite vertices it needs in these lists (opp = NULL, off / adj
 * given).  count_next[v_next] = the count array of the NEXT extraction, zeroed by this launch: callers alternate two arrays. */
int a3d_mesh_topology_finalize_max_vertices(void);
int a3d_mesh_topology_finalize(const int32_t* tri, int V, int F, int32_t* count, int32_t* off, int32_t* adj, int32_t* count_next, int v_next,
                               a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused G-buffer over the covered-pixel list -- replaces the five dr.interpolate calls + face-normal torch ops of
 * render_layer, /root/reference/model/render/render.py:182-209, and in backward also dr.rasterize's gradient.
 * pix[P] = flat indices (b*H + y)*W + x of the covered pixels (int64); out[P,12] =
 * [world position | normalised face normal | interpolated vertex normal | interpolated canonical position].
 * extra (optional, [B,V,E], E <= 3): one more per-vertex attribute interpolated the same way into extra_out[P,E] (the sequence models'
 * 2-D vertex motion, render.py:281-288 / :206-207), so that its backward rides the same scatter instead of dr.interpolate's.
 * Backward: g_rows[B*V, A3D_GBUFFER_GRAD_COLS] (64-byte aligned, zeroed by callee), one 64-byte row per (image, vertex) so that the
 * atomics of a vertex are one line request: columns 0..2 d/d v_pos, 3..5 d/d v_nrm, 6..8 d/d canonical position (per image even
 * when the canonical mesh is shared: the caller sums over B; left zero unless want_prior), 9..11 d/d extra (first E; zero without),
 * 12..15 d/d clip (x, y, 0, w -- through the barycentrics; left zero when clip is null).  Callers take strided views of the rows.
 * clip is [B,V,4].
 */
#define A3D_GBUFFER_GRAD_COLS 16
/* (403) aux: what the texture / feature fields take from the G-buffer, written by the same launch in the form they take it -- so that no
 * torch op stands between this path and model/networks in either direction (render.py:53-57: material.sample(gb_tex_pos, feat)):
 * tex_out[>= P, 3] = the canonical position of every listed pixel as dense rows (the fields' input; its gradient comes back as dense
 * rows too: g_tex of a3d_gbuffer_bwd), img_out[>= P] = the image of every listed pixel (the index into the per-image feature rows).
 * pad_to > 0: the fields see the list padded to a multiple of pad_to rows (GEMM shapes that repeat from step to step); the launch
 * fills the padding rows [P, min(round_up(P, pad_to), rows)) itself -- tex_out zeros, img_out = B - 1 (the index stays non-decreasing) --
 * also where it only learns P on the device (a3d_rast_resolve_gbuffer_fwd: the last work-group of the look-back does it). */
typedef struct a3d_gb_aux {
    uint32_t size;
    int32_t reserved;
    float* tex_out;
    int64_t* img_out;
    int64_t rows;    /* rows allocated behind tex_out / img_out */
    int64_t pad_to;  /* 0: no padding rows */
} a3d_gb_aux;
int a3d_gbuffer_fwd(const float* rast, const int32_t* tri, const int64_t* pix, int64_t P, const float* v_pos, const float* v_nrm,
                    const float* prior, int prior_batch, int B, int V, int F, int H, int W, float* out, const float* extra_or_null, int E,
                    float* extra_out_or_null, float* g_rows_to_clear_or_null, const a3d_gb_aux* aux_or_null, a3d_stream_t stream);
/* g_rows_to_clear (forward, optional): the backward's g_rows buffer, cleared by the forward launch; the backward is then called with
 * g_rows_are_clear = 1 and skips its memset (a caller that runs the backward twice clears the second time itself: flag 0). */
int a3d_gbuffer_bwd(const float* g_out, const float* rast, const int32_t* tri, const int64_t* pix, int64_t P, const float* v_pos,
                    const float* v_nrm, const float* prior, int prior_batch, const float* clip_or_null, int B, int V, int F, int H, int W,
                    float* g_rows, int g_rows_are_clear, int want_prior, const float* extra_or_null, int E, const float* g_extra_out_or_null,
                    const float* g_tex_or_null /* (403) [>= P,3]: the gradient of the canonical position as rows of its own (aux.tex_out's);
                                                  columns 9..11 of g_out are then ignored */,
                    a3d_stream_t stream);
/* (403) g_prior[V,3] = the sum over the B images of columns 6..8 of g_rows: the gradient of a canonical mesh that all images share
 * (prior_batch == 1; render.py:209 interpolates prior_mesh.v_pos for every image). */
int a3d_gbuffer_prior_grad(const float* g_rows, int B, int V, float* g_prior, a3d_stream_t stream);
/* The covered-pixel list AND its G-buffer rows in one launch (= a3d_cover_emit + a3d_gbuffer_fwd; render.py:139-221 on the covered
 * pixels): cover_scratch as for a3d_cover_emit (tile = 8: H, W multiples of 8), P = the list's length (sum of the group sums, read
 * back by the caller), pix[P] / inv[B*H*W] and out[P,12] (+ extra_out[P,E]) written together; every texel is read once. */
int a3d_cover_gbuffer_fwd(const float* rast, const int32_t* tri, int B, int V, int F, int H, int W, const void* cover_scratch, int64_t P,
                          int64_t* pix, int32_t* inv_or_null, const float* v_pos, const float* v_nrm, const float* prior, int prior_batch,
                          float* out, const float* extra_or_null, int E, float* extra_out_or_null, float* g_rows_to_clear_or_null,
                          const a3d_gb_aux* aux_or_null, a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * out[B,C] (zeroed by callee) = per-image sums of g[P,C] under the point -> image map img[P] (int64): the adjoint of
 * broadcasting per-image rows (feat / light / w2c, render.py:53-94) to the covered pixels.  Replaces torch index_add.
 */
int a3d_rows_segsum(const float* g, const int64_t* img, int64_t P, int C, int B, float* out, a3d_stream_t stream);
/* y[p,:] = max(y[p,:] + rows[img[p],:], 0) in place -- a per-image addend folded into the ReLU that follows a GEMM over the point
 * list (the per-image feature path of the texture field: /root/reference/model/networks/MLPs.py:84-90 concatenates the feature to
 * every point instead).  bwd: g_pre = g * (y > 0) (the gradient of the GEMM output) and g_rows[B,C] = its per-image sums, one pass. */
int a3d_rows_add_relu_fwd(float* y /*[P,C] in/out*/, const float* rows /*[B,C]*/, const int64_t* img /*[P]*/, int64_t P, int C, int B,
                          a3d_stream_t stream);
int a3d_rows_add_relu_bwd(const float* g, const float* y, const int64_t* img, int64_t P, int C, int B, float* g_pre /*[P,C]*/,
                          float* g_rows /*[B,C], zeroed by callee*/, a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Antialias -- replaces dr.antialias(color, rast, pos, tri), /root/reference/model/render/render.py:264-267.
 *   a3d_aa_topology : once per mesh topology: opp[F,3] = vertex opposite edge i in the adjacent triangle, -1 on
 *                     a boundary (nvdiffrast's topology hash).  hash = scratch of a3d_aa_hash_bytes(F) bytes.
 *   a3d_aa_analyze  : once per (rast, clip): finds every silhouette crossing between adjacent pixels; work =
 *                     scratch of capacity*16 bytes with capacity >= a3d_aa_capacity(B, H, W) (the list is kept in
 *                     a3d_aa_shards() segments, each with its own append counter), count[a3d_aa_shards()] zeroed by callee,
 *                     screen = scratch of clip_batch*V*2 floats (pixel-space vertex positions).  Per pair the geometric tests
 *                     (exit edge, slope, crossing distance) run first; the silhouette test needs the neighbouring triangle and runs
 *                     only for the pairs that pass them, for the one exit edge.
 *   a3d_aa_fwd      : out = color, then blends across each recorded crossing; any number of colour buffers can
 *                     share one analysis (the reference re-analyses per buffer, render.py:311-315).
 *   a3d_aa_bwd      : g_color[B,H,W,C] and g_clip[clip_batch,V,4] (both fully written / zeroed by callee).
 */
size_t a3d_aa_hash_bytes(int F);
int a3d_aa_shards(void);
int a3d_aa_capacity(int B, int H, int W);
int a3d_aa_topology(const int32_t* tri, int F, int V, void* hash, int32_t* opp, a3d_stream_t stream);
int a3d_aa_topology_from_lists(const int32_t* tri, int F, const int32_t* off, const int32_t* adj, int32_t* opp, int lists_stride,
                               a3d_stream_t stream); /* the same table from the vertex -> face lists (no hash) */
int a3d_aa_analyze(const float* rast, const float* clip, int clip_batch, const int32_t* tri, const int32_t* opp_or_null, int B, int V,
                   int F, int H, int W, float* screen, void* work, int capacity, int32_t* count, int prepared,
                   const int32_t* off_or_null, const int32_t* adj_or_null, int lists_stride, a3d_stream_t stream);
/* (opp_or_null = NULL with off / adj = the vertex -> face lists of a3d_normals_adjacency / a3d_mesh_topology[_finalize]: the opposite
 * vertex is looked up in the lists, only for the pixel pairs that passed every geometric test -- same records as with the table) */
int a3d_aa_fwd(const float* color, int C, const void* work, const int32_t* count, int capacity, int B, int H, int W, float* out,
               a3d_stream_t stream);
int a3d_aa_bwd(const float* g_out, const float* color, int C, const void* work, const int32_t* count, int capacity,
               const float* clip, int clip_batch, const int32_t* tri, int B, int V, int F, int H, int W, float* g_color,
               float* g_clip, a3d_stream_t stream);
/* Compositing fused with the antialiasing -- the lerp(bg, [value, 1], coverage) of /root/reference/model/render/render.py:261-262 followed
 * by dr.antialias (render.py:311-315) for a buffer that exists as rows vals[P,C] at the covered pixels (pix / inv of a3d_cover_emit):
 *   out[B,H,W,C+1] = [vals row, 1] at covered pixels, bg[bg_batch,H,W,C+1] (null = zeros) elsewhere, then the blends of a3d_aa_fwd
 *   computed from the same sources.  One pass over the image instead of fill + scatter + copy + blend.
 * A second buffer (vals2 / C2 / bg2 / out2; null = none) against the same pixel list and crossing records rides in the same launches
 * (render_mesh antialiases the colour and the feature image of a step).
 * bwd: g_vals[P,C] (fully written: g_out at the covered pixels + the blend adjoints; no dense colour gradient exists) -- likewise
 *   g_vals2 -- and g_clip[clip_batch,V,4] (zeroed by callee; both buffers add to it).  The backgrounds receive no gradient. */
/* The silhouette analysis riding in the first launch of a compositor call (round 4: by name instead of ten positional arguments): the
 * arguments of a3d_aa_analyze(prepared = 1) -- rast, the `screen` and zeroed `count` that a3d_rast_fwd left (count is the compositor's own
 * argument), tri, opp or the vertex -> face lists. */
typedef struct a3d_aa_ride {
    uint32_t size;        /* sizeof(a3d_aa_ride) of the caller's header (fields are only ever appended) */
    int32_t clip_batch;
    const float* rast;    /* [B,H,W,4] */
    const float* screen;  /* [clip_batch,V,2] */
    const int32_t* tri;   /* [F,3] */
    const int32_t* opp;   /* [F,3] or NULL with off / adj */
    const int32_t* off;
    const int32_t* adj;
    int32_t V;
    int32_t F;
    int32_t lists_stride; /* layout of off / adj (see a3d_normals_*) */
    int32_t reserved;
} a3d_aa_ride;
/* The shaded colour of the FIRST buffer computed on the spot (round 4): with `shade` and vals = NULL (C = 3) the value of point q is
 * kd[q] * shading(q) -- a3d_shade_fwd's arithmetic, bit for bit (one shared device function), from the G-buffer row, the image's
 * camera / light row (ncol 17) and kd -- instead of a [P,3] array that a launch of its own wrote: a3d_shade_fwd is then not called at all
 * (its other outputs, the shading normal and the shading term, are not produced: for render modes that want them the caller runs it).
 * `clear`: n_clear floats zeroed by the forward's first launch (the per-image row gradient a3d_shade_bwd accumulates into: that clear was
 * a3d_shade_fwd's).  The backward takes the same struct (its blend adjoints read the same sources); g_vals is the gradient of the
 * shaded colour, which a3d_shade_bwd consumes as before. */
typedef struct a3d_ca_shade {
    uint32_t size;      /* sizeof(a3d_ca_shade) of the caller's header (fields are only ever appended) */
    int32_t kd_stride;  /* floats between two rows of kd */
    const float* gb;    /* [P,12] rows of a3d_cover_gbuffer_fwd / a3d_gbuffer_fwd */
    const float* par;   /* [B,17] w2c rotation (9) | view position (3) | light direction (3), ambient, diffuse */
    const float* kd;    /* [P,3] */
    float* clear;
    int32_t n_clear;
    int32_t two_sided;
    const a3d_shade_params* params; /* (403) non-NULL: the camera / light rows where the caller keeps them, `par` is ignored */
    float* shaded_out;  /* (403) forward, optional [P,3]: the colour of every covered pixel as the compose launch computed it, kept (12 B per
                         * point) so that the blend launch and the whole backward read it as plain value rows (first buffer's `vals`, no
                         * `shade` struct) instead of re-deriving the shading per crossing record and channel */
} a3d_ca_shade;
/* One buffer of a compositor call (round 4: the two buffers of a call by name instead of ten / twelve positional arguments). */
typedef struct a3d_ca_buffer {
    uint32_t size;       /* sizeof(a3d_ca_buffer) of the caller's header (fields are only ever appended) */
    int32_t C;           /* channels of vals; the image has C + 1 */
    const float* vals;   /* [P,C] (NULL with a3d_ca_shade, first buffer only) */
    const float* bg;     /* [bg_batch,H,W,C+1] or NULL = zeros */
    float* out;          /* forward: [B,H,W,C+1] */
    const float* g_out;  /* backward: [B,H,W,C+1] */
    float* g_vals;       /* backward: [P,C], fully written */
    int32_t bg_batch;    /* 1 or B */
    int32_t reserved;
    /* (403) what a caller needs to hand the reference's own tensors over without torch copies around the call (0 = the defaults above): */
    int32_t bg_channels; /* channels stored per background pixel (<= C+1; the missing trailing ones read as 0): the reference's background is
                          * [B,H,W,3] and gets its zero alpha appended per call (render.py:254-256) -- here it is read as it is */
    int32_t g_stride;    /* backward: floats between two pixels of g_out (default C+1) */
    int32_t g_channels;  /* backward: leading channels of the image that HAVE a gradient in g_out (default C+1; the rest: zero) -- render_mesh
                          * returns dino_pred / flow without their alpha channel (render.py:320-331), so their gradient comes without it */
    int32_t out_channels;/* forward: leading channels of the composited image that are MATERIALISED: out is [B,H,W,out_channels] (default C+1).
                          * A mode whose alpha channel the caller cuts off anyway (dino_pred, flow: render.py:326-331) is written without it. */
    int64_t vals_rows;   /* backward: rows of g_vals to write (default P; >= P: the rows past P -- padding rows of a field's point list -- zero) */
} a3d_ca_buffer;
int a3d_composite_aa_fwd(const a3d_ca_buffer* first, const a3d_ca_buffer* second_or_null, const int32_t* inv, void* work, int32_t* count,
                         int capacity, int B, int H, int W, const a3d_aa_ride* analyze_or_null, const a3d_ca_shade* shade_or_null,
                         a3d_stream_t stream);
/* analyze != NULL: the records do not exist yet -- a3d_aa_analyze(prepared = 1)'s launch (same arguments: rast, the `screen` and
 * zeroed `count` that a3d_rast_fwd left, tri, opp or the lists) runs as extra work-groups of this call's first launch, which only moves
 * pixels; the blend launch that follows is the first consumer of `work` / `count`.  Same records as the stand-alone analysis. */
/* The same for a render without texture and light -- every covered pixel is (1, .., 1, alpha = 1), only the silhouette is differentiated:
 * Fauna's random-view mask (/root/reference/model/models/Fauna.py:111-173: render_mesh(material = None, lgt = None, ['shaded']), of which
 * only the alpha channel is used).  Coverage comes straight from the raster texels (id channel > 0): no covered-pixel list, no G-buffer,
 * no shading, no host read-back.  out[B,H,W,C+1]; analyze != NULL: the silhouette analysis rides in the first launch (see
 * a3d_composite_aa_fwd).  bwd: g_clip[clip_batch,V,4] (zeroed by callee). */
int a3d_mask_aa_fwd(const float* rast, int C, const float* bg_or_null, int bg_batch, float* out, void* work, int32_t* count, int capacity,
                    int B, int H, int W, const a3d_aa_ride* analyze_or_null /* (its rast = this call's) */, a3d_stream_t stream);
int a3d_mask_aa_bwd(const float* g_out, const float* rast, int C, const float* bg_or_null, int bg_batch, const void* work,
                    const int32_t* count, int capacity, const float* clip, int clip_batch, const int32_t* tri, int B, int V, int F, int H,
                    int W, float* g_clip,
                    int g_channels_first /* (404) 1: g_out is [B,C+1,H,W] -- what the caller's autograd hands back when the image went on as
                                            permute(0, 3, 1, 2) and a channel slice, Fauna.py:166-173 -- read in place; 0: [B,H,W,C+1] */,
                    a3d_stream_t stream);
int a3d_composite_aa_bwd(const a3d_ca_buffer* first, const a3d_ca_buffer* second_or_null, const int64_t* pix, int64_t P, const int32_t* inv,
                         const void* work, const int32_t* count, int capacity, const float* clip, int clip_batch, const int32_t* tri, int B,
                         int V, int F, int H, int W, float* g_clip, const a3d_ca_shade* shade_or_null, a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * fp32 MFMA GEMM with the ReLU adjoint in the epilogue: C[M,N] = (A[M,K] . B[K,N]) * (X[M,N] > 0), row-major, N = 256,
 * K % 32 == 0, X may be NULL (plain product).  The input-gradient GEMM of a 256-wide Linear over a long point list whose input
 * X is the previous layer's ReLU output (/root/reference/model/networks/MLPs.py:9-32, the Linear/ReLU stack): the mask is that
 * layer's ReLU backward, which the reference (PyTorch autograd) runs as a separate pass.
 */
int a3d_gemm_nn_relumask(const float* A, const float* B, const float* X, int64_t M, int N, int K, float* C, a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Harmonic embedding of the field inputs -- /root/reference/model/networks/HarmonicEmbedding.py:33-44 as CoordMLP applies it
 * (networks/MLPs.py:73-83): out[P, 3+6n(+1)] = [x (|x_0| if symmetrize), sin(x_c f_k), cos(x_c f_k) (c-major), (1)].
 * The optional ones column folds the first Linear's bias into its weight.  bwd: g_x[P,3].
 */
int a3d_harmonic_embed_fwd(const float* x /*[P,3]*/, const float* freq /*[n]*/, int n, int symmetrize, int ones, int64_t P, float* out,
                           a3d_stream_t stream);
int a3d_harmonic_embed_bwd(const float* g_out, const float* x, const float* freq, int n, int symmetrize, int ones, int64_t P, float* g_x,
                           a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Reconstruction losses -- the image-space consumers of render_mesh's outputs, fused:
 * /root/reference/model/models/AnimalModel.py:260-307 (compute_reconstruction_losses, background_mode 'none'; B here = images x frames).
 * shaded[B,H,W,4] / dino[B,H,W,D] = the renderer's NHWC buffers (D = 0: no DINO term); image_gt[B,3,H,W], dino_gt[B,D,H,W],
 * mask_gt / valid [B,H,W], dt0 = mask_dt[:,0] and dt1 = mask_dt[:,1] (may be NULL) with image stride dt_stride floats.
 * loss[B,a3d_recon_losses_columns()] = per-image mask, mask_inv_dt, rgb, dino, mask_dt.  bwd: g_loss -> g_shaded[B,H,W,4],
 * g_dino[B,H,W,D] (every element written).  both[B*H*W] = the eroded common mask, written by fwd, read by bwd and by the flow loss.
 * (403) dino_stride / g_dino_stride: floats between two pixels of dino / g_dino (D = contiguous).  render_mesh returns 'dino_pred' as
 * the first D channels of a (D+1)-channel image (render.py:330-331): with stride D + 1 the image is read where it is and its gradient
 * is written in the same layout (the alpha slot untouched), where the compositor's backward reads it (a3d_ca_buffer.g_stride /
 * g_channels) -- no contiguous copy forward, no zero-padded gradient backward.
 *
 * Flow loss between consecutive frames (AnimalModel.py:285-298): flow = the renderer's 'flow' buffer over B*F frames with pix_stride
 * floats per pixel (3: two flow channels + alpha), flow_gt[B,F-1,2,H,W]; loss[B,F-1]; scale[B,F-1] is kept for the backward;
 * g_flow[B*F,H,W,2] (every element written; the last frame of a sequence gets zeros).

 */
size_t a3d_recon_losses_scratch_bytes(int B, int H, int W);
size_t a3d_recon_losses_mask_bytes(int B, int H, int W);
int a3d_recon_losses_columns(void);
int a3d_recon_losses_fwd(const float* shaded, const float* dino, int D, int dino_stride, const float* image_gt, const float* dino_gt, const float* mask_gt,
                         const float* dt0, const float* dt1_or_null, int64_t dt_stride, const float* valid, int B, int H, int W, void* scratch,
                         uint8_t* both, float* loss, a3d_stream_t stream);
int a3d_recon_losses_bwd(const float* g_loss, const float* shaded, const float* dino, int D, int dino_stride, int g_dino_stride, const float* image_gt, const float* dino_gt,
                         const float* mask_gt, const float* dt0, const float* dt1_or_null, int64_t dt_stride, const float* valid, int B, int H,
                         int W, const uint8_t* both, float* g_shaded, float* g_dino, a3d_stream_t stream);
size_t a3d_flow_loss_scratch_bytes(int B, int F, int H, int W);
int a3d_flow_loss_fwd(const float* flow, int pix_stride, const float* flow_gt, const uint8_t* both, int B, int F, int H, int W, void* scratch,
                      float* loss, float* scale, a3d_stream_t stream);
int a3d_flow_loss_bwd(const float* g_loss, const float* scale, const float* flow, int pix_stride, const float* flow_gt, const uint8_t* both,
                      int B, int F, int H, int W, float* g_flow, a3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Clip-space transform (403): xfm_points(points, matrix, use_python=True), /root/reference/model/render/renderutils/ops.py:515-531, the
 * first line of render_mesh (render.py:279): out[B,V,4] = [points,1] . matrix^T, points[points_batch,V,3], matrix[matrix_batch,4,4]
 * (batches 1 or B).  bwd: g_out with a vertex stride of g_stride floats (4 = contiguous; 16 = the clip columns of a3d_gbuffer_bwd's
 * gradient rows read in place), g_points[B,V,3] (per image also when points are shared: the caller sums), g_matrix[matrix_batch,4,4]
 * accumulated with atomics (zeroed by the callee unless g_matrix_is_clear: the forward clears it when handed the buffer).
 */
int a3d_xfm_points_fwd(const float* points, int points_batch, const float* matrix, int matrix_batch, int B, int V, float* out,
                       float* g_matrix_to_clear_or_null, a3d_stream_t stream);
int a3d_xfm_points_bwd(const float* g_out, int g_stride, const float* points, int points_batch, const float* matrix, int matrix_batch, int B, int V,
                       float* g_points_or_null, float* g_matrix_or_null, int g_matrix_is_clear,
                       const float* g_points_addend_or_null /* [B,V,3] with a vertex stride of addend_stride floats, added to g_points: the
                                                               gradient the same points get from their other consumer in render_mesh (the
                                                               position columns of a3d_gbuffer_bwd's rows) -- no accumulation launch */,
                       int addend_stride,
                       const float* g_points_addend2_or_null /* (404) a third gradient of the same points, added after the second: the one
                                                                a3d_normals_bwd wrote for the posed mesh (mesh.py:276-304 reads v_pos too) */,
                       int addend2_stride,
                       const float* g_out2_or_null /* (404) a second gradient of the clip positions themselves, [B,V,4] with a vertex stride of
                                                      g2_stride floats: g_out2 + g_out is what is transformed (the antialiasing's gradient
                                                      w.r.t. the clip positions, render.py:264-268, which otherwise meets the rasteriser's in
                                                      an accumulation launch of the caller's autograd engine) */,
                       int g2_stride, a3d_stream_t stream);
/* (404) The 2-D motion of every vertex to the next frame of its sequence -- /root/reference/model/render/render.py:281-288:
 * ndc = clip[..., :2] / clip[..., -1:], delta[b,f] = ndc[b,f+1] - ndc[b,f], zeros for the last frame of a sequence.  clip[N,V,4] with
 * N = B*F frames (sequence-major), delta[N,V,2]; bwd: g_clip[N,V,4] fully written (z column zero).  One launch each way for the ~25 torch
 * launches of the expression (the Ponymation step renders 'flow'). */
int a3d_flow_delta_fwd(const float* clip, int N, int F, int V, float* delta, a3d_stream_t stream);
int a3d_flow_delta_bwd(const float* g_delta, int g_stride /* floats between two vertices of g_delta (2 = contiguous) */, const float* clip, int N,
                       int F, int V, float* g_clip, a3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* A3D_H */
