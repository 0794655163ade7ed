// Error string + version for the flat C ABI (include/a3d.h).
#include <stdarg.h>
#include <stdlib.h>

#include "a3d_common.h"

static thread_local char g_err[512] = "";

void a3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* a3d_last_error(void) { return g_err; }
extern "C" int a3d_version(void) { return 401; /* 0.4.x: round-4 ABI (400: a3d_dmtet_count_ordered -- the culled count pass for grids in any numbering; 401: a3d_dmtet_emit_sparse, a3d_mask_aa_*, the optional groups of a3d_rast_fwd / a3d_dmtet_emit / a3d_composite_aa_fwd in structs); 0.3.x: round-3 ABI (skin_pose, scan-free covered-pixel list and DMTet, topology inside the DMTet emit; 301: culled DMTet count; 302: normals ride in the rasteriser launch; 303: the silhouette analysis rides in the compositor launch; 304: the DMTet emit writes the vertex -> face lists itself; 305: ... and covers only the blocks the count pass listed; 306: skin_pose_bwd without ticket; 307: link derivatives from the forward; 308: speculative DMTet emit) */ }

int a3d_exp(void) {
    const char* e = getenv("A3D_EXP");
    return e ? atoi(e) : 0;
}
