#!/usr/bin/env python
"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch) into HBM traffic per C-ABI call.

usage: python tools/pmc_traffic.py <fetch-dir> <write-dir> <out.json> [calibration.json]

gfx950 correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide coalesced
streaming reads, so traffic = 2*FETCH + WRITE.  Calibrated in round 4 on known byte counts (tools/pmc_calibrate.sh ->
profiles/r04_pmc_calibration.json, carried in the output under "calibration"): WRITE_SIZE is EXACT for streaming stores (16- and
4-byte alike); a scattered 4-byte store and a fire-and-forget 64-bit device atomic each count 32 B of WRITE (and nothing of FETCH), a
16-lane float atomicAdd onto one 64-byte row 64 B; FETCH_SIZE is halved for every COALESCED load (4-byte lanes too) but counts a
random 4-byte gather as its full 64-byte line -- so 2*FETCH + WRITE over-counts a gather-dominated kernel by up to 2x; that bound
(FETCH + WRITE) is reported next to it.  Infinity-Cache hits are counted, i.e. this is memory-side traffic of the L2s, not DRAM
traffic; lines still dirty in an L2 when a kernel ends are written back later, under another kernel's name.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

# entry point -> (kernels of one call, kernel whose dispatch count = number of calls)
ENTRY = {
    # (the culled form -- sign plane, culled count, scan -- on grids with word groups; the plain count otherwise)
    # (round 6: the scan rides in the culled count launch -- one dm_sign_kernel per call there; the plain form has no sign pre-pass below 2^20 vertices)
    "a3d_dmtet_count": (["dm_sign_kernel", "dm_count_cull_kernel", "dm_count_kernel", "dm_scan_kernel"], "dm_sign_kernel"),
    "a3d_dmtet_emit": (["dm_emit_kernel"], "dm_emit_kernel"),
    "a3d_dmtet_bwd": (["dm_bwd_kernel"], "dm_bwd_kernel"),
    "a3d_skin_fwd": (["sk_fwd_kernel<20, false>", "sk_fwd_kernel<32, false>", "sk_fwd_kernel<64, false>"], None),
    "a3d_skin_bwd": (["sk_bwd_kernel<5, false>", "sk_bwd_kernel<8, false>", "sk_bwd_kernel<16, false>"], None),
    "a3d_skin_pose_fwd": (["sk_fwd_kernel<20, true>"], None),
    "a3d_skin_pose_bwd": (["sk_bwd_kernel<5, true>"], None),
    "a3d_mesh_topology_finalize": (["tp_finalize_kernel"], "tp_finalize_kernel"),
    "a3d_cover_gbuffer_fwd": (["gb_cover_fwd_kernel"], "gb_cover_fwd_kernel"),
    "a3d_bone_transforms_fwd": (["bn_fwd_kernel"], "bn_fwd_kernel"),
    "a3d_bone_transforms_bwd": (["bn_bwd_kernel"], "bn_bwd_kernel"),
    "a3d_mesh_topology": (["tp_init_kernel", "tp_count_insert_kernel", "nr_adj_scan_kernel", "tp_fill_lookup_kernel", "nr_adj_sort_kernel"],
                          "nr_adj_scan_kernel"),
    "a3d_normals_fwd": (["nr_fwd_kernel"], "nr_fwd_kernel"),
    "a3d_normals_bwd": (["nr_vert_bwd_kernel", "nr_bwd_kernel", "nr_face_bwd_kernel", "nr_sum_bwd_kernel", "nr_face12_bwd_kernel", "nr_sum12_bwd_kernel"], ("nr_sum_bwd_kernel", "nr_sum12_bwd_kernel", "nr_bwd_kernel")),
    # (the triangle launch also carries the vertex normals of the step: nr_fwd's work as extra work-groups)
    "a3d_rast_fwd": (["rs_clear_kernel", "rs_tri_kernel", "rs_resolve_kernel"], "rs_tri_kernel"),
    # (round 5: the resolve of a deferred rasterisation + the covered-pixel list + the G-buffer rows, one launch)
    "a3d_rast_resolve_gbuffer_fwd": (["rs_resolve_cover_kernel"], "rs_resolve_cover_kernel"),
    "a3d_cover_count": (["cv_count_kernel"], "cv_count_kernel"),
    "a3d_cover_emit": (["cv_emit_kernel"], "cv_emit_kernel"),
    "a3d_gbuffer_fwd": (["gb_fwd_kernel"], "gb_fwd_kernel"),
    "a3d_gbuffer_bwd": (["gb_bwd_kernel"], "gb_bwd_kernel"),
    "a3d_shade_fwd": (["sh_fwd_kernel"], "sh_fwd_kernel"),
    "a3d_shade_bwd": (["sh_bwd_kernel"], "sh_bwd_kernel"),
    # (round 6: the same kernel launched by the compositor's backward node, gradients written in the consumers' layouts)
    "a3d_shade_bwd_rows": (["sh_bwd_kernel"], "sh_bwd_kernel"),
    "a3d_estimate_bones": (["eb_kernel"], "eb_kernel"),
    "a3d_gbuffer_prior_grad": (["gb_prior_sum_kernel"], "gb_prior_sum_kernel"),
    "a3d_xfm_points_fwd": (["xf_fwd_kernel"], "xf_fwd_kernel"),
    "a3d_xfm_points_bwd": (["xf_bwd_kernel"], "xf_bwd_kernel"),
    "a3d_dmtet_gather_rows": (["dm_gather_rows_kernel"], "dm_gather_rows_kernel"),
    "a3d_flow_delta_fwd": (["xf_flow_fwd_kernel"], "xf_flow_fwd_kernel"),
    "a3d_flow_delta_bwd": (["xf_flow_bwd_kernel"], "xf_flow_bwd_kernel"),
    "a3d_rows_add_relu_fwd": (["ss_add_relu4_kernel", "ss_add_relu1_kernel"], "ss_add_relu4_kernel"),
    "a3d_rows_add_relu_bwd": (["ss_kernel<4, true", "ss_kernel<1, true"], None),
    "a3d_rows_segsum": (["ss_kernel<4, false>", "ss_kernel<1, false>"], None),
    "a3d_gemm_nn_relumask": (["gm_nn3_kernel", "gm_nn_kernel"], None),
    "a3d_harmonic_embed_fwd": (["he_fwd_kernel"], "he_fwd_kernel"),
    "a3d_harmonic_embed_bwd": (["he_bwd_kernel"], "he_bwd_kernel"),
    "a3d_recon_losses_fwd": (["ls_fwd_kernel", "ls_finish_kernel"], "ls_fwd_kernel"),
    "a3d_recon_losses_bwd": (["ls_bwd_kernel"], "ls_bwd_kernel"),
    "a3d_aa_analyze": (["aa_screen_kernel", "aa_analyze_kernel"], "aa_analyze_kernel"),
    "a3d_aa_fwd": (["aa_fwd_kernel"], "aa_fwd_kernel"),
    "a3d_aa_bwd": (["aa_copy_zero_kernel", "aa_bwd_kernel"], "aa_bwd_kernel"),
    # (one call per step handles the 4- and the 17-channel buffer together; its compose launch also carries the silhouette analysis)
    "a3d_composite_aa_fwd": (["ca_compose_kernel", "ca_blend_kernel"], "ca_blend_kernel"),
    "a3d_composite_aa_bwd": (["ca_gather_kernel", "ca_bwd_kernel"], "ca_gather_kernel"),
    "a3d_dmtet_count_ordered": (["dm_sign_order_kernel", "dm_count_order_kernel", "dm_scan_kernel"], "dm_count_order_kernel"),
    "a3d_dmtet_emit_sparse": (["dm_emit_words_kernel"], "dm_emit_words_kernel"),
    "a3d_flow_loss_fwd": (["fl_fwd_kernel", "fl_finish_kernel"], "fl_fwd_kernel"),
    "a3d_flow_loss_bwd": (["fl_bwd_kernel"], "fl_bwd_kernel"),
}


def load(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
            key = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
            tot[key] += float(r["Counter_Value"]) * 1024.0  # KiB -> bytes
            cnt[key] += 1
    return tot, cnt


def main():
    fetch, fcnt = load(sys.argv[1], "FETCH_SIZE")
    write, wcnt = load(sys.argv[2], "WRITE_SIZE")
    per_call = {}
    for entry, (kernels, main_k) in ENTRY.items():
        def pick(tot, cnt):
            ks = [k for k in tot if any(k.startswith(s) or k == s for s in kernels)]
            if entry in ("a3d_gbuffer_fwd",):  # (prefix of gb_fwd_kernel only, not gb_cover_fwd_kernel)
                ks = [k for k in ks if not k.startswith("gb_cover")]
            if not ks:
                return None
            # (the kernel that runs once per call, by PREFIX: template arguments are part of the name -- an exact match missed
            # rs_tri_kernel<4> and the sum was divided by the dispatches of both kernels, i.e. halved: round 3's 27 MB for a3d_rast_fwd)
            mains = (main_k,) if isinstance(main_k, str) else (main_k or ())
            calls = sum(cnt[k] for k in cnt if any(k == m or k.startswith(m + "<") for m in mains)) if mains else sum(cnt[k] for k in ks)
            calls = calls or sum(cnt[k] for k in ks)
            return sum(tot[k] for k in ks) / calls
        fe, wr = pick(fetch, fcnt), pick(write, wcnt)
        if fe is None and wr is None:
            continue
        fe, wr = fe or 0.0, wr or 0.0
        per_call[entry] = dict(fetch_MB_raw=round(fe / 1e6, 2), write_MB=round(wr / 1e6, 2), traffic_MB=round((2 * fe + wr) / 1e6, 2),
                               traffic_if_all_reads_were_gathers_MB=round((fe + wr) / 1e6, 2))
    import os

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    calibration = None
    if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
        calibration = json.load(open(sys.argv[4]))["factors"]
    out = dict(kernel_source_sha16=bench.kernel_source_sha16(), workload=bench.PMC_WORKLOAD, calibration=calibration, note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_traffic.sh), bench workload B=16 256x256 Kuhn R=64, "
                    "MB per C-ABI call; traffic = 2*FETCH (gfx950: every coalesced load is tallied at half its bytes; a random gather at its full line, "
                    "so this is an upper bound for gather-dominated kernels, whose lower bound FETCH + WRITE is given too) + WRITE (exact for stores; 32 B "
                    "per device atomic or scattered 4-byte store); "
                    "the copy/memset helpers of an entry point (hipMemcpyAsync / hipMemsetAsync) are not included", per_call=per_call)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in per_call.items():
        print(f"{k:28s} fetch {v['fetch_MB_raw']:9.2f} MB (x2 = {2 * v['fetch_MB_raw']:9.2f})  write {v['write_MB']:9.2f} MB  traffic {v['traffic_MB']:9.2f} MB")


if __name__ == "__main__":
    main()
