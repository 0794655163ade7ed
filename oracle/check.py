"""Oracle-side checker for one SyntheticScene step (used by __graft_entry__.smoke() and tests).  TEST INFRASTRUCTURE ONLY.

Recomputes, on CPU with the oracle, every stage of the step the HIP path just ran -- from the very same SDF values,
bones, angles, cameras and network weights -- and reports the largest deviations.
"""
import copy

import numpy as np
import torch

from . import dmtet_ref, mesh_ref, raster_ref, render_ref, skinning_ref


def compare_step(scene, out):
    geo = scene.netShape
    cpu = lambda t: t.detach().float().cpu()
    pos, sdf, tets = cpu(geo.current_pos), cpu(geo.current_sdf).reshape(-1), geo.indices.cpu()
    verts, faces, _, uv_idx = dmtet_ref.marching_tets(pos, sdf, tets)
    prior, shape = scene.last["prior"], scene.last["shape"]
    rep = {}
    rep["faces_equal"] = bool(np.array_equal(faces.numpy(), prior.t_pos_idx[0].cpu().numpy())
                              and np.array_equal(uv_idx.numpy(), prior.t_tex_idx[0].cpu().numpy()))
    rep["num_verts"], rep["num_faces"] = int(verts.shape[0]), int(faces.shape[0])
    if not rep["faces_equal"]:
        rep["max_abs_image_err"] = float("inf")
        rep["loss"] = float(out["loss"])
        return rep
    rep["max_abs_vert_err"] = float((verts - cpu(prior.v_pos[0])).abs().max())
    nrm = mesh_ref.vertex_normals(verts[None], faces)
    rep["max_abs_prior_normal_err"] = float((nrm - cpu(prior.v_nrm)).abs().max())
    bones, arti = cpu(scene.bones), cpu(scene.arti)
    sk, _ = skinning_ref.skinning(verts[None, None], bones, scene.kinematic_tree, arti, scene.temperature)
    sk = sk.view(scene.batch, -1, 3)
    rep["max_abs_skin_err"] = float((sk - cpu(shape.v_pos)).abs().max())
    snrm = mesh_ref.vertex_normals(sk, faces)
    rep["max_abs_posed_normal_err"] = float((snrm - cpu(shape.v_nrm)).abs().max())
    scene.netLight.light_params = None  # non-leaf cache of the last forward; not deep-copyable
    tex, dino, lgt = (copy.deepcopy(m).cpu() for m in (scene.netTexture, scene.netDINO, scene.netLight))

    def render(posed):
        with torch.no_grad():
            return render_ref.render_mesh(posed, faces, mesh_ref.vertex_normals(posed, faces), cpu(scene.mvp), cpu(scene.w2c), cpu(scene.campos),
                                          tex, lgt, scene.resolution, background=cpu(scene.background), feat=cpu(scene.feat),
                                          render_modes=("shaded", "dino_pred"), prior_v_pos=verts[None], dino_net=dino)

    # stage-wise parity: the renderer is checked on the SAME posed vertices the HIP renderer saw (the skinning stage has its own
    # figure above).  Even so the clip-space transform is a GPU matmul on one side and a CPU matmul on the other: a last-bit
    # difference there can hand a pixel on a shared edge or on the silhouette to another triangle (or to the background), which
    # says nothing about the kernels -- such pixels (from the two id buffers) and their antialiasing neighbours are excluded and
    # their fraction is reported; the rasteriser's own tests compare ids bit for bit on identical clip-space inputs.
    posed = cpu(shape.v_pos)
    shaded, dino_pred = render(posed)
    rast_o = raster_ref.rasterize(render_ref.xfm_points(posed, cpu(scene.mvp)).contiguous(), faces.int(), scene.resolution)
    flip = rast_o[..., 3] != cpu(scene.last["rast"])[..., 3]
    near = torch.nn.functional.max_pool2d(flip.float()[:, None], 3, 1, 1)[:, 0] > 0
    rep["frac_pixels_owner_flip"] = float(flip.float().mean())
    e1 = (shaded - cpu(out["shaded"])).abs() * (~near)[:, None]
    e2 = (dino_pred - cpu(out["dino_pred"])).abs() * (~near)[:, None]
    rep["max_abs_image_err"] = float(max(e1.max(), e2.max()))
    rep["frac_pixels_gt_1e-4"] = float(((e1.amax(1) > 1e-4) | (e2.amax(1) > 1e-4)).float().mean())
    rep["coverage"] = float((shaded[:, 3] > 0).float().mean())
    # end to end from the oracle's own skinning: informational (silhouette pixels may flip, see above)
    shaded_o, dino_o = render(sk)
    rep["frac_pixels_gt_1e-4_end_to_end"] = float((((shaded_o - cpu(out["shaded"])).abs().amax(1) > 1e-4)
                                                   | ((dino_o - cpu(out["dino_pred"])).abs().amax(1) > 1e-4)).float().mean())
    rep["loss"] = float(out["loss"])
    return rep
