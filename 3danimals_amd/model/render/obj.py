"""OBJ export -- `write_obj` of /root/reference/model/render/obj.py:128-177 (output side of the test_* configs).

Same file, line for line (`v x y z `, `vt u 1-v `, `vn x y z`, `f a/b/c ...`, numpy-float text formatting), written with
joined buffers instead of one `f.write` per element.  The material (`.mtl` + baked textures through `render_uv`) is
written by the reference's own `model/render/material.py` when this package is overlaid on the reference tree and by `save_mtl` below
otherwise; OBJ import (`load_obj`) is outside the reconstruct-and-render path and not provided.
"""
import os

_STANDALONE_ONLY = ("load_obj",)  # placeholders for stand-alone use: never overlaid on the reference's real definitions


def write_obj(folder, fname, mesh, idx, save_material=True, feat=None, resolution=[256, 256]):
    obj_file = os.path.join(folder, fname + ".obj")
    print("Writing mesh: ", obj_file)
    npy = lambda t: None if t is None else t.detach().cpu().numpy()
    v_pos = npy(mesh.v_pos[idx]) if mesh.v_pos is not None else None
    v_nrm = npy(mesh.v_nrm[idx]) if mesh.v_nrm is not None else None
    v_tex = npy(mesh.v_tex[idx]) if mesh.v_tex is not None else None
    t_pos_idx = npy(mesh.t_pos_idx[0]) if mesh.t_pos_idx is not None else None
    t_nrm_idx = npy(mesh.t_nrm_idx[0]) if mesh.t_nrm_idx is not None else None
    t_tex_idx = npy(mesh.t_tex_idx[0]) if mesh.t_tex_idx is not None else None
    with open(obj_file, "w") as f:
        f.write(f"mtllib {fname}.mtl\n")
        f.write("g default\n")
        print("    writing %d vertices" % len(v_pos))
        f.write("".join("v {} {} {} \n".format(v[0], v[1], v[2]) for v in v_pos))
        if v_tex is not None and save_material:
            print("    writing %d texcoords" % len(v_tex))
            assert len(t_pos_idx) == len(t_tex_idx)
            f.write("".join("vt {} {} \n".format(v[0], 1.0 - v[1]) for v in v_tex))
        if v_nrm is not None:
            print("    writing %d normals" % len(v_nrm))
            assert len(t_pos_idx) == len(t_nrm_idx)
            f.write("".join("vn {} {} {}\n".format(v[0], v[1], v[2]) for v in v_nrm))
        f.write("s 1 \n")
        f.write("g pMesh1\n")
        f.write("usemtl defaultMat\n")
        print("    writing %d faces" % len(t_pos_idx))
        lines = []
        for i in range(len(t_pos_idx)):
            parts = " ".join("%s/%s/%s" % (str(t_pos_idx[i][j] + 1), "" if v_tex is None else str(t_tex_idx[i][j] + 1),
                                           "" if v_nrm is None else str(t_nrm_idx[i][j] + 1)) for j in range(3))
            lines.append("f  " + parts + "\n")
        f.write("".join(lines))
    if save_material and mesh.material is not None:
        mtl_file = os.path.join(folder, fname + ".mtl")
        print("Writing material: ", mtl_file)
        try:
            from model.render import material  # the reference's (unchanged) material writer, when overlaid on its tree
        except ImportError:
            save_mtl(mtl_file, mesh.material, mesh=mesh.get_n(idx), feat=feat, resolution=resolution)
        else:
            material.save_mtl(mtl_file, mesh.material, mesh=mesh.get_n(idx), feat=feat, resolution=resolution)
    print("Done exporting mesh")


def save_mtl(fn, material, mesh=None, feat=None, resolution=[256, 256]):
    """The .mtl + baked texture maps of /root/reference/model/render/material.py:106-140 for the material kind the models on this path
    produce: a dict / ModuleDict with 'bsdf' and the texture field under 'kd_ks_normal', baked in uv space by ``render_uv``
    (render.py:342-360) into <prefix>texture_kd.png / _ks.png / _n.png (8-bit, values * 255 truncated, like misc.save_images).
    Stand-alone counterpart: overlaid on the reference tree, write_obj hands over to the reference's own save_mtl."""
    import numpy as np
    import torch
    from PIL import Image

    from . import render

    folder, name = os.path.dirname(fn), os.path.basename(fn)
    prefix = "_".join(name.split("_")[:-1]) + "_"
    with open(fn, "w") as f:
        f.write("newmtl defaultMat\n")
        if material is None:
            f.write("Kd 1 1 1\nKs 0 0 0\nKa 0 0 0\nTf 1 1 1\nNi 1\nNs 0\n")
            return
        f.write("bsdf   %s\n" % material["bsdf"])
        if "kd_ks_normal" not in material.keys():
            raise NotImplementedError("only MLP materials ('kd_ks_normal') are written stand-alone; Texture2D materials need the reference's texture.py")
        assert mesh is not None
        with torch.no_grad():
            _, kd, ks, normal = render.render_uv(None, mesh, resolution, material["kd_ks_normal"], feat=feat)
        for tag, key, img in (("map_Kd", "texture_kd", kd), ("map_Ks", "texture_ks", ks), ("bump", "texture_n", normal)):
            f.write(f"{tag} {prefix}{key}.png\n")
            for frame in img.detach().cpu().numpy():  # every batch entry goes to the same file name, as in the reference
                Image.fromarray(np.uint8(np.clip(frame, 0.0, 1.0) * 255.0)).save(os.path.join(folder, prefix + key + ".png"))


def load_obj(*args, **kwargs):
    raise NotImplementedError("OBJ import is outside the reconstruct-and-render hot path")
