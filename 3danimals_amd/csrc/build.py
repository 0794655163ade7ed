"""Build liba3d_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python 3danimals_amd/csrc/build.py [--force] [--profile | --exp]

One object per .hip file, linked into 3danimals_amd/lib/liba3d_hip.so (in-tree, git-ignored, travels to the
GPU box with the snapshot).  raster/dmtet/antialias/normals are compiled with -ffp-contract=off: their arithmetic is
specified operation by operation (oracle/raster_ref.c, reference dmtet.py:124-131, mesh.py:276-304).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "liba3d_hip.so")
OBJ_DIR = os.path.join(HERE, "build")

SOURCES = {
    "common.hip": [],
    "dmtet.hip": ["-ffp-contract=off"],
    "skin.hip": [],
    "bones.hip": [],
    "normals.hip": ["-ffp-contract=off"],
    "raster.hip": ["-ffp-contract=off"],
    "cover.hip": [],
    "shade.hip": ["-ffp-contract=off"],
    "interp.hip": [],
    "gbuffer.hip": [],
    "segsum.hip": [],
    "losses.hip": [],
    "embed.hip": ["-ffp-contract=off"],
    "gemm.hip": [],
    "antialias.hip": ["-ffp-contract=off"],
    "topology.hip": [],
    "xfm.hip": ["-ffp-contract=off"],
}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall",
          "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, profile=False, exp=False):
    """profile: the instrumented twin liba3d_hip_prof.so (-DA3D_PROFILE: phase stamps inside the kernels, a3d_common.h) that
    tools/kernel_phases.py loads instead of the library; never loaded by the package on its own.
    exp: liba3d_hip_exp.so (-DA3D_EXPERIMENT): the same kernels with the A3D_EXP measurement knobs live (a3d_exp() reads the environment
    only there); loaded through A3D_LIB by the tools under tools/, never by the package on its own."""
    assert not (profile and exp)
    lib = os.path.join(LIB_DIR, "liba3d_hip_prof.so") if profile else (os.path.join(LIB_DIR, "liba3d_hip_exp.so") if exp else LIB)
    obj_dir = os.path.join(OBJ_DIR, "prof") if profile else (os.path.join(OBJ_DIR, "exp") if exp else OBJ_DIR)
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    # every header of csrc/ (a kernel's object is stale when ANY of them is newer: a header left out of a hand-kept list once kept a
    # stale interp.o in the library), the public header and this script
    headers = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")) + [os.path.join(os.path.dirname(PKG), "include", "a3d.h"), os.path.abspath(__file__)]
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(HERE, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, "-x", "hip", "-c", s, "-o", o] + COMMON + extra + (["-DA3D_PROFILE"] if profile else []) + (["-DA3D_EXPERIMENT"] if exp else []))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, profile="--profile" in sys.argv, exp="--exp" in sys.argv))
