"""Build libgaot_hip.so (gfx950) in-tree with hipcc.  `python -m gaot_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgaot_hip.so")
SOURCES = ["capi.cpp", "gemm.hip", "gemm_glds.hip", "gemm_split.hip", "gemm_ad.hip", "skinny.hip", "gno.hip", "gno_ep.hip", "edge_drop.hip", "kernel_mlp.hip", "radius.hip", "pointwise.hip", "glue.hip", "attention.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "segsort.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(HERE, "..", "include", "gaot_hip.h"), os.path.join(HERE, "..", "include", "gaot_hip_debug.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-array-bounds"] + os.environ.get("GAOT_HIPCC_EXTRA", "").split()      # (extra: A/B builds, tools)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.rsplit(".", 1)[0] + ".o")
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append((s, cmd))

    def run(job):
        name, cmd = job
        if verbose:
            print("[gaot_amd.build] hipcc", name, flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}:\n{r.stdout}\n{r.stderr}")

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(obj_dir, s.rsplit(".", 1)[0] + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print("[gaot_amd.build] linked", LIB_PATH, flush=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
