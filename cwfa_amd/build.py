"""Build libcwfa_hip.so (gfx950) in-tree with hipcc.  No JIT cache, no torch extension machinery:
the .so sits next to this file so that it travels with the source tree and is seen by the loader."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libcwfa_hip.so")
SOURCES = ["elementwise.hip", "conv2d.hip", "conv_wino.hip", "conv_wino2d.hip", "conv_split_layer.hip", "conv_split3x3.hip", "conv3d.hip", "conv3d_split.hip", "lrnn_ops.hip", "conv_bwd.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I", INCLUDE, "-I", CSRC,
         "-Wall", "-Wno-unused-function"]


# per-file flags.  conv3d_split.hip: paired into v_pk_add_f32 the residual subtractions of its three-way split are slow beside MFMAs
EXTRA = {"conv3d_split.hip": ["-fno-slp-vectorize"]}


def _newer(src, dst):
    return not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst)


def build_all(force=False, verbose=False):
    deps = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_internal.h"), os.path.join(INCLUDE, "cwfa_hip.h"),
            os.path.abspath(__file__)]
    objs, jobs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(d, obj) for d in deps):
            jobs.append([HIPCC, *FLAGS, *EXTRA.get(s, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn, file=sys.stderr)
    if jobs or force or not os.path.exists(LIB) or any(_newer(o, LIB) for o in objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
