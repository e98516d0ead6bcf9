"""Build libvlsat_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m vlsat_amd.build        (or: python cvpr2023-vlsat_amd/build.py)        the RELEASE library
    python cvpr2023-vlsat_amd/build.py --experiments      tools/bin/libvlsat_hip_exp.so: -DVLSAT_EXPERIMENTS -- timing-ablation
                                                          kernels and the lab switches of vlsat_debug_option (csrc/common.h);
                                                          `bench.py --lib tools/bin/libvlsat_hip_exp.so` and the probes load it

Objects are rebuilt only when a source or header is newer.  The .so is git-ignored but
travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libvlsat_hip.so")
SOURCES = ["engine_weights.hip", "engine_plan.hip", "engine_forward.hip", "engine_api.hip", "gemm_f32.hip", "gemm_bf16_ring.hip", "gemm_bf16_p8.hip", "gemm_splitk.hip", "flash_attn_f32.hip", "flash_attn_bf16.hip", "pointnet.hip", "pointnet_bf16.hip", "edge_gate.hip", "edge_gate_bf16.hip", "edge_gate_heads.hip", "edge_gate_bf16_heads.hip", "small_ops.hip", "stn.hip", "eval_ranks.hip", "prep.hip", "comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP library cannot be built")
    return exe


def _newest_header() -> float:
    t = 0.0
    for d in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


EXP_LIB = os.path.join(os.path.dirname(HERE), "tools", "bin", "libvlsat_hip_exp.so")


def build(force: bool = False, verbose: bool = False, experiments: bool = False) -> str:
    obj_dir = OBJ + "_exp" if experiments else OBJ
    lib = EXP_LIB if experiments else LIB
    flags = FLAGS + (["-DVLSAT_EXPERIMENTS"] if experiments else [])
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    hipcc = _hipcc()
    hdr_t = _newest_header()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            jobs.append([hipcc, *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for warn in ex.map(run, jobs):
            if warn and verbose:
                print(warn, file=sys.stderr)
    if jobs or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, experiments="--experiments" in sys.argv))
