"""The C ABI used the way a compiled-language host would: examples/c_abi_demo.cpp (no Python, no PyTorch in the
process -- only libvlsat_hip.so and the HIP runtime) runs BASELINE configs[0] and is checked against the golden
outputs of the real reference.  Needs an MI355X and g++."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import vlsat_amd  # noqa: F401
from vlsat_amd import VLSATConfig, synth, build as B

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_demo_matches_reference_golden(tmp_path, golden_dir):
    gxx = shutil.which("g++")
    assert gxx, "g++ not found"
    lib_dir = os.path.dirname(B.LIB)
    assert os.path.exists(B.LIB), "libvlsat_hip.so is not built"
    exe = str(tmp_path / "c_abi_demo")
    subprocess.run([gxx, "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "examples", "c_abi_demo.cpp"), "-L", lib_dir, "-lvlsat_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{lib_dir}:/opt/rocm/lib", "-o", exe], check=True)
    cfg = VLSATConfig(N_LAYERS=2)
    w = synth.make_weights(cfg)
    b = synth.make_batch(1, 8, 256, seed0=1000)
    z = np.load(os.path.join(golden_dir, "cfg1_n8_p256_l2.npz"))
    d = tmp_path / "case"
    d.mkdir()
    n, e, p = b["obj_points"].shape[0], b["edge_indices"].shape[1], b["obj_points"].shape[2]
    (d / "meta.txt").write_text(f"{cfg.N_LAYERS} {n} {e} {p} {len(w)}\n")
    with open(d / "weights.bin", "wb") as f:
        for name, arr in w.items():
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<q", arr.size) + np.ascontiguousarray(arr, np.float32).tobytes())
    for fn, arr, dt in (("obj_points", b["obj_points"], np.float32), ("obj_2d_feats", b["obj_2d_feats"], np.float32),
                        ("descriptor", b["descriptor"], np.float32), ("edges", b["edge_indices"], np.int64),
                        ("batch_ids", b["batch_ids"].reshape(-1), np.int64)):
        np.ascontiguousarray(arr, dt).tofile(d / f"{fn}.bin")
    for k in ("obj3d", "obj2d", "rel3d", "rel2d"):
        np.ascontiguousarray(z[k], np.float32).tofile(d / f"expect_{k}.bin")
    r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.returncode, r.stdout, r.stderr)
