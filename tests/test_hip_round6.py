"""Round 6 on the device: switches added this round must not change results where they are pure re-orderings of memory traffic, and
must stay inside the parity tolerance where they re-order a floating-point sum.  Needs an MI355X."""
import numpy as np
import pytest
import torch

import vlsat_amd  # noqa: F401
from vlsat_amd import VLSATConfig, synth
from vlsat_amd.model import VLSATModel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ("obj3d", "obj2d", "rel3d", "rel2d")


def _batch(n_scenes, n_obj, n_pts, seed0):
    b = synth.make_batch(n_scenes, n_obj, n_pts, seed0=seed0)
    return b, {k: torch.from_numpy(v).to(DEV) for k, v in b.items()}


def _run(m, d):
    out = m(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
    torch.cuda.synchronize()
    return [o.clone() for o in out]


@pytest.mark.parametrize("shape", [(3, 40, 64), (1, 70, 32)])     # 128-query tiles; one scene of 4830 edges: the 256-query tiles
def test_flash_asm_transpose_reads_are_bit_identical(shape):
    """"flash_asmv": the bf16 edge attention reads its V fragments with inline-asm ds_read_b64_tr_b16 and counted waits instead of the
    builtin (which makes hipcc drain the LDS-direct loads of the next tile in front of the P.V product).  Same instructions on the
    same data: outputs must be bit-identical, in both tile sizes of the LDS-direct kernel."""
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path cannot run and there is no fallback")
    cfg = VLSATConfig(N_LAYERS=2)
    w = synth.make_weights(cfg)
    _, d = _batch(*shape, seed0=4100)
    ref = VLSATModel(cfg, DEV).load_state(w).eval().set_gemm_precision("bf16_mixed").debug_option("flash_asmv", 0)
    asm = VLSATModel(cfg, DEV).load_state(w).eval().set_gemm_precision("bf16_mixed").debug_option("flash_asmv", 1)
    a, b = _run(ref, d), _run(asm, d)
    for n, x, y in zip(NAMES, a, b):
        assert torch.isfinite(x).all() and torch.equal(x, y), n
    ref.close(); asm.close()


@pytest.mark.parametrize("precision,tol", [("bf16_mixed", 1e-2), ("bf16x3", 1e-3), ("fp32", 1e-3)])
def test_k_tile_rotation_of_the_8_phase_gemm_stays_inside_the_tolerance(precision, tol):
    """"gemm_k_rot" r: column tile tn of a row panel walks its K-tiles starting at tn * r (the blocks that share an A panel then ask
    L2 for the same lines a K-tile apart instead of in the same microsecond).  A rotation of the fp32 summation order: not
    bit-identical, but every output stays inside the mode's tolerance against the CPU oracle, on a batch large enough for the
    8-phase kernel to take the edge-row launches (E >= 65536)."""
    from oracle import vlsat_oracle as O
    cfg = VLSATConfig(N_LAYERS=1)
    w = synth.make_weights(cfg)
    b, d = _batch(44, 40, 32, seed0=4200)                          # E = 68 640
    c = {k: torch.from_numpy(v) for k, v in b.items()}
    ref = O.forward(O.to_torch(w), cfg, c["obj_points"], c["obj_2d_feats"], c["edge_indices"], c["descriptor"], c["batch_ids"])
    outs = {}
    for r in (0, 1, 3):
        m = VLSATModel(cfg, DEV).load_state(w).eval().set_gemm_precision(precision).debug_option("gemm_k_rot", r)
        outs[r] = _run(m, d)
        m.close()
        for n, g, x in zip(NAMES, outs[r], ref):
            assert float((g.cpu() - x).abs().max()) < tol, (precision, r, n)
    if precision != "fp32":
        assert any(not torch.equal(x, y) for x, y in zip(outs[0], outs[1]))        # (the switch reaches the kernel)
