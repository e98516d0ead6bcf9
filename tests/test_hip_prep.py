"""GPU input-preparation kernels (csrc/prep.hip) against the reference golden and the oracle."""
import os

import numpy as np
import pytest
import torch

import vlsat_amd  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_prepare_objects_and_edges(golden_dir):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path cannot run and there is no fallback")
    from vlsat_amd import prep
    from oracle import prep_oracle as PO
    z = np.load(os.path.join(golden_dir, "prep_small.npz"))
    pts, desc = prep.prepare_objects(torch.from_numpy(z["scene"]).to(DEV), torch.from_numpy(z["choice"]).to(DEV))
    torch.cuda.synchronize()
    assert np.allclose(desc.cpu().numpy(), z["desc_f64"], rtol=2e-6, atol=1e-6)          # reference gen_descriptor (fp64 input)
    assert np.allclose(desc.cpu().numpy(), z["desc_f32"], rtol=2e-6, atol=1e-6)
    ref_pts, _ = PO.prepare_objects(z["scene"], z["choice"])
    assert float((pts.cpu() - ref_pts).abs().max()) < 2e-6
    # bigger, ragged point count, several waves per object
    g = np.random.default_rng(1)
    scene = g.normal(size=(5000, 3)).astype(np.float32) * np.array([2, 1, 0.5], np.float32) + 3
    choice = g.integers(0, 5000, (37, 777)).astype(np.int32)
    pts, desc = prep.prepare_objects(torch.from_numpy(scene).to(DEV), torch.from_numpy(choice).to(DEV))
    ref_pts, ref_desc = PO.prepare_objects(scene, choice, torch.float64)
    assert np.allclose(desc.cpu().numpy(), ref_desc.numpy(), rtol=5e-6, atol=2e-6)
    assert float((pts.cpu() - ref_pts).abs().max()) < 5e-6
    for ns in ([3, 4], [1, 5, 2], [40] * 7):
        e, bid = prep.fc_edges(ns, DEV)
        re, rb = PO.fc_edges_batch(ns)
        assert torch.equal(e.cpu(), re.t().contiguous()) and torch.equal(bid.cpu(), rb)
