"""Input-preparation oracle against the reference's gen_descriptor golden.  CPU only."""
import os

import numpy as np
import torch

import vlsat_amd  # noqa: F401
from vlsat_amd import synth
from oracle import prep_oracle as PO


def test_descriptor_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "prep_small.npz"))
    _, d32 = PO.prepare_objects(z["scene"], z["choice"], torch.float32)
    assert np.array_equal(d32.numpy(), z["desc_f32"])
    _, d64 = PO.prepare_objects(z["scene"], z["choice"], torch.float64)
    assert np.allclose(d64.numpy(), z["desc_f64"], rtol=1e-7, atol=0)


def test_synth_generator_uses_the_same_semantics():
    """The synthetic scene generator (product side) and the oracle agree on descriptor / zero-mean / edges."""
    g = np.random.default_rng(3)
    raw = g.uniform(-1, 1, (5, 40, 3)).astype(np.float32)
    scene = raw.reshape(-1, 3)
    choice = np.arange(200, dtype=np.int32).reshape(5, 40)
    obj, desc = PO.prepare_objects(scene, choice, torch.float64)
    assert np.allclose(desc.numpy(), synth.gen_descriptor(raw), rtol=1e-6, atol=1e-7)
    e, bid = PO.fc_edges_batch([3, 4])
    want = synth.collate([dict(obj_points=np.zeros((3, 3, 1), np.float32), obj_2d_feats=np.zeros((3, 1), np.float32),
                               edge_indices=synth.fc_edges(3), descriptor=np.zeros((3, 11), np.float32)),
                          dict(obj_points=np.zeros((4, 3, 1), np.float32), obj_2d_feats=np.zeros((4, 1), np.float32),
                               edge_indices=synth.fc_edges(4), descriptor=np.zeros((4, 11), np.float32))])
    assert np.array_equal(e.t().numpy(), want["edge_indices"]) and np.array_equal(bid.numpy(), want["batch_ids"])
