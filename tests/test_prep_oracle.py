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


def test_sample_choice_restatement_draws_from_the_right_points_uniformly():
    """oracle.prep_oracle.sample_choice (the documented generator of vlsat_sample_objects over np.where's index lists, reference
    dataset_3dssg.py:285-289): every drawn index belongs to its instance, draws are a function of (seed, object, draw) only, and
    the per-point histogram of 200 000 draws from a 50-point instance is uniform (chi-square within 4 sigma of its mean)."""
    import numpy as np
    from oracle import prep_oracle as P
    g = np.random.default_rng(3)
    inst = g.integers(0, 12, 5000).astype(np.int32)
    inst[g.integers(0, 5000, 40)] = 77                                    # a small instance
    ids = [3, 77, 5, 99]                                                   # 99 does not occur
    ch, cnt = P.sample_choice(inst, ids, 64, seed=1234)
    assert cnt.tolist() == [int((inst == i).sum()) for i in ids] and cnt[3] == 0
    for o, iid in enumerate(ids[:3]):
        assert (inst[ch[o]] == iid).all()
    ch2, _ = P.sample_choice(inst, ids, 64, seed=1234)
    assert (ch == ch2).all() and not (ch == P.sample_choice(inst, ids, 64, seed=1235)[0]).all()
    one = np.full(50, 7, dtype=np.int32)
    big, _ = P.sample_choice(one, [7], 200000, seed=9)
    h = np.bincount(big[0], minlength=50)
    chi2 = ((h - 4000.0) ** 2 / 4000.0).sum()                              # 49 degrees of freedom: mean 49, sigma ~9.9
    assert abs(chi2 - 49) < 4 * np.sqrt(2 * 49), chi2
