#!/usr/bin/env python3
"""Golden vectors for the per-object input preparation: the reference's
gen_descriptor (src/utils/op_utils.py:47-64) called on sampled object points, as
data_preparation does (src/dataset/dataset_3dssg.py:289-293).  The dataset class itself (data_preparation, zero_mean,
collate_fn_mmg, the readers) is run by tests/golden/make_golden_scan.py -> scan_small.npz."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (stand-ins for torch_geometric, needed to import op_utils)

G.install_standins()
from src.utils import op_utils  # noqa: E402

g = np.random.default_rng(11)
scene = (g.uniform(-3, 3, (600, 3)) * np.array([1.0, 0.6, 0.3])).astype(np.float32)
inst = g.integers(0, 6, 600)
P = 50
choice = np.stack([g.choice(np.where(inst == i)[0], P, replace=True) for i in range(6)]).astype(np.int32)
desc32 = torch.stack([op_utils.gen_descriptor(torch.from_numpy(scene[choice[i]])) for i in range(6)])
desc64 = torch.stack([op_utils.gen_descriptor(torch.from_numpy(scene[choice[i]].astype(np.float64))) for i in range(6)])
np.savez_compressed(os.path.join(HERE, "prep_small.npz"), scene=scene, choice=choice, desc_f32=desc32.numpy(),
                    desc_f64=desc64.numpy().astype(np.float64))
print("written", desc32.shape)
