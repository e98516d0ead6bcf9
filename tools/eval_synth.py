#!/usr/bin/env python3
"""End-to-end synthetic evaluation: the MI355X counterpart of `python main.py --mode eval`
(reference main.py:41-42 -> MMGNet.validation, src/model/model.py:181-362) with every stage on the GPU:

  raw scene points -> prep.prepare_objects / prep.fc_edges      (data loader, dataset_3dssg.py:264-294)
                   -> VLSATModel.forward                        (Mmgnet.forward, SGFN_MMG/model.py:288-335)
                   -> metrics.eval_ranks                        (process_val ranking, :463-472)
                   -> evaluate: counts vector, ONE all-reduce   (validation() summaries, model.py:214-282)

    python tools/eval_synth.py [--scenes 256] [--batch 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/eval_synth.py

Data, labels and weights are synthetic (there is no dataset or checkpoint in this environment), so
the accuracies are chance level; what it shows is the pipeline, its sharding invariance and its speed."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402
from vlsat_amd import VLSATConfig, synth, prep, evaluate as EV, dist as vdist  # noqa: E402
from vlsat_amd.model import VLSATModel  # noqa: E402


def scene_batch(scene_ids, n_obj, n_pts, dev):
    """Raw synthetic scenes -> the loader's item dict, prepared ON THE DEVICE."""
    raws, f2d, gts, rels = [], [], [], []
    for s in scene_ids:
        g = np.random.default_rng([s, 77])
        centre = g.uniform(0, 4, (n_obj, 1, 3))
        ext = g.uniform(0.2, 1.0, (n_obj, 1, 3))
        raws.append((centre + g.uniform(-0.5, 0.5, (n_obj, 2 * n_pts, 3)) * ext).astype(np.float32))   # 2x points to sample from
        f = g.standard_normal((n_obj, 512))
        f2d.append((f / np.linalg.norm(f, axis=-1, keepdims=True)).astype(np.float32))
        gts.append(g.integers(0, 160, n_obj))
        rels.append((g.random((n_obj * (n_obj - 1), 26)) < 0.04).astype(np.int64))
    raw = np.concatenate(raws, 0)                                     # [N, 2P, 3]
    n = raw.shape[0]
    g = np.random.default_rng([scene_ids[0], 99])
    choice = g.integers(0, 2 * n_pts, (n, n_pts)) + (np.arange(n) * 2 * n_pts)[:, None]     # np.random.choice(..., replace=True)
    pts, desc = prep.prepare_objects(torch.from_numpy(raw.reshape(-1, 3)).to(dev), torch.from_numpy(choice.astype(np.int32)).to(dev))
    edges, bids = prep.fc_edges([n_obj] * len(scene_ids), dev)
    return {"obj_points": pts, "descriptor": desc, "obj_2d_feats": torch.from_numpy(np.concatenate(f2d)).to(dev),
            "gt_class": torch.from_numpy(np.concatenate(gts)).to(dev), "gt_rel_cls": torch.from_numpy(np.concatenate(rels)).to(dev),
            "edge_indices": edges.t().contiguous(), "batch_ids": bids}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--objects", type=int, default=40)
    ap.add_argument("--points", type=int, default=256)
    ap.add_argument("--layers", type=int, default=3)
    a = ap.parse_args()
    rank, local, world = vdist.init()
    local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = VLSATConfig(N_LAYERS=a.layers)
    model = VLSATModel(cfg, str(dev)).load_state(synth.make_weights(cfg)).eval()
    mine = list(vdist.shard(a.scenes, rank, world))
    batches = [scene_batch(mine[i:i + a.batch], a.objects, a.points, dev) for i in range(0, len(mine), a.batch)]
    EV.validation(model, batches[:1], device=None)                    # warm-up (plans, allocator) without the collective
    torch.cuda.synchronize()
    vdist.barrier()
    t0 = time.perf_counter()
    summary = EV.validation(model, batches, device=dev)
    torch.cuda.synchronize()
    dt = vdist.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        print(f"evaluated {int(summary['scenes'])} scenes on {world} GPU(s) in {dt:.3f} s = {summary['scenes'] / dt:.1f} scenes/s "
              f"(forward + ranking + counts; reference CPU ranking alone: ~116 s/scene)")
        for k in ("obj_acc@1_3d", "obj_acc@10_3d", "rel_acc@1_3d", "rel_acc@5_3d", "tri_acc@50_3d", "tri_acc@100_3d",
                  "mean_recall@50_3d", "mean_recall@100_3d", "mean_rel_acc@1_3d", "obj_acc@1_2d", "tri_acc@100_2d"):
            print(f"  {k:22s} {summary[k]:7.3f}")


if __name__ == "__main__":
    main()
