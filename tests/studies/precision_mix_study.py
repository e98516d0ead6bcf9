#!/usr/bin/env python3
"""Which part of the single-rounding bf16 mode costs the accuracy?  (precision-by-depth mixes.)

A NUMERICAL EMULATION on the CPU, not a product path: the fp64 oracle with the operands of chosen matrix products rounded to
bf16 (and the edge tensors those kernels store rounded to bf16 too), i.e. what `bf16_mixed` does to the edge-row work, switched
on per LAYER and per kernel FAMILY.  Node-row products, the hoisted node parts of nn_edge.0 / the gate, LayerNorm, softmax and
all accumulation stay exact, like the split-bf16 / fp32 parts of the mode.  It answers "which mix would hold 1e-2 at x1.5" before
anything is built; the emulation is validated against the measured library errors of the all-single mix (6.95e-3 at x1.0,
1.53e-2 at x1.5, profiles/r04_probes/stress_scan.txt).

    python tests/studies/precision_mix_study.py [--scales 1,1.5] [--scenes 2]        (a few minutes on the host)
"""
import argparse
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vlsat_amd  # noqa: E402,F401
from vlsat_amd import VLSATConfig, synth  # noqa: E402
from oracle import vlsat_oracle as O  # noqa: E402

POLICY = {"layers": set(), "families": set(), "heads": False, "encoders": True}


def rb(x, on=True):
    return x.to(torch.bfloat16).to(x.dtype) if on else x


def layer_of(prefix):
    for tok in prefix.split("."):
        if tok.isdigit():
            return int(tok)
    return -1


def single(prefix, family):
    """is this product single-rounded under the current policy?"""
    return layer_of(prefix) in POLICY["layers"] and family in POLICY["families"]


def lin_r(x, w, name, on):
    weight = w[name + ".weight"]
    if weight.dim() == 3:
        weight = weight[:, :, 0]
    return F.linear(rb(x, on), rb(weight, on), w[name + ".bias"])


def edge_atten(x, e, ei, w, prefix, n_heads, taps=None):
    on = single(prefix, "gcn")
    xi, xj = O.gen_index(x, ei, "target_to_source")
    E = e.shape[0]
    p = prefix + ".edgeatten."
    D = x.shape[1]
    W0 = w[p + "nn_edge.0.weight"]
    # nn_edge.0: node parts hoisted to node rows (exact), edge part on the bf16 matrix cores; hidden stored as bf16
    h = F.linear(xi, W0[:, :D]) + F.linear(xj, W0[:, 2 * D:]) + F.linear(rb(e, on), rb(W0[:, D:2 * D], on)) + w[p + "nn_edge.0.bias"]
    h = rb(torch.relu(h), on)
    e_new = rb(lin_r(h, w, p + "nn_edge.2", on), on)                     # stored as bf16 (half rows)
    v = O.lin(xj, w, p + "proj_value.0")
    q = O.lin(xi, w, p + "proj_query.0")
    k = rb(lin_r(e, w, p + "proj_edge.0", on), on)
    q = q.view(E, q.shape[1] // n_heads, n_heads)
    k = k.view(E, k.shape[1] // n_heads, n_heads)
    w0, b0 = w[p + "nn.0.weight"][:, :, 0], w[p + "nn.0.bias"]
    w3, b3 = w[p + "nn.3.weight"][:, :, 0], w[p + "nn.3.bias"]
    dq = q.shape[1]
    z = torch.einsum("oc,ech->eoh", w0[:, :dq], q) + torch.einsum("oc,ech->eoh", rb(w0[:, dq:], on), rb(k, on)) + b0[None, :, None]
    z = rb(torch.relu(z), on)
    z = torch.einsum("oc,ech->eoh", rb(w3, on), z) + b3[None, :, None]
    prob = z.softmax(1)
    return prob.reshape(E, v.shape[1]) * v, e_new, prob


def mha(q_in, kv_in, w, prefix, n_heads, bias=None, q_chunk=1024):
    if "cross_attn_rel" not in prefix:
        return O_mha(q_in, kv_in, w, prefix, n_heads, bias, q_chunk)
    on = single(prefix, "attn")
    nq, nk = q_in.shape[0], kv_in.shape[0]
    if nq == 0:
        return q_in
    p = prefix + ".attention."
    dk = q_in.shape[1] // n_heads
    q = rb(lin_r(q_in, w, p + "fc_q", on), on).view(nq, n_heads, dk).permute(1, 0, 2)
    k = rb(lin_r(kv_in, w, p + "fc_k", on), on).view(nk, n_heads, dk).permute(1, 2, 0)
    v = rb(lin_r(kv_in, w, p + "fc_v", on), on).view(nk, n_heads, dk).permute(1, 0, 2)
    outs = []
    for s in range(0, nq, q_chunk):
        att = torch.softmax(torch.matmul(q[:, s:s + q_chunk], k) / math.sqrt(dk), -1)
        outs.append(torch.matmul(rb(att, on), v))
    o = rb(torch.cat(outs, 1).permute(1, 0, 2).reshape(nq, n_heads * dk), on)
    o = lin_r(o, w, p + "fc_o", on)
    y = F.layer_norm(q_in + o, (q_in.shape[1],), w[prefix + ".layer_norm.weight"], w[prefix + ".layer_norm.bias"], 1e-5)
    return rb(y, on)                                                     # E2 stored as bf16


def rel_head(e, w, prefix, multi=True):
    on = POLICY["heads"]
    h = rb(torch.relu(lin_r(e, w, prefix + ".fc1", on)), on)
    h = rb(torch.relu(lin_r(h, w, prefix + ".fc2", on)), on)
    h = lin_r(h, w, prefix + ".fc3", on)
    return torch.sigmoid(h) if multi else torch.log_softmax(h, dim=1)


O_mha = O.mha
O.mha, O.edge_atten, O.rel_head = mha, edge_atten, rel_head


def run(w64, cfg, scene):
    c = {k: torch.from_numpy(v) for k, v in synth.collate([scene]).items()}
    return O.forward(w64, cfg, c["obj_points"].double(), c["obj_2d_feats"].double(), c["edge_indices"], c["descriptor"].double(), c["batch_ids"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scales", default="1,1.5")
    ap.add_argument("--scenes", type=int, default=2)
    ap.add_argument("--objects", type=int, default=40)
    a = ap.parse_args()
    torch.set_num_threads(16)
    cfg = VLSATConfig(N_LAYERS=3)
    L = {0, 1, 2}
    mixes = [("exact (reference)", set(), set(), False),
             ("all single = bf16_mixed", L, {"gcn", "attn"}, True),
             ("heads split, rest single", L, {"gcn", "attn"}, False),
             ("layer 0 split", {1, 2}, {"gcn", "attn"}, True),
             ("layer 2 split", {0, 1}, {"gcn", "attn"}, True),
             ("layers 1, 2 split", {0}, {"gcn", "attn"}, True),
             ("layers 0, 1 split", {2}, {"gcn", "attn"}, True),
             ("edge attention split, gcn single", L, {"gcn"}, True),
             ("gcn split, edge attention single", L, {"attn"}, True),
             ("only heads single", set(), set(), True),
             ("heads + layers 1, 2 split (layer 0 single)", {0}, {"gcn", "attn"}, False),
             ("heads + gcn split (edge attention single)", L, {"attn"}, False),
             ("heads + edge attention split (gcn single)", L, {"gcn"}, False)]
    for sc in [float(x) for x in a.scales.split(",")]:
        w = synth.make_weights_stress(cfg, sc) if sc != 1.0 else synth.make_weights_stress(cfg, 1.0)
        w64 = O.to_torch(w, torch.float64)
        scenes = [synth.make_scene(a.objects, 256, 1000 + 21 * s) for s in range(a.scenes)]
        POLICY.update(layers=set(), families=set(), heads=False)
        refs = [run(w64, cfg, s) for s in scenes]
        print(f"stress weights x{sc} ({a.scenes} scenes of {a.objects} objects; max-abs-err of obj3d / obj2d / rel3d / rel2d vs exact):")
        for name, layers, fams, heads in mixes[1:]:
            POLICY.update(layers=layers, families=fams, heads=heads)
            worst = [0.0] * 4
            for s, ref in zip(scenes, refs):
                got = run(w64, cfg, s)
                worst = [max(x, float((g - r).abs().max())) for x, g, r in zip(worst, got, ref)]
            print(f"  {name:46s} " + "  ".join(f"{x:.2e}" for x in worst) + f"   worst {max(worst):.2e}", flush=True)


if __name__ == "__main__":
    main()
