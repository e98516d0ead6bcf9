#!/usr/bin/env python3
"""Per-block view of one persistent fp32 GEMM launch: every block reports its shader cycles, wall time, tiles done and
where it ran (XCC, SE, CU).  Shows how evenly the two blocks that share a CU split its matrix pipes.
    python tools/gemm_block_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402
from vlsat_amd import lib as L  # noqa: E402

lib = L.load()
dev = "cuda:0"
buf = torch.zeros(4 * 1024, dtype=torch.int64, device=dev)
for (M, N, K) in ((98304, 512, 512), (98304, 512, 1024)):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    Cb = torch.empty(M, N, device=dev)

    def run():
        L.check(lib.vlsat_k_gemm(A.data_ptr(), K, W.data_ptr(), K, Cb.data_ptr(), N, M, N, K, 0, 0, 0, 0, 1.0,
                                 0, 0, 0, 0, 0, 0, 0, 0, L.stream_ptr()))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    buf.zero_()
    L.check(lib.vlsat_debug_gemm_clock_probe(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    L.check(lib.vlsat_debug_gemm_clock_probe(None))
    raw = buf.view(-1, 4).cpu()
    ids = torch.arange(raw.shape[0])[raw[:, 3] == 1]
    b = raw[raw[:, 3] == 1]
    cyc, wall = b[:, 0].double(), b[:, 1].double() / 100.0
    tiles = b[:, 2] & 0xffff
    hw = (b[:, 2] >> 16) & 0xffff
    xcc = (b[:, 2] >> 32) & 0xf
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 0x1, (hw >> 13) & 0x7
    q = torch.quantile(cyc, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64))
    print(f"M{M} N{N} K{K}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us; {len(b)} blocks, tiles/block {int(tiles.min())}..{int(tiles.max())}; "
          f"block cycles min/p10/p50/p90/max = {[int(x) for x in q]}; wall us min/med/max = {wall.min():.1f}/{wall.median():.1f}/{wall.max():.1f}")
    # group by physical CU
    key = (xcc * 8 + se) * 32 + sh * 16 + cu
    groups = {}
    for i in range(len(b)):
        groups.setdefault(int(key[i]), []).append(i)
    sizes = sorted(len(v) for v in groups.values())
    print(f"   distinct (xcc,se,sh,cu) = {len(groups)}; blocks per CU: min {sizes[0]} max {sizes[-1]}")
    pair = [(float(cyc[v[0]]), float(cyc[v[1]])) for v in groups.values() if len(v) == 2]
    if pair:
        lo = torch.tensor([min(p) for p in pair]); hi = torch.tensor([max(p) for p in pair])
        print(f"   pairs on one CU: faster block {lo.mean():.0f} cycles (min {lo.min():.0f} max {lo.max():.0f}), slower {hi.mean():.0f} (min {hi.min():.0f} max {hi.max():.0f})")
    per_x = [float(wall[xcc == x].mean()) for x in range(8) if (xcc == x).any()]
    print("   mean block wall time per XCC (us): " + " ".join(f"{v:.0f}" for v in per_x))
    first = ids < len(b) // 2
    print(f"   blocks with id < grid/2: {float(wall[first].mean()):.1f} us mean; id >= grid/2: {float(wall[~first].mean()):.1f} us")
