#!/usr/bin/env python3
"""Which tile of gemm_f32_kernel the node-row GEMMs (M = number of nodes of the batch, 2560 at the bench batch) should take:
every shape the forward launches on node rows x {fp32, split-bf16} x {heuristic, heuristic + split-K, 128x128, 128x64, 64x128,
64x64}, through the C ABI (GemmArgs::force_tile).  Prints microseconds per launch (median of 5 x `iters` launches).
    python tools/gemm_tile_sweep.py [--rows 2560] [--iters 40]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402
from vlsat_amd import lib as L  # noqa: E402

SHAPES = [("self qkv", 1536, 512, False), ("cross kv", 1024, 512, False), ("q / out-proj(+resid)", 512, 512, True),
          ("wnode", 3328, 512, False), ("prop.0", 768, 768, False), ("prop.2", 512, 768, False), ("mlp_3d", 504, 768, False),
          ("adapter fc1", 256, 512, False), ("obj head", 160, 512, False)]
TILES = [("auto", 0, False), ("auto+splitk", 0, True), ("128x128", 1, False), ("128x64", 2, False), ("64x128", 3, False), ("64x64", 4, False),
         ("64x64 x4/CU", 5, False), ("64x128 x3/CU", 6, False), ("64x64 x3/CU", 7, False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2560)
    ap.add_argument("--iters", type=int, default=40)
    a = ap.parse_args()
    lib = L.load()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    M = a.rows
    print(f"M = {M}; microseconds per launch")
    print(f"{'shape':28s} {'prec':6s} " + " ".join(f"{t[0]:>12s}" for t in TILES))
    for name, N, K, resid in SHAPES:
        A = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        Cb = torch.empty(M, N, device=dev)
        bias = torch.randn(N, generator=g).to(dev)
        R = torch.randn(M, N, device=dev) if resid else None
        hi = torch.empty(N * K + 128, dtype=torch.int16, device=dev)
        lo = torch.empty_like(hi)
        L.check(lib.vlsat_k_split_bf16(W.data_ptr(), N * K, hi.data_ptr(), lo.data_ptr(), L.stream_ptr()))
        for prec in (0, 3):
            row = []
            for _, tile, sk in TILES:
                def fn():
                    if prec == 0:
                        L.check(lib.vlsat_k_gemm(A.data_ptr(), K, W.data_ptr(), K, Cb.data_ptr(), N, M, N, K, bias.data_ptr(), 0,
                                                 L.ptr(R), N if resid else 0, 1.0, 0, 0, 0, 0, 0, 0, (4 if sk else 0) | (tile << 4), 0, L.stream_ptr()))
                    else:
                        L.check(lib.vlsat_k_gemm_planes(A.data_ptr(), K, W.data_ptr(), hi.data_ptr(), lo.data_ptr(), K, Cb.data_ptr(), N, M, N, K,
                                                        bias.data_ptr(), L.ptr(R), N if resid else 0, 1.0, 0, 0, 0, 0, 0, 0,
                                                        0, 0, 3, 0, -1, (64 if sk else 0) | (tile << 19), 1.0, L.stream_ptr()))
                try:
                    for _ in range(3):
                        fn()
                except L.VlsatError:
                    row.append(float("nan"))
                    continue
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.iters):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / a.iters * 1e3)
                row.append(sorted(ts)[2])
            print(f"{name + f' N={N} K={K}':28s} {'fp32' if prec == 0 else 'bf16x3':6s} " + " ".join(f"{x:12.1f}" for x in row), flush=True)


if __name__ == "__main__":
    main()
