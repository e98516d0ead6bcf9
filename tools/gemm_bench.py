#!/usr/bin/env python3
"""Micro-benchmark of the MFMA GEMM through the C ABI on the shapes the cfg-2 forward launches.
    python tools/gemm_bench.py [--iters 20] [--only NAME] [--prec 0|1|3] [--no-dma] [--prefetch D[,D...]] [--rows M]
Prints one line per shape: time, TFLOP/s, fraction of the mode's MFMA peak (fp32 157.3 TF; bf16 2500 TF; bf16x3 833 TF)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402
from vlsat_amd import lib as L  # noqa: E402

E, N = 99840, 2560
SHAPES = [
    # name, M, N, K, resid, gather
    ("kproj/q/fc1 E x512 x512", E, 512, 512, False, False),
    ("out-proj +resid E x512 x512", E, 512, 512, True, False),
    ("kv E x1024 x512", E, 1024, 512, False, False),
    ("nn_edge.0 +gather E x1024 x512", E, 1024, 512, False, True),
    ("nn_edge.2 E x512 x1024", E, 512, 1024, False, False),
    ("rel conv3 E x512 x128", E, 512, 128, False, False),
    ("rel conv2 E x128 x64", E, 128, 64, False, False),
    ("fc2 E x256 x512", E, 256, 512, False, False),
    ("fc3 E x26 x256", E, 26, 256, False, False),
    ("node proj N x3328 x512", N, 3328, 512, False, False),
    ("node qkv N x1536 x512", N, 1536, 512, False, False),
    ("node 512", N, 512, 512, True, False),
    ("prop.0 N x768 x768", N, 768, 768, False, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--prec", type=int, default=0, choices=[0, 1, 3])
    ap.add_argument("--no-dma", action="store_true", help="bf16 modes: the VGPR-staged operand pipe of round 1")
    ap.add_argument("--prefetch", default="-1", help="bf16 LDS-direct pipe: A-prefetch look-ahead(s) in slices, comma separated")
    ap.add_argument("--no-prefetch", action="store_true", help="fp32: without the A-panel prefetch")
    ap.add_argument("--fmt", type=int, default=0, help="bit 0: A, bit 1: resid, bit 2: C in the split-pair format, bit 5: half rows instead (timing only)")
    ap.add_argument("--rows", type=int, default=0, help="override M of the edge-row shapes (e.g. 8192: operands stay in L2)")
    ap.add_argument("--no-p8", action="store_true", help="fp32: large launches stay off the 256 x 256 8-phase kernel")
    ap.add_argument("--splitk", action="store_true", help="small launches may take the split-K kernel (what the engine does)")
    ap.add_argument("--nodes", type=int, default=0, help="override M of the node-row shapes (one scene: 9..80)")
    a = ap.parse_args()
    peak = {0: 157.3, 1: 2500.0, 3: 2500.0 / 3}[a.prec]
    lib = L.load()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    for name, M, Nn, K, resid, gather in SHAPES:
        if a.only and a.only not in name:
            continue
        if a.rows and M == E:
            M = a.rows
        elif a.nodes and M == N:
            M = a.nodes
        A = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(Nn, K, generator=g) * 0.05).to(dev)
        Cb = torch.empty(M, Nn, device=dev)
        bias = torch.randn(Nn, generator=g).to(dev)
        R = torch.randn(M, Nn, device=dev) if resid else None
        G0 = torch.randn(N, 2 * Nn, device=dev) if gather else None
        gi = torch.randint(0, N, (M,), dtype=torch.int32, device=dev) if gather else None

        hi = torch.empty(Nn * K + 128, dtype=torch.int16, device=dev)
        lo = torch.empty_like(hi)
        if a.prec:
            L.check(lib.vlsat_k_split_bf16(W.data_ptr(), Nn * K, hi.data_ptr(), lo.data_ptr(), L.stream_ptr()))

        def run_planes(pf):
            L.check(lib.vlsat_k_gemm_planes(A.data_ptr(), K, W.data_ptr(), hi.data_ptr(), lo.data_ptr(), K, Cb.data_ptr(), Nn, M, Nn, K,
                                            bias.data_ptr(), L.ptr(R), Nn if resid else 0, 1.0,
                                            L.ptr(G0), L.ptr(gi), 2 * Nn if gather else 0,
                                            (G0.data_ptr() + 4 * Nn) if gather else 0, L.ptr(gi), 2 * Nn if gather else 0,
                                            0, 1, a.prec, int(a.no_dma), pf, a.fmt | (64 if a.splitk else 0), 1.0, L.stream_ptr()))

        def run():
            L.check(lib.vlsat_k_gemm(A.data_ptr(), K, W.data_ptr(), K, Cb.data_ptr(), Nn, M, Nn, K, bias.data_ptr(), 0,
                                     L.ptr(R), Nn if resid else 0, 1.0,
                                     L.ptr(G0), L.ptr(gi), 2 * Nn if gather else 0,
                                     (G0.data_ptr() + 4 * Nn) if gather else 0, L.ptr(gi), 2 * Nn if gather else 0,
                                     (2 if a.no_prefetch else 0) | (4 if a.splitk else 0) | (8 if a.no_p8 else 0), 1, L.stream_ptr()))
        variants = [("", run)] if not a.prec else [(f" pf={d}", (lambda d=d: run_planes(int(d)))) for d in a.prefetch.split(",")]
        for tag, fn in variants:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            tf = 2.0 * M * Nn * K / (ms * 1e-3) / 1e12
            print(f"{name + tag:42s} M={M:6d} {ms * 1e3:9.1f} us  {tf:7.1f} TF  {100 * tf / peak:5.1f} %", flush=True)


if __name__ == "__main__":
    main()
