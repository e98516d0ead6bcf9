#!/usr/bin/env python3
"""Soak test of the split-K exchange (gemm_splitk.hip): partial tiles travel between blocks through device-coherent stores /
loads and an arrival counter, without fences.  Every launch must reproduce the first result bit for bit (the reduction
order is fixed) and agree with fp64; a lost or stale partial sum would show as a mismatch.
    python tools/splitk_soak.py [--iters 3000]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402
from vlsat_amd import lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=3000)
ap.add_argument("--load", action="store_true", help="keep a second stream busy with large matmuls (memory + cache pressure) meanwhile")
a = ap.parse_args()
lib = L.load()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
bad = 0
side = torch.cuda.Stream()
X = torch.randn(8192, 8192, device=dev)
Y = torch.empty_like(X)


def background():
    if a.load:
        with torch.cuda.stream(side):
            for _ in range(2):
                torch.matmul(X, X, out=Y)
                Y.copy_(X)


for M, N, K in ((40, 512, 512), (80, 3328, 512), (9, 1536, 512), (600, 512, 1024), (1500, 1024, 512), (72, 160, 512)):
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev)
    outs = [torch.empty(M, N, device=dev) for _ in range(8)]
    ref = (A.double() @ W.double().t() + b.double() + R.double()).float()

    def run(C):
        L.check(lib.vlsat_k_gemm(A.data_ptr(), K, W.data_ptr(), K, C.data_ptr(), N, M, N, K, b.data_ptr(), 0, R.data_ptr(), N, 1.0,
                                 0, 0, 0, 0, 0, 0, 4, 0, L.stream_ptr()))
    run(outs[0])
    torch.cuda.synchronize()
    first = outs[0].clone()
    err = float((first - ref).abs().max())
    mism = 0
    for it in range(a.iters):
        C = outs[it % 8]
        C.fill_(float("nan"))
        run(C)
        if it % 8 == 0:
            background()
        if it % 8 == 7:                      # 8 launches back to back, then compare them all
            torch.cuda.synchronize()
            mism += sum(int(not torch.equal(o, first)) for o in outs)
    torch.cuda.synchronize()
    bad += mism
    print(f"M={M:5d} N={N:5d} K={K:5d}: {a.iters} launches, {mism} differ from the first; max |err| vs fp64 {err:.2e}")
print("RESULT:", "ok" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
