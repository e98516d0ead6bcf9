#!/usr/bin/env python3
"""Shader clock the GEMM actually runs at (DVFS), by in-kernel s_memtime vs s_memrealtime.
    python tools/gemm_clock_probe.py [--prec 0|1|3]   (bf16 modes: the 128 x 128 LDS-direct kernel, which carries the probe)
    -> profiles/r01_gemm_clock.txt (fp32), profiles/r02_probes/gemm_clock_bf16.txt are copies of its output"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402
from vlsat_amd import lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prec", type=int, default=0, choices=[0, 1, 3])
args = ap.parse_args()
PEAK = {0: 157.3, 1: 2500.0, 3: 2500.0 / 3}[args.prec]
lib = L.load()
dev = "cuda:0"
buf = torch.zeros(4 * 1024, dtype=torch.int64, device=dev)
for name, M, N, K, scale in (("E x512 x512  A~N(0,1) W~N(0,0.05)", 99840 - 1536, 512, 512, 1.0),
                             ("E x512 x1024 A~N(0,1) W~N(0,0.05)", 99840 - 1536, 512, 1024, 1.0),
                             ("E x512 x512  A = W = 0", 99840 - 1536, 512, 512, 0.0)):
    A = torch.randn(M, K, device=dev) * scale
    W = torch.randn(N, K, device=dev) * 0.05 * scale
    Cb = torch.empty(M, N, device=dev)
    hi = torch.empty(N * K + 128, dtype=torch.int16, device=dev)
    lo = torch.empty_like(hi)
    if args.prec:
        L.check(lib.vlsat_k_split_bf16(W.data_ptr(), N * K, hi.data_ptr(), lo.data_ptr(), L.stream_ptr()))

    def run():
        if args.prec:
            L.check(lib.vlsat_k_gemm_planes(A.data_ptr(), K, W.data_ptr(), hi.data_ptr(), lo.data_ptr(), K, Cb.data_ptr(), N, M, N, K,
                                            0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, args.prec, 0, -1, 16, 1.0, L.stream_ptr()))
            return
        L.check(lib.vlsat_k_gemm(A.data_ptr(), K, W.data_ptr(), K, Cb.data_ptr(), N, M, N, K, 0, 0, 0, 0, 1.0,
                                 0, 0, 0, 0, 0, 0, 0, 0, L.stream_ptr()))
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    buf.zero_()
    L.check(lib.vlsat_debug_gemm_clock_probe(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    L.check(lib.vlsat_debug_gemm_clock_probe(None))
    b = buf.view(-1, 4).cpu()
    b = b[b[:, 3] == 1].double()
    ghz = (b[:, 0] / (b[:, 1] / 1e8)).mean().item() / 1e9
    ms = e0.elapsed_time(e1)
    tf = 2.0 * M * N * K / ms / 1e9
    # busy cycles per SIMD: fp32 32x32x2 = 4096 flop in 64 cycles; bf16 32x32x16 = 32768 flop in 32 cycles, args.prec of them per product
    mfma_cycles = 2.0 * M * N * K / 4096 * 64 / 1024 if not args.prec else args.prec * 2.0 * M * N * K / 32768 * 32 / 1024
    # two blocks share a CU and the first-dispatched one finishes early (tools/gemm_block_probe.py): the SIMD is
    # occupied until the slower one ends = the upper half of the sorted per-block cycle counts
    cyc = b[:, 0].sort().values
    slow = cyc[len(cyc) // 2:].mean().item()
    print(f"{name:36s} {ms * 1e3:7.1f} us  {tf:6.1f} TFLOP/s  shader clock {ghz:.3f} GHz  "
          f"(MFMA ceiling of the mode at that clock {PEAK * ghz / 2.4:.1f} TF; matrix pipe busy "
          f"{100 * mfma_cycles / slow:.0f} % of the cycles the slower block of each CU ran)")
