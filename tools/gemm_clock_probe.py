#!/usr/bin/env python3
"""Shader clock the fp32 GEMM actually runs at (DVFS), by in-kernel s_memtime vs s_memrealtime.
    python tools/gemm_clock_probe.py          -> profiles/r01_gemm_clock.txt is a copy of its output"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402
from vlsat_amd import lib as L  # noqa: E402

lib = L.load()
dev = "cuda:0"
buf = torch.zeros(4 * 1024, dtype=torch.int64, device=dev)
for name, M, N, K, scale in (("E x512 x512  A~N(0,1) W~N(0,0.05)", 99840 - 1536, 512, 512, 1.0),
                             ("E x512 x1024 A~N(0,1) W~N(0,0.05)", 99840 - 1536, 512, 1024, 1.0),
                             ("E x512 x512  A = W = 0", 99840 - 1536, 512, 512, 0.0)):
    A = torch.randn(M, K, device=dev) * scale
    W = torch.randn(N, K, device=dev) * 0.05 * scale
    Cb = torch.empty(M, N, device=dev)

    def run():
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
    mfma_cycles = 2.0 * M * N * K / 4096 * 64 / 1024            # busy cycles per SIMD
    # two blocks share a CU and the first-dispatched one finishes early (tools/gemm_block_probe.py): the SIMD is
    # occupied until the slower one ends = the upper half of the sorted per-block cycle counts
    cyc = b[:, 0].sort().values
    slow = cyc[len(cyc) // 2:].mean().item()
    print(f"{name:36s} {ms * 1e3:7.1f} us  {tf:6.1f} TFLOP/s  shader clock {ghz:.3f} GHz  "
          f"(fp32-MFMA ceiling at that clock {157.3 * ghz / 2.4:.1f} TF; matrix pipe busy "
          f"{100 * mfma_cycles / slow:.0f} % of the cycles the slower block of each CU ran)")
