#!/usr/bin/env python3
"""Cycle budget of the opt-in one-wave-per-SIMD GEMM (gemm_f32_big.hip): shader cycles per block vs the MFMA
cycles it needs, over rounds (tiles per block) and K (steps per tile).  A linear fit of `lost` gives the per-tile
and per-step overheads quoted in DESIGN.md section 5.   python tools/gemm_step_probe.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import vlsat_amd
from vlsat_amd import lib as L
lib = L.load(); dev = "cuda:0"
L.check(lib.vlsat_debug_gemm_variant(2))
buf = torch.zeros(4 * 1024, dtype=torch.int64, device=dev)
N = 512
for rounds in (1, 2, 6):
    for K in (512, 1024, 2048):
        M = rounds * 64 * 256   # 256 blocks * rounds tiles / 4 n-tiles
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
        Cb = torch.empty(M, N, device=dev)
        def run():
            L.check(lib.vlsat_k_gemm(A.data_ptr(), K, W.data_ptr(), K, Cb.data_ptr(), N, M, N, K, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, L.stream_ptr()))
        for _ in range(5): run()
        torch.cuda.synchronize(); buf.zero_()
        L.check(lib.vlsat_debug_gemm_clock_probe(buf.data_ptr()))
        run(); torch.cuda.synchronize()
        L.check(lib.vlsat_debug_gemm_clock_probe(None))
        b = buf.view(-1, 4).cpu(); b = b[b[:, 3] == 1].double()
        cyc = b[:, 0].mean().item(); ghz = (b[:, 0] / (b[:, 1] / 1e8)).mean().item() / 1e9
        steps = rounds * K // 16
        print(f"rounds {rounds} K {K}: blocks {len(b)} cycles {cyc:9.0f}  mfma {steps*4096:9d}  lost {cyc-steps*4096:8.0f}  per-step-equivalent {(cyc-steps*4096)/steps:6.0f}  clock {ghz:.2f} GHz  min/max {b[:,0].min().item():.0f}/{b[:,0].max().item():.0f}")
