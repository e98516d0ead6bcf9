#!/usr/bin/env python3
"""The 256 x 256 8-phase bf16 GEMM (csrc/gemm_bf16_p8.hip) against the kernels it replaces, through the C ABI:
bit-identity with the 128 x 128 kernel on the forward's half-row shapes, then launch times next to the ring kernel.
    python tools/p8_check.py [--iters 20] [--no-check] [--rows M]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402,F401
from vlsat_amd import lib as L  # noqa: E402

DEV = "cuda:0"
HALF_A, HALF_R, HALF_C, HALF = 1, 2, 4, 32
NO_RING, NO_P8 = 16, 1 << 12


def to_half_rows(x):
    M, N = x.shape
    out = torch.zeros(M, N, dtype=torch.float32)
    out.view(torch.bfloat16).view(M, 2 * N)[:, :N] = x.to(torch.bfloat16)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--rows", type=int, default=99840)
    ap.add_argument("--only", default="", help="substring of the shape names to run")
    ap.add_argument("--cold", type=int, default=1, help="cycle through this many copies of A and C (4: nothing a launch reads is in a cache)")
    ap.add_argument("--structured", action="store_true", help="gather indices of the bench batch instead of random ones")
    ap.add_argument("--ab", action="store_true", help="round 6: K-tile rotation per column tile, A/B on the release library")
    ap.add_argument("--ablate", action="store_true", help="time the kernel with loads / MFMAs / fragment reads removed (first shape)")
    a = ap.parse_args()
    lib = L.load()
    g = torch.Generator().manual_seed(0)
    NG = 2560
    shapes = [  # name, N, K, resid, gather, relu_a, act, c_half
        ("kproj/q/fc1 Ex512x512", 512, 512, 0, 0, 0, 1, 1),
        ("kv Ex1024x512", 1024, 512, 0, 0, 0, 0, 1),
        ("nn_edge.2 Ex512x1024", 512, 1024, 0, 0, 0, 0, 1),
        ("fc2 Ex256x512", 256, 512, 0, 0, 0, 1, 1),
        ("out-proj+resid Ex512x512 (fp32 out)", 512, 512, 1, 0, 0, 0, 0),
        ("nn_edge.0+gather Ex1024x512", 1024, 512, 0, 1, 1, 1, 1),
    ]
    M = a.rows
    for name, N, K, resid, gather, relu_a, act, c_half in shapes:
        if a.only and a.only not in name:
            continue
        A = torch.randn(M, K, generator=g)
        Ah = to_half_rows(A).to(DEV)
        W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        R = to_half_rows(torch.randn(M, N, generator=g)).to(DEV) if resid else None
        G0 = torch.randn(NG, 2 * N, generator=g).to(DEV) if gather else None
        gi0 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV) if gather else None
        gi1 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV) if gather else None
        if gather and a.structured:           # the bench batch: scenes of 40 objects, every ordered pair, source-major
            n = 40
            i, j = torch.meshgrid(torch.arange(n), torch.arange(n), indexing="ij")
            keep = i != j
            src = torch.cat([i[keep] + n * s for s in range(M // (n * (n - 1)))])
            dst = torch.cat([j[keep] + n * s for s in range(M // (n * (n - 1)))])
            gi0, gi1 = src.int().to(DEV), dst.int().to(DEV)
        hi = torch.empty(N * K + 128, dtype=torch.int16, device=DEV)
        lo = torch.empty_like(hi)
        L.check(lib.vlsat_k_split_bf16(W.data_ptr(), N * K, hi.data_ptr(), lo.data_ptr(), L.stream_ptr()))
        fmt = HALF | HALF_A | (HALF_C if c_half else 0) | (HALF_R if resid else 0)
        C = torch.empty(M, N, device=DEV)

        As = [Ah] + [Ah.clone() for _ in range(a.cold - 1)]
        Cs = [C] + [torch.empty_like(C) for _ in range(a.cold - 1)]
        turn = [0]

        def run(f):
            turn[0] += 1
            Ax, Cx = As[turn[0] % a.cold], Cs[turn[0] % a.cold]
            L.check(lib.vlsat_k_gemm_planes(Ax.data_ptr(), K, W.data_ptr(), hi.data_ptr(), lo.data_ptr(), K, Cx.data_ptr(), N, M, N, K,
                                            b.data_ptr(), L.ptr(R), N if resid else 0, 0.5,
                                            L.ptr(G0), L.ptr(gi0), 2 * N if gather else 0,
                                            (G0.data_ptr() + 4 * N) if gather else 0, L.ptr(gi1), 2 * N if gather else 0,
                                            relu_a, act, 1, 0, -1, f, 1.0, L.stream_ptr()))
        if not a.no_check:
            C.fill_(float("nan"))
            run(fmt)
            torch.cuda.synchronize()
            got = C.clone()
            C.fill_(float("nan"))
            run(fmt | NO_RING | NO_P8)
            torch.cuda.synchronize()
            ref = C.clone()
            if c_half:
                gv, rv = got.view(torch.bfloat16).view(M, 2 * N)[:, :N].float(), ref.view(torch.bfloat16).view(M, 2 * N)[:, :N].float()
            else:
                gv, rv = got, ref
            bad = int((gv != rv).sum())
            err = float((gv - rv).abs().max())
            scale = float(rv.abs().max())
            nan = int((~torch.isfinite(gv)).sum())
            # the bias sits at the other end of the fp32 summation: a few results may round to the neighbouring bf16 / fp32 value
            tol = scale * 2.0 ** -7 if c_half else scale * 1e-5
            ok = nan == 0 and err <= tol and bad <= gv.numel() * (0.02 if c_half else 1.0)
            print(f"{name:40s} check: {'OK ' if ok else 'FAIL'} {bad} of {gv.numel()} differ, max abs diff {err:.3e} (max |ref| {scale:.2f}), non-finite {nan}", flush=True)
            if not ok:
                d = ((gv - rv).abs() > tol).nonzero()
                print("   first bad (row, col):", d[:8].tolist(), " rows hit:", int(((gv - rv).abs() > tol).any(1).sum()), flush=True)
        variants = [("p8", fmt), ("ring", fmt | NO_P8)]
        if a.ab:                                   # round 6 A/B, release library: K-tile rotation per column tile
            variants = [("p8", fmt)] + [(f"k_rot={r}", fmt | (r << 22)) for r in (1, 2, 3)] + [("p8 again", fmt)]
        if a.ablate and name.startswith("kproj"):
            ab = lambda bits: ((bits & 3) << 8) | (((bits >> 2) & 63) << 13)
            variants += [("-load", fmt | ab(1)), ("-mfma", fmt | ab(2)), ("-read", fmt | ab(4)), ("-ld-mf", fmt | ab(3)), ("-mf-rd", fmt | ab(6)), ("none", fmt | ab(7)),
                         ("-epi", fmt | ab(8)), ("bars", fmt | ab(15)),
                         ("mfma+bars", fmt | ab(5)), ("mfma+B1", fmt | ab(21)), ("mfma nobar", fmt | ab(37)), ("full-B2", fmt | ab(16)),
                         ("-ld rd-nowait", fmt | ab(65)), ("rd-nowait", fmt | ab(64)), ("noprio", fmt | ab(128)), ("-ld noprio", fmt | ab(129))]
        if a.ablate and gather:
            ab = lambda bits: ((bits & 3) << 8) | (((bits >> 2) & 63) << 13)
            variants += [("row-contiguous init loads", fmt | ab(1)), ("no init loads", fmt | ab(2))]
        for tag, f in variants:
            for _ in range(3):
                run(f)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run(f)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            print(f"{name:40s} {tag:14s} {ms * 1e3:9.1f} us  {tf:7.1f} TF  {100 * tf / 2500:5.1f} %", flush=True)


if __name__ == "__main__":
    main()
