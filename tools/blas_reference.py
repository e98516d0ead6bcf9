#!/usr/bin/env python3
"""Known-good reference for the fp32 GEMM ceiling on this chip: the vendor library behind torch.matmul
(rocBLAS / hipBLASLt) on the same shapes and the same random data as tools/gemm_bench.py.
Not part of the product path; it answers "what does a tuned fp32 GEMM reach here?"."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda:0"
for name, M, N, K in (("E x512 x512", 99840, 512, 512), ("E x1024 x512", 99840, 1024, 512), ("E x512 x1024", 99840, 512, 1024),
                      ("4096^3", 4096, 4096, 4096), ("8192 x 8192 x 512", 8192, 8192, 512)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
    for _ in range(5): C = A @ W.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): C = A @ W.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:20s} {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:6.1f} TFLOP/s (torch.matmul fp32)")
