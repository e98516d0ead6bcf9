"""Single-kernel parity tests: each hand-written HIP kernel through the C ABI against a plain
PyTorch fp32/fp64 CPU reference of the same op.  Needs an MI355X:  pytest -m gpu"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

import vlsat_amd  # noqa: F401
from vlsat_amd import VLSATConfig, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    from vlsat_amd import lib as L
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path cannot run and there is no fallback")
    return L


def _sync():
    torch.cuda.synchronize()


def _gemm(L, A, W, bias=None, rowscale=None, resid=None, resid_scale=1.0, g0=None, gi0=None, g1=None, gi1=None,
          relu_a=0, act=0, ldc=None):
    l = L.load()
    M, K = A.shape
    N = W.shape[0]
    ldc = ldc or N
    Cbuf = torch.full((M, ldc), float("nan"), device=DEV)
    L.check(l.vlsat_k_gemm(A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), Cbuf.data_ptr(), ldc, M, N, K,
                           L.ptr(bias), L.ptr(rowscale), L.ptr(resid), 0 if resid is None else resid.stride(0),
                           resid_scale, L.ptr(g0), L.ptr(gi0), 0 if g0 is None else g0.stride(0),
                           L.ptr(g1), L.ptr(gi1), 0 if g1 is None else g1.stride(0), relu_a, act, L.stream_ptr()))
    _sync()
    return Cbuf[:, :N].cpu()


def _ref_gemm(A, W, bias=None, rowscale=None, resid=None, resid_scale=1.0, g0=None, gi0=None, g1=None, gi1=None,
              relu_a=0, act=0):
    A, W = A.double().cpu(), W.double().cpu()
    if relu_a:
        A = A.clamp_min(0)
    y = A @ W.t()
    if rowscale is not None:
        y = y * rowscale.double().cpu()[:, None]
    if bias is not None:
        y = y + bias.double().cpu()
    if resid is not None:
        y = y + resid_scale * resid.double().cpu()
    if g0 is not None:
        y = y + g0.double().cpu()[gi0.cpu().long()][:, :y.shape[1]]
    if g1 is not None:
        y = y + g1.double().cpu()[gi1.cpu().long()][:, :y.shape[1]]
    if act == 1:
        y = y.clamp_min(0)
    elif act == 2:
        y = torch.sigmoid(y)
    return y.float()


@pytest.mark.parametrize("M,N,K", [(1, 512, 512), (8, 504, 768), (56, 26, 256), (130, 160, 512), (1560, 1024, 512),
                                   (2560, 3328, 512), (4097, 128, 64), (9000, 512, 1024), (70000, 64, 128)])
def test_gemm_plain_shapes(lib, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    got = _gemm(lib, A, W, bias=b)
    ref = _ref_gemm(A, W, bias=b)
    err = float((got - ref).abs().max())
    assert err < 2e-5 * math.sqrt(K / 64) + 1e-5, f"gemm {M}x{N}x{K}: {err:.3e}"


def test_gemm_transpose_detecting_identity(lib):
    """A = I with an asymmetric W: a swapped C-write or operand would show (guide §3)."""
    K = 64
    A = torch.eye(K).to(DEV)
    W = (torch.arange(96 * K, dtype=torch.float32).view(96, K) * 1e-3).to(DEV)   # W[n,k] = (n*K + k)/1000
    got = _gemm(lib, A, W)
    assert torch.equal(got, W.cpu().t().contiguous())


def test_gemm_all_epilogues(lib):
    g = torch.Generator().manual_seed(11)
    M, N, K, NG = 777, 200, 96, 37
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    rs = (torch.rand(M, generator=g) + 0.5).to(DEV)
    resid = torch.randn(M, N + 8, generator=g).to(DEV)[:, :N]          # strided residual
    gbuf = torch.randn(NG, 2 * N + 4, generator=g).to(DEV)
    gi0 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    gi1 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    g0v, g1v = gbuf[:, :N], gbuf[:, N:2 * N]
    for act in (0, 1, 2):
        for relu_a in (0, 1):
            for kw in (dict(bias=bias, resid=resid, resid_scale=0.5, g0=g0v, gi0=gi0, g1=g1v, gi1=gi1),
                       dict(bias=bias, rowscale=rs), dict(resid=resid), dict(g1=g1v, gi1=gi1)):
                kw = dict(kw, relu_a=relu_a, act=act)
                got = _gemm(lib, A, W, ldc=N + 3, **kw)
                ref = _ref_gemm(A, W, **kw)
                err = float((got - ref).abs().max())
                assert err < 3e-5, f"act={act} relu_a={relu_a} {sorted(kw)}: {err:.3e}"
    with pytest.raises(lib.VlsatError):
        _gemm(lib, A, W, rowscale=rs, resid=resid)


def test_gemm_persistent_rounds_with_epilogues(lib):
    """Several rounds of the persistent grid plus the small-tile tail launch, with gathered
    rows, residual and ReLU-on-A (the nn_edge / out-proj launch shapes in miniature)."""
    g = torch.Generator().manual_seed(21)
    M, N, K, NG = 3 * 512 * 128 // 4 + 777, 512, 64, 301      # 1536 tiles of 128x128 -> 3 rounds + tail rows
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    gbuf = torch.randn(NG, 2 * N, generator=g).to(DEV)
    gi0 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    gi1 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    kw = dict(bias=bias, resid=resid, resid_scale=1.0, g0=gbuf[:, :N], gi0=gi0, g1=gbuf[:, N:], gi1=gi1, relu_a=1, act=1)
    got = _gemm(lib, A, W, **kw)
    ref = _ref_gemm(A, W, **kw)
    err = float((got - ref).abs().max())
    assert err < 3e-5, f"{err:.3e}"


@pytest.mark.parametrize("M,N,K,lda", [(1000, 300, 96, 96), (70000, 512, 512, 512), (4097, 130, 64, 200), (33, 26, 256, 256)])
def test_gemm_lds_direct_pipe_is_bit_identical_to_vgpr_pipe(lib, M, N, K, lda):
    """Launches without ReLU-on-A stage their operands with buffer_load ... lds into swizzled LDS rows
    (PipeF32Dma), ReLU-on-A launches through registers (PipeF32).  On a non-negative A both compute the same
    sums in the same order, so the outputs must agree bit for bit -- ragged M/N (rows past the edge come back as
    zeros from the buffer descriptor instead of being clamped) and a strided A included."""
    g = torch.Generator().manual_seed(M + N)
    Abuf = torch.randn(M, lda, generator=g).abs().to(DEV)
    A = Abuf[:, :K]
    W = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    for kw in (dict(bias=bias, act=1), dict(bias=bias, resid=resid, resid_scale=0.25)):
        dma = _gemm(lib, A, W, relu_a=0, **kw)
        vgpr = _gemm(lib, A, W, relu_a=1, **kw)
        assert torch.equal(dma, vgpr), f"max diff {float((dma - vgpr).abs().max()):.3e}"
    ref = _ref_gemm(A, W, bias=bias, act=1)
    assert float((_gemm(lib, A, W, bias=bias, act=1) - ref).abs().max()) < 2e-5 * math.sqrt(K / 64) + 2e-5


def test_gemm_strided_a_and_inplace_residual(lib):
    """The forward feeds A with a 768 pitch and adds the residual in place (C == resid)."""
    g = torch.Generator().manual_seed(5)
    M, K, N = 300, 512, 512
    X = torch.randn(M, 768, generator=g).to(DEV)
    O = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
    ref = _ref_gemm(O, W, resid=X[:, :N])
    l = lib.load()
    lib.check(l.vlsat_k_gemm(O.data_ptr(), K, W.data_ptr(), K, X.data_ptr(), 768, M, N, K, 0, 0, X.data_ptr(), 768, 1.0,
                             0, 0, 0, 0, 0, 0, 0, 0, lib.stream_ptr()))
    _sync()
    assert float((X[:, :N].cpu() - ref).abs().max()) < 3e-5


def test_gemm_rejects_bad_k(lib):
    A = torch.zeros(4, 48, device=DEV)
    W = torch.zeros(4, 48, device=DEV)
    with pytest.raises(lib.VlsatError):
        _gemm(lib, A, W)


# ------------------------------------------------------------------------------------------------
def _pointnet(L, pts, w):
    l = L.load()
    n, _, p = pts.shape
    out = torch.full((n, 768), float("nan"), device=DEV)
    d = {k: torch.from_numpy(np.ascontiguousarray(w["obj_encoder." + k])).to(DEV) for k in
         ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "conv3.weight", "conv3.bias")}
    L.check(l.vlsat_k_pointnet(pts.data_ptr(), n, p, d["conv1.weight"].data_ptr(), d["conv1.bias"].data_ptr(),
                               d["conv2.weight"].data_ptr(), d["conv2.bias"].data_ptr(),
                               d["conv3.weight"].data_ptr(), d["conv3.bias"].data_ptr(), 768, out.data_ptr(),
                               L.stream_ptr()))
    _sync()
    return out.cpu()


@pytest.mark.parametrize("n,p", [(8, 256), (3, 1024), (5, 100), (1, 1), (700, 64), (40, 128)])
def test_pointnet_vs_oracle(lib, n, p):
    from oracle import vlsat_oracle as O
    w = synth.make_weights(VLSATConfig())
    b = synth.make_batch(1, n, p, seed0=77 + n)
    pts = torch.from_numpy(b["obj_points"])
    ref = O.pointnet_feat(pts.double(), O.to_torch(w, torch.float64), "obj_encoder").float()
    got = _pointnet(lib, pts.to(DEV), w)
    err = float((got - ref).abs().max())
    assert err < 2e-5, f"pointnet n={n} p={p}: {err:.3e}"


def test_pointnet_output_needs_only_4_byte_alignment(lib):
    """vlsat_k_pointnet with an output pointer that is 4- but not 16-byte aligned, few objects (the points of an object are
    split over several blocks that merge with atomicMax: the start value comes from the library's own fill kernel, which takes
    the scalar path here) -- same result as the aligned call, and nothing written outside the [n, 768] block."""
    w = synth.make_weights(VLSATConfig())
    b = synth.make_batch(1, 5, 512, seed0=321)
    pts = torch.from_numpy(b["obj_points"]).to(DEV)
    want = _pointnet(lib, pts, w)
    l = lib.load()
    buf = torch.full((5 * 768 + 8,), float("nan"), device=DEV)
    out = buf[1:1 + 5 * 768]
    d = {k: torch.from_numpy(np.ascontiguousarray(w["obj_encoder." + k])).to(DEV) for k in
         ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "conv3.weight", "conv3.bias")}
    lib.check(l.vlsat_k_pointnet(pts.data_ptr(), 5, 512, d["conv1.weight"].data_ptr(), d["conv1.bias"].data_ptr(),
                                 d["conv2.weight"].data_ptr(), d["conv2.bias"].data_ptr(), d["conv3.weight"].data_ptr(),
                                 d["conv3.bias"].data_ptr(), 768, out.data_ptr(), lib.stream_ptr()))
    _sync()
    assert out.data_ptr() % 16 == 4 and torch.equal(out.cpu().view(5, 768), want)
    assert bool(torch.isnan(buf[0])) and bool(torch.isnan(buf[1 + 5 * 768:]).all())


def test_pointnet_golden(lib, golden_dir):
    import os
    w = synth.make_weights(VLSATConfig(N_LAYERS=3))
    z = np.load(os.path.join(golden_dir, "pointnet_n3_p1024.npz"))
    b = synth.make_batch(1, 3, 1024, seed0=5000)
    got = _pointnet(lib, torch.from_numpy(b["obj_points"]).to(DEV), w)
    assert float(np.abs(got.numpy() - z["obj_encoder"]).max()) < 2e-5
    z1 = np.load(os.path.join(golden_dir, "cfg1_n8_p256_l2.npz"))
    b1 = synth.make_batch(1, 8, 256, seed0=1000)
    got1 = _pointnet(lib, torch.from_numpy(b1["obj_points"]).to(DEV), synth.make_weights(VLSATConfig(N_LAYERS=2)))
    assert float(np.abs(got1.numpy() - z1["tap.obj_encoder"]).max()) < 2e-5


# ------------------------------------------------------------------------------------------------
def _flash(L, q, k, v, tok_ptr, scale):
    l = L.load()
    o = torch.full_like(q, float("nan"))
    tp = torch.tensor(tok_ptr, dtype=torch.int64)
    L.check(l.vlsat_k_flash_attn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), q.stride(0), tp.data_ptr(),
                                 len(tok_ptr) - 1, 8, scale, L.stream_ptr()))
    _sync()
    return o.cpu()


def _ref_attn(q, k, v, tok_ptr, scale):
    q, k, v = q.double().cpu(), k.double().cpu(), v.double().cpu()
    out = torch.zeros_like(q)
    for s in range(len(tok_ptr) - 1):
        a, b = tok_ptr[s], tok_ptr[s + 1]
        qq = q[a:b].view(b - a, 8, 64).permute(1, 0, 2)
        kk = k[a:b].view(b - a, 8, 64).permute(1, 2, 0)
        vv = v[a:b].view(b - a, 8, 64).permute(1, 0, 2)
        att = torch.softmax(qq @ kk * scale, -1)
        out[a:b] = (att @ vv).permute(1, 0, 2).reshape(b - a, 512)
    return out.float()


@pytest.mark.parametrize("tok_ptr", [[0, 56], [0, 1560], [0, 20, 65, 66, 400], [0, 1], [0, 129, 129 + 32]])
def test_flash_attn_vs_softmax(lib, tok_ptr):
    g = torch.Generator().manual_seed(tok_ptr[-1])
    T = tok_ptr[-1]
    q = torch.randn(T, 512, generator=g).to(DEV)
    k = torch.randn(T, 512, generator=g).to(DEV)
    v = torch.randn(T, 512, generator=g).to(DEV)
    got = _flash(lib, q, k, v, tok_ptr, 0.125)
    ref = _ref_attn(q, k, v, tok_ptr, 0.125)
    err = float((got - ref).abs().max())
    assert err < 2e-5, f"flash {tok_ptr}: {err:.3e}"


def test_flash_attn_forced_rescale(lib):
    """Force the online-softmax rescale branch: one late key dominates one query (guide rule 26),
    and large-magnitude scores exercise the running-max bookkeeping."""
    g = torch.Generator().manual_seed(9)
    T = 300
    q = torch.randn(T, 512, generator=g)
    k = torch.randn(T, 512, generator=g)
    v = torch.randn(T, 512, generator=g)
    k[257] = q[5] * 4.0          # key in the 9th tile spikes against query 5 (all heads)
    k[31] = -q[100] * 6.0
    q[200] *= 30.0               # huge logits for one query
    got = _flash(lib, q.to(DEV), k.to(DEV), v.to(DEV), [0, T], 0.125)
    ref = _ref_attn(q, k, v, [0, T], 0.125)
    err = float((got - ref).abs().max())
    assert torch.isfinite(got).all()
    assert err < 5e-5, f"{err:.3e}"


def test_layernorm(lib):
    g = torch.Generator().manual_seed(2)
    for rows, relu in ((1, 0), (7, 1), (1560, 0), (4099, 1)):
        x = (torch.randn(rows, 768, generator=g) * 3 + 1).to(DEV)
        gamma, beta = torch.randn(512, generator=g).to(DEV), torch.randn(512, generator=g).to(DEV)
        ref = torch.nn.functional.layer_norm(x[:, :512].double().cpu(), (512,), gamma.double().cpu(), beta.double().cpu(), 1e-5)
        if relu:
            ref = ref.clamp_min(0)
        tail = x[:, 512:].clone()
        lib.check(lib.load().vlsat_k_layernorm(x.data_ptr(), 768, rows, 512, gamma.data_ptr(), beta.data_ptr(), relu,
                                               lib.stream_ptr()))
        _sync()
        assert float((x[:, :512].cpu() - ref.float()).abs().max()) < 2e-5
        assert torch.equal(x[:, 512:], tail), "layernorm wrote outside its 512 columns"


# ---- bf16 matrix-core kernels of BASELINE configs[2] -----------------------------------------------------------------
def _gemm_bf16(L, A, W, prec, no_dma=0, bias=None, resid=None, resid_scale=1.0, g0=None, gi0=None, g1=None, gi1=None,
               relu_a=0, act=0, ldc=None):
    l = L.load()
    M, K = A.shape
    N = W.shape[0]
    ldc = ldc or N
    Cbuf = torch.full((M, ldc), float("nan"), device=DEV)
    L.check(l.vlsat_k_gemm_bf16(A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), Cbuf.data_ptr(), ldc, M, N, K,
                                L.ptr(bias), L.ptr(resid), 0 if resid is None else resid.stride(0), resid_scale,
                                L.ptr(g0), L.ptr(gi0), 0 if g0 is None else g0.stride(0),
                                L.ptr(g1), L.ptr(gi1), 0 if g1 is None else g1.stride(0), relu_a, act, prec, no_dma,
                                L.stream_ptr()))
    _sync()
    return Cbuf[:, :N].cpu()


@pytest.mark.parametrize("M,N,K,lda", [(1, 512, 512, 512), (56, 26, 256, 256), (1560, 1024, 512, 512), (2560, 3328, 512, 768),
                                       (4097, 130, 64, 200), (9000, 512, 1024, 1024), (70000, 64, 128, 128)])
def test_gemm_split_bf16_lds_direct_pipe(lib, M, N, K, lda):
    """Split-bf16 GEMM (three bf16 MFMAs per product) with LDS-direct operand staging (PipeSplitDma): against fp64 at
    ~2^-16 relative accuracy, and BIT-IDENTICAL to the VGPR-staged pipe of round 1 (same hi/lo split, same MFMA
    order) -- ragged M/N, strided A, ReLU-on-A (applied at the fragment read), all additive modes the forward uses."""
    g = torch.Generator().manual_seed(M + 3 * N)
    A = torch.randn(M, lda, generator=g).to(DEV)[:, :K]
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    NG = 61
    gbuf = torch.randn(NG, 2 * N, generator=g).to(DEV)
    gi0 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    gi1 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    cases = [dict(bias=b), dict(bias=b, relu_a=1, act=1), dict(bias=b, resid=resid, resid_scale=0.5),
             dict(g0=gbuf[:, :N], gi0=gi0, g1=gbuf[:, N:], gi1=gi1, relu_a=1, act=1)]
    for kw in cases:
        ref = _ref_gemm(A, W, **kw)
        dma = _gemm_bf16(lib, A, W, 3, 0, **kw)
        old = _gemm_bf16(lib, A, W, 3, 1, **kw)
        err = float((dma - ref).abs().max())
        assert err < 1e-4 * math.sqrt(K / 64) + 2e-5, f"bf16x3 {M}x{N}x{K} {sorted(kw)}: {err:.3e}"
        assert torch.equal(dma, old), f"LDS-direct vs VGPR-staged differ: {float((dma - old).abs().max()):.3e} {sorted(kw)}"
        one = _gemm_bf16(lib, A, W, 1, 0, **kw)
        one_old = _gemm_bf16(lib, A, W, 1, 1, **kw)
        err1 = float((one - ref).abs().max())
        assert err1 < 2e-2 * math.sqrt(K / 64) + 2e-2, f"bf16 {M}x{N}x{K}: {err1:.3e}"
        assert torch.equal(one, one_old)


def test_gemm_split_bf16_transpose_detecting(lib):
    """A = I (exactly representable) with an asymmetric bf16-exact W: any k-slot / row / column mix-up shows."""
    K = 64
    A = torch.eye(K).to(DEV)
    W = (torch.arange(96 * K, dtype=torch.float32).view(96, K) % 251 - 125).to(DEV)       # |w| <= 125: exact in bf16
    for prec in (1, 3):
        got = _gemm_bf16(lib, A, W, prec)
        assert torch.equal(got, W.cpu().t().contiguous()), prec


def _flash_bf16(L, q, k, v, tok_ptr, scale, terms, use_tr):
    l = L.load()
    o = torch.full_like(q, float("nan"))
    tp = torch.tensor(tok_ptr, dtype=torch.int64)
    L.check(l.vlsat_k_flash_attn_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), q.stride(0), tp.data_ptr(),
                                      len(tok_ptr) - 1, 8, scale, terms, use_tr, L.stream_ptr()))
    _sync()
    return o.cpu()


@pytest.mark.parametrize("tok_ptr", [[0, 56], [0, 1560], [0, 20, 65, 66, 400], [0, 1], [0, 129, 129 + 32], [0, 64, 128, 257]])
def test_flash_attn_bf16_vs_softmax(lib, tok_ptr):
    """Split-bf16 attention (QK^T and PV on bf16 MFMAs, three per product) against the fp64 softmax attention; the
    LDS-transpose-read path for V must agree BIT FOR BIT with the gather path (same numbers, other load instruction)."""
    g = torch.Generator().manual_seed(sum(tok_ptr))
    T = tok_ptr[-1]
    q, k, v = (torch.randn(T, 512, generator=g).to(DEV) for _ in range(3))
    v = v * 3 + torch.arange(512, device=DEV)[None, :] * 0.01          # column-asymmetric values: transposes show
    ref = _ref_attn(q, k, v, tok_ptr, 0.125)
    tr = _flash_bf16(lib, q, k, v, tok_ptr, 0.125, 3, 1)
    ga = _flash_bf16(lib, q, k, v, tok_ptr, 0.125, 3, 0)
    assert torch.isfinite(ga).all() and torch.isfinite(tr).all()
    err = float((ga - ref).abs().max())
    assert err < 2e-4, f"bf16x3 attention (gather): {err:.3e}"
    assert torch.equal(tr, ga), f"transpose-read path differs from the gather path by {float((tr - ga).abs().max()):.3e}"
    one = _flash_bf16(lib, q, k, v, tok_ptr, 0.125, 1, 1)
    err1 = float((one - ref).abs().max())
    assert err1 < 8e-2, f"bf16 attention: {err1:.3e}"
    assert torch.equal(one, _flash_bf16(lib, q, k, v, tok_ptr, 0.125, 1, 0))


def test_flash_attn_bf16_forced_rescale(lib):
    """A key whose score dwarfs the others late in the sequence forces the running-maximum rescale (guide rule 26)."""
    g = torch.Generator().manual_seed(5)
    T = 700
    q, k, v = (torch.randn(T, 512, generator=g) for _ in range(3))
    k[613] = q[17] * 3.0
    k[64 * 5 + 1] = q[300] * 2.0
    q, k, v = q.to(DEV), k.to(DEV), v.to(DEV)
    ref = _ref_attn(q, k, v, [0, T], 0.125)
    got = _flash_bf16(lib, q, k, v, [0, T], 0.125, 3, 1)
    err = float((got - ref).abs().max())
    assert err < 2e-4, f"{err:.3e}"


def _pack_split(x):
    """Host restatement of common.h pack_split: (bf16 rne(x) << 16) | bf16 rne(x - hi), as float32 bit patterns."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    w = (hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xFFFF)
    return w.view(torch.float32)


def _unpack_split(w):
    u = w.view(torch.int32)
    return (u & -65536).view(torch.float32) + (u << 16).view(torch.float32)


def test_split_pair_format_gemm(lib):
    """The split-pair tensor format of the bf16 modes: A read as packed hi/lo words (ReLU on the packed word), residual
    decoded, C packed and scaled -- all against the plain-fp32 operand path of the same kernel on the same values."""
    l = lib.load()
    g = torch.Generator().manual_seed(77)
    M, N, K = 3000, 512, 256
    A = torch.randn(M, K, generator=g)
    A = _unpack_split(_pack_split(A))                       # values exactly representable as hi + lo
    R = _unpack_split(_pack_split(torch.randn(M, N, generator=g)))
    W = (torch.randn(N, K, generator=g) / 16).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    hi = torch.empty(N * K + 128, dtype=torch.int16, device=DEV)
    lo = torch.empty_like(hi)
    lib.check(l.vlsat_k_split_bf16(W.data_ptr(), N * K, hi.data_ptr(), lo.data_ptr(), lib.stream_ptr()))
    Ad, Rd, As, Rs = A.to(DEV), R.to(DEV), _pack_split(A).to(DEV), _pack_split(R).to(DEV)

    def run(a, r, fmt, prec, relu_a, scale=1.0):
        Cb = torch.full((M, N), float("nan"), device=DEV)
        lib.check(l.vlsat_k_gemm_planes(a.data_ptr(), K, W.data_ptr(), hi.data_ptr(), lo.data_ptr(), K, Cb.data_ptr(), N, M, N, K,
                                        b.data_ptr(), r.data_ptr(), N, 0.5, 0, 0, 0, 0, 0, 0, relu_a, 1, prec, 0, -1, fmt, scale,
                                        lib.stream_ptr()))
        _sync()
        return Cb.cpu()
    for prec in (3, 1):
        for relu_a in (0, 1):
            plain = run(Ad, Rd, 0, prec, relu_a)
            packed_in = run(As, Rs, 3, prec, relu_a)
            # (not bit-identical: where x - hi is an exact tie the in-kernel split of the fp32 operand picks another hi/lo
            #  pair with the same sum, which changes the fp32 summation order)
            d = float((plain - packed_in).abs().max())
            assert d < 2e-5, f"prec {prec} relu {relu_a}: packed operands differ from plain ones by {d:.3e}"
            packed_out = run(As, Rs, 7, prec, relu_a, 0.25)
            d = float((_unpack_split(packed_out) - plain * 0.25).abs().max())
            assert d < 2e-5, f"prec {prec} relu {relu_a}: packed + scaled output off by {d:.3e}"


def test_flash_attn_bf16_split_pair_io(lib):
    """Attention with Q (pre-scaled), K, V, O in the split-pair format == the fp32-I/O kernel on the same values."""
    g = torch.Generator().manual_seed(9)
    tok = [0, 70, 70 + 333]
    T = tok[-1]
    sc = 0.125 * 1.4426950408889634
    q, k, v = (_unpack_split(_pack_split(torch.randn(T, 512, generator=g))) for _ in range(3))
    qs = _unpack_split(_pack_split(q * sc))                 # what the Q projection's epilogue hands over
    ref = _ref_attn(qs / sc, k, v, tok, 0.125)
    for terms, tol in ((3, 3e-4), (1, 8e-2)):
        plain = _flash_bf16(lib, (qs / sc).to(DEV), k.to(DEV), v.to(DEV), tok, 0.125, terms, 1)
        packed = _flash_bf16(lib, _pack_split(qs).to(DEV), _pack_split(k).to(DEV), _pack_split(v).to(DEV), tok, 0.125, terms, 2)
        got = _unpack_split(packed)
        assert float((got - ref).abs().max()) < tol, (terms, float((got - ref).abs().max()))
        assert float((got - plain).abs().max()) < (2e-5 if terms == 3 else 2e-2)


@pytest.mark.parametrize("M,N,K,fmt", [(70000, 512, 512, 0), (66000, 1024, 512, 0), (99840, 512, 1024, 1), (70001, 200, 128, 0)])
def test_gemm_bf16_ring_kernel(lib, M, N, K, fmt):
    """Large-M bf16 launches: the full rounds run on the 3-stage ring kernel (256 x 128 tiles, two slices in flight), the
    rest on the 128 x 128 kernel.  Same k order and MFMA order, so the result must be BIT-IDENTICAL to the launch that
    keeps everything on the 128 x 128 kernel (fmt bit 4) -- residual, gathered rows, ReLU-on-A, ragged N included."""
    l = lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    if fmt & 1:
        A = _pack_split(A)
    A = A.to(DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    R = torch.randn(M, N, generator=g).to(DEV)
    NG = 999
    gbuf = torch.randn(NG, 2 * N, generator=g).to(DEV)
    gi0 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    gi1 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    hi = torch.empty(N * K + 128, dtype=torch.int16, device=DEV)
    lo = torch.empty_like(hi)
    lib.check(l.vlsat_k_split_bf16(W.data_ptr(), N * K, hi.data_ptr(), lo.data_ptr(), lib.stream_ptr()))

    def run(prec, f, resid, gather, relu_a):
        Cb = torch.full((M, N), float("nan"), device=DEV)
        lib.check(l.vlsat_k_gemm_planes(A.data_ptr(), K, W.data_ptr(), hi.data_ptr(), lo.data_ptr(), K, Cb.data_ptr(), N, M, N, K,
                                        b.data_ptr(), R.data_ptr() if resid else 0, N if resid else 0, 0.5,
                                        gbuf.data_ptr() if gather else 0, gi0.data_ptr() if gather else 0, 2 * N if gather else 0,
                                        gbuf.data_ptr() + 4 * N if gather else 0, gi1.data_ptr() if gather else 0, 2 * N if gather else 0,
                                        relu_a, 1, prec, 0, -1, f, 1.0, lib.stream_ptr()))
        _sync()
        return Cb.cpu()
    Af = (_unpack_split(A.cpu()) if fmt & 1 else A.cpu())
    for prec in (3, 1):
        for resid, gather, relu_a in ((0, 0, 0), (1, 0, 1), (0, 1, 0)):
            ring = run(prec, fmt, resid, gather, relu_a)
            flat = run(prec, fmt | 16, resid, gather, relu_a)
            assert torch.isfinite(ring).all()
            assert torch.equal(ring, flat), f"prec {prec} resid {resid} gather {gather}: ring vs 128x128 differ by {float((ring - flat).abs().max()):.3e}"
        kw = dict(bias=b, act=1)
        ref = _ref_gemm(Af.to(DEV), W, **kw)
        err = float((run(prec, fmt, 0, 0, 0) - ref).abs().max())
        assert err < (1e-4 if prec == 3 else 3e-2) * math.sqrt(K / 64) + 2e-5, f"prec {prec}: {err:.3e}"


def _to_half_rows(x):
    """fp32-pitched rows carrying bf16 values at byte 2 * column (the half-row format of the single-rounding modes)."""
    M, N = x.shape
    out = torch.zeros(M, N, dtype=torch.float32)
    out.view(torch.bfloat16).view(M, 2 * N)[:, :N] = x.to(torch.bfloat16)
    return out


def _from_half_rows(w, N):
    return w.view(torch.bfloat16).view(w.shape[0], -1)[:, :N].float()


def test_half_row_format_gemm_and_attention(lib):
    """Half-row tensors (plain bf16 inside fp32-pitched rows): GEMM with A / residual / C in that format, small and
    ring-kernel sizes, and the attention with Q / K / V / O in it, against the same kernels on fp32 tensors holding the
    same bf16-representable values."""
    l = lib.load()
    g = torch.Generator().manual_seed(123)
    for M in (3000, 70000):
        N, K = 512, 256
        A = torch.randn(M, K, generator=g).to(torch.bfloat16).float()
        R = torch.randn(M, N, generator=g).to(torch.bfloat16).float()
        W = (torch.randn(N, K, generator=g) / 16).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        hi = torch.empty(N * K + 128, dtype=torch.int16, device=DEV)
        lo = torch.empty_like(hi)
        lib.check(l.vlsat_k_split_bf16(W.data_ptr(), N * K, hi.data_ptr(), lo.data_ptr(), lib.stream_ptr()))
        Ad, Rd, Ah, Rh = A.to(DEV), R.to(DEV), _to_half_rows(A).to(DEV), _to_half_rows(R).to(DEV)

        def run(a, r, fmt, relu_a):
            Cb = torch.zeros(M, N, device=DEV)
            lib.check(l.vlsat_k_gemm_planes(a.data_ptr(), K, W.data_ptr(), hi.data_ptr(), lo.data_ptr(), K, Cb.data_ptr(), N, M, N, K,
                                            b.data_ptr(), r.data_ptr(), N, 0.5, 0, 0, 0, 0, 0, 0, relu_a, 1, 1, 0, -1, fmt, 1.0,
                                            lib.stream_ptr()))
            _sync()
            return Cb.cpu()
        for relu_a in (0, 1):
            plain = run(Ad, Rd, 0, relu_a)
            half_in = run(Ah, Rh, 32 | 3 | (1 << 12), relu_a)             # (bit 12: not the 8-phase kernel, tested below)
            assert torch.equal(plain, half_in), f"M={M} relu {relu_a}: {float((plain - half_in).abs().max()):.3e}"
            half_out = _from_half_rows(run(Ah, Rh, 32 | 7 | (1 << 12), relu_a), N)
            assert torch.equal(half_out, plain.to(torch.bfloat16).float())
            if M >= 65536:        # large launches: the 256 x 256 8-phase kernel takes the full rounds (bias first in the fp32 sum)
                p8_in = run(Ah, Rh, 32 | 3, relu_a)
                assert float((p8_in - plain).abs().max()) <= 2e-5 * float(plain.abs().max())
                p8_out = _from_half_rows(run(Ah, Rh, 32 | 7, relu_a), N)
                d = (p8_out - half_out).abs()
                assert bool((d <= 2.0 ** -7 * half_out.abs() + 2e-5 * float(half_out.abs().max())).all()) and float((d > 0).float().mean()) < 0.01
    tok = [0, 70, 70 + 333]
    T = tok[-1]
    sc = 0.125 * 1.4426950408889634
    q, k, v = (torch.randn(T, 512, generator=g).to(torch.bfloat16).float() for _ in range(3))
    qs = (q * sc).to(torch.bfloat16).float()
    plain = _flash_bf16(lib, (qs / sc).to(DEV), k.to(DEV), v.to(DEV), tok, 0.125, 1, 1)
    got = _from_half_rows(_flash_bf16(lib, _to_half_rows(qs).to(DEV), _to_half_rows(k).to(DEV), _to_half_rows(v).to(DEV), tok, 0.125, 1, 3), 512)
    ref = _ref_attn(qs / sc, k, v, tok, 0.125)
    assert float((got - ref).abs().max()) < 8e-2
    assert float((got - plain).abs().max()) < 3e-2          # (the outputs themselves are rounded to bf16 here)


@pytest.mark.parametrize("N,K,resid,gather,relu_a,act,c_half", [(512, 512, 0, 0, 0, 1, 1), (1024, 256, 0, 0, 1, 0, 1), (512, 1024, 0, 0, 0, 0, 0),
                                                                (256, 512, 1, 0, 0, 1, 0), (1024, 512, 0, 1, 1, 1, 1), (1024, 512, 0, 2, 1, 1, 1),
                                                                (512, 256, 0, 3, 0, 0, 0), (1024, 512, 0, 4, 1, 1, 1)])
def test_gemm_bf16_p8_kernel(lib, N, K, resid, gather, relu_a, act, c_half):
    """Half-row launches with M >= one full round of 256 x 256 tiles: the 8-phase kernel (gemm_bf16_p8.hip) takes the
    full rounds, the remaining row panels the older kernels.  Same k order per accumulator as the 128 x 128 kernel; the bias
    enters first instead of last in the fp32 sum, so results may differ by one rounding of the output format -- checked
    element by element against the launch that keeps everything on the 128 x 128 kernel (fmt bits 4, 12), and against a
    plain fp32 reference."""
    l = lib.load()
    M = (256 * 256 // (N // 256)) + 256 * 9 + 77            # one full round, nine more panels and a ragged end
    g = torch.Generator().manual_seed(N + K + resid)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).float()
    Ah = _to_half_rows(A).to(DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    R = torch.randn(M, N, generator=g).to(torch.bfloat16).float()
    Rh = _to_half_rows(R).to(DEV)
    NG = 777
    gbuf = torch.randn(NG, 2 * N, generator=g).to(DEV)
    gi0 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32)
    gi1 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    # gather = 1: random indices; 2: source-major runs of 39 like the bench batch's edge list; 3: runs of 1..12; 4: runs of 39 with a
    # few out-of-order rows
    if gather >= 2:
        lens = {2: lambda: 39, 3: lambda: int(torch.randint(1, 13, (1,), generator=g)), 4: lambda: 39}[gather]
        rows, node = [], 0
        while len(rows) < M:
            rows += [node % NG] * lens()
            node += 1
        gi0 = torch.tensor(rows[:M], dtype=torch.int32)
        if gather == 4:
            hit = torch.randint(0, M, (M // 50,), generator=g)
            gi0[hit] = torch.randint(0, NG, (len(hit),), generator=g, dtype=torch.int32)
    gi0 = gi0.to(DEV)
    hi = torch.empty(N * K + 128, dtype=torch.int16, device=DEV)
    lo = torch.empty_like(hi)
    lib.check(l.vlsat_k_split_bf16(W.data_ptr(), N * K, hi.data_ptr(), lo.data_ptr(), lib.stream_ptr()))
    fmt = 32 | 1 | (4 if c_half else 0) | (2 if resid else 0)

    def run(f):
        Cb = torch.full((M, N), float("nan"), device=DEV)
        lib.check(l.vlsat_k_gemm_planes(Ah.data_ptr(), K, W.data_ptr(), hi.data_ptr(), lo.data_ptr(), K, Cb.data_ptr(), N, M, N, K,
                                        b.data_ptr(), Rh.data_ptr() if resid else 0, N if resid else 0, 0.5,
                                        gbuf.data_ptr() if gather else 0, gi0.data_ptr() if gather else 0, 2 * N if gather else 0,
                                        gbuf.data_ptr() + 4 * N if gather else 0, gi1.data_ptr() if gather else 0, 2 * N if gather else 0,
                                        relu_a, act, 1, 0, -1, f, 1.0, lib.stream_ptr()))
        _sync()
        return (_from_half_rows(Cb.cpu(), N) if c_half else Cb.cpu())
    got, flat = run(fmt), run(fmt | 16 | (1 << 12))
    assert torch.isfinite(got).all()
    d = (got - flat).abs()
    if c_half:
        assert bool((d <= 2.0 ** -7 * flat.abs() + 2e-5 * float(flat.abs().max())).all()), float(d.max())   # (one bf16 rounding; near a ReLU zero: fp32 roundoff)
        assert float((d > 0).float().mean()) < 0.01
    else:
        assert float(d.max()) <= 2e-5 * float(flat.abs().max())
    Af = torch.relu(A) if relu_a else A
    ref = Af.double() @ W.cpu().to(torch.bfloat16).double().t() + b.cpu().double()
    if resid:
        ref = ref + 0.5 * R.double()
    if gather:
        ref = ref + gbuf.cpu().double()[gi0.cpu().long(), :N] + gbuf.cpu().double()[gi1.cpu().long(), N:]
    if act:
        ref = torch.relu(ref)
    tol = (2.0 ** -7 if c_half else 1e-4) * float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) <= tol


@pytest.mark.parametrize("M", [300, 2560, 256 * 128 + 256 * 9 + 77])
def test_gemm_fp16_half_row_tables(lib, M):
    """The node-side tables of nn_edge.0 as fp16 half rows (round 6; GemmArgs::c_f16_cols / g_f16, fmt bits 25..28 / 3 of
    vlsat_k_gemm_planes; reference network_MMG.py:59-60,92 -- cat[x_i, e, x_j] . W^T as three partial products).
    Producer: a split-bf16 launch on fp32 rows writes its first 512 of 1024 columns as fp16 (element n at byte 2 n of the row), the rest as
    fp32 -- checked against the same launch with an fp32 output, rounded to fp16 on the host (bit for bit) -- on the 64 x 64 / 64 x 128
    kernels, the split-K kernel (M = 300, fmt bit 6) and the ring kernel.  Consumer: a half-row launch that gathers [P_i | P_j] from such
    a table must equal, BIT FOR BIT, the launch that gathers the same (fp16-representable) values from an fp32 table -- 8-phase kernel
    with a ragged last panel, the older kernels (M = 2560, 300)."""
    l = lib.load()
    g = torch.Generator().manual_seed(M)
    K, N = 512, 1024
    hi = torch.empty(N * K + 128, dtype=torch.int16, device=DEV)
    lo = torch.empty_like(hi)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    lib.check(l.vlsat_k_split_bf16(W.data_ptr(), N * K, hi.data_ptr(), lo.data_ptr(), lib.stream_ptr()))
    # ---- producer ----
    Mp = min(M, 2560)
    X = torch.randn(Mp, K, generator=g).to(DEV)

    def produce(f):
        C = torch.full((Mp, N), float("nan"), device=DEV)
        lib.check(l.vlsat_k_gemm_planes(X.data_ptr(), K, W.data_ptr(), hi.data_ptr(), lo.data_ptr(), K, C.data_ptr(), N, Mp, N, K, b.data_ptr(),
                                        0, 0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 0, -1, f, 1.0, lib.stream_ptr()))
        _sync()
        return C.cpu()
    for extra in (0, 64, 16):                                  # default cascade / small launches on the split-K kernel / no ring kernel
        plain, mixed = produce(extra), produce(extra | (2 << 25))
        as16 = mixed.view(torch.float16).view(Mp, 2 * N)
        assert torch.equal(as16[:, :512], plain[:, :512].clamp(-65504, 65504).half()), extra
        assert torch.equal(mixed[:, 512:], plain[:, 512:]), extra
        assert bool(torch.isnan(mixed[:, 256:512]).all())       # bytes 1024..2047 of a row belong to nobody
    # ---- consumer ----
    NG, Nc = 777, 512
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).float()
    Ah = _to_half_rows(A).to(DEV)
    tab16 = (torch.randn(NG, 2 * Nc, generator=g) * 3).half()
    tab32 = tab16.float().to(DEV)                               # [P_i | P_j] as fp32, pitch 2 Nc floats
    buf = torch.zeros(NG, 2 * Nc + 64)                          # the same values as half rows of a row with another pitch
    buf.view(torch.float16).view(NG, 2 * (2 * Nc + 64))[:, :2 * Nc] = tab16
    buf = buf.to(DEV)
    rows, node = [], 0
    while len(rows) < M:
        rows += [node % NG] * 39
        node += 1
    gi0 = torch.tensor(rows[:M], dtype=torch.int32).to(DEV)
    gi1 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    hic = hi[:Nc * K + 128]
    Wc = W[:Nc].contiguous()
    lib.check(l.vlsat_k_split_bf16(Wc.data_ptr(), Nc * K, hic.data_ptr(), lo.data_ptr(), lib.stream_ptr()))

    def consume(f16, f):
        C = torch.full((M, Nc), float("nan"), device=DEV)
        t, ld, off = (buf, 2 * Nc + 64, 2 * Nc) if f16 else (tab32, 2 * Nc, 4 * Nc)     # P_j: Nc halfs / Nc floats into the row
        lib.check(l.vlsat_k_gemm_planes(Ah.data_ptr(), K, Wc.data_ptr(), hic.data_ptr(), lo.data_ptr(), K, C.data_ptr(), Nc, M, Nc, K, 0,
                                        0, 0, 0.0, t.data_ptr(), gi0.data_ptr(), ld, t.data_ptr() + off, gi1.data_ptr(), ld,
                                        1, 1, 1, 0, -1, f | (8 if f16 else 0), 1.0, lib.stream_ptr()))
        _sync()
        return _from_half_rows(C.cpu(), Nc)
    base = 32 | 1 | 4
    for extra in (0, 16 | (1 << 12)):                           # the shipped cascade (8-phase kernel for the large M) / the older kernels only
        a, c = consume(True, base | extra), consume(False, base | extra)
        assert torch.isfinite(a).all() and torch.equal(a, c), extra
    ref = torch.relu(torch.relu(A).double() @ Wc.cpu().to(torch.bfloat16).double().t() + tab16.double()[gi0.cpu().long(), :Nc] + tab16.double()[gi1.cpu().long(), Nc:])
    assert float((a.double() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    # ---- the whole output as fp16 half rows (the out-projection of the single-rounded edge attention; 8-phase kernel CF = 3 for the large M) ----
    def whole(f):
        C = torch.full((M, Nc), float("nan"), device=DEV)
        lib.check(l.vlsat_k_gemm_planes(Ah.data_ptr(), K, Wc.data_ptr(), hic.data_ptr(), lo.data_ptr(), K, C.data_ptr(), Nc, M, Nc, K, b.data_ptr(),
                                        0, 0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, -1, f, 1.0, lib.stream_ptr()))
        _sync()
        return C.cpu()
    for extra in (0, 16 | (1 << 12)):
        f32out, f16out = whole(32 | 1 | extra), whole(32 | 1 | extra | ((Nc // 256) << 25))
        got16 = f16out.view(torch.float16).view(M, 2 * Nc)[:, :Nc].float()
        assert bool(torch.isnan(f16out[:, Nc // 2:]).all())        # the upper half of every row is not written
        d = (got16 - f32out).abs()
        assert bool((d <= 2.0 ** -11 * f32out.abs() + 1e-7).all()), float(d.max())       # one fp16 rounding of the fp32 launch's value
        assert torch.equal(got16, f32out.half().float()) or float((got16 != f32out.half().float()).float().mean()) < 1e-3
    # misuse is refused: fp16 tables next to a residual
    C = torch.empty(M, Nc, device=DEV)
    r = l.vlsat_k_gemm_planes(Ah.data_ptr(), K, Wc.data_ptr(), hic.data_ptr(), lo.data_ptr(), K, C.data_ptr(), Nc, M, Nc, K, 0, C.data_ptr(), Nc, 1.0,
                              buf.data_ptr(), gi0.data_ptr(), 2 * Nc + 64, buf.data_ptr() + 2 * Nc, gi1.data_ptr(), 2 * Nc + 64, 0, 0, 1, 0, -1, base | 8, 1.0,
                              lib.stream_ptr())
    assert r != 0


@pytest.mark.parametrize("tail", [5, 90])          # remainder panels: 5 -> small kernels; 90 (+ 33 rows) -> a partial 8-phase round WITH the ragged last panel
@pytest.mark.parametrize("N,K,resid,gather,relu_a,act", [(512, 512, 0, 0, 0, 1), (1024, 128, 0, 0, 1, 0), (512, 1024, 1, 0, 0, 0), (1024, 512, 0, 1, 0, 1)])
def test_gemm_f32_p8_kernel_is_bit_identical(lib, N, K, resid, gather, relu_a, act, tail):
    """Exact-fp32 launches with M >= one full round of 256 x 256 tiles take the 8-phase kernel for the full rounds.  Same k
    order per accumulator, same epilogue order (accumulator init, bias, activation) as the 128 x 128 kernel: the results must
    be BIT-IDENTICAL to the launch that stays off it (relu_a bit 3), with every additive operand."""
    l = lib.load()
    M = (256 * 256 // (N // 256)) + 256 * tail + 33
    g = torch.Generator().manual_seed(N * 3 + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    R = torch.randn(M, N, generator=g).to(DEV)
    NG = 555
    gbuf = torch.randn(NG, 2 * N, generator=g).to(DEV)
    gi0 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    gi1 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)

    def run(flags):
        Cb = torch.full((M, N), float("nan"), device=DEV)
        lib.check(l.vlsat_k_gemm(A.data_ptr(), K, W.data_ptr(), K, Cb.data_ptr(), N, M, N, K, b.data_ptr(), 0,
                                 R.data_ptr() if resid else 0, N if resid else 0, 0.5,
                                 gbuf.data_ptr() if gather else 0, gi0.data_ptr() if gather else 0, 2 * N if gather else 0,
                                 gbuf.data_ptr() + 4 * N if gather else 0, gi1.data_ptr() if gather else 0, 2 * N if gather else 0,
                                 relu_a | flags, act, lib.stream_ptr()))
        _sync()
        return Cb.cpu()
    got, flat = run(0), run(8)
    assert torch.isfinite(got).all()
    assert torch.equal(got, flat), f"{int((got != flat).sum())} elements differ, max {float((got - flat).abs().max()):.3e}"
    kw = dict(bias=b, act=act)
    if resid:
        kw.update(resid=R, resid_scale=0.5)
    if gather:
        kw.update(g0=gbuf[:, :N], gi0=gi0, g1=gbuf[:, N:], gi1=gi1)
    ref = _ref_gemm(torch.relu(A) if relu_a else A, W, **kw)
    assert float((got - ref.cpu()).abs().max()) < 2e-4 * math.sqrt(K / 64)


@pytest.mark.parametrize("tail", [5, 90])          # remainder panels: 5 -> small kernels; 90 (+ 33 rows) -> a partial 8-phase round WITH the ragged last panel
@pytest.mark.parametrize("N,K,resid,gather,relu_a,act,c_pairs", [(512, 512, 0, 0, 0, 1, 1), (1024, 128, 0, 0, 1, 0, 1), (512, 1024, 0, 0, 0, 0, 0),
                                                                 (1024, 512, 0, 1, 1, 1, 1)])
def test_gemm_bf16x3_p8_kernel_is_bit_identical(lib, N, K, resid, gather, relu_a, act, c_pairs, tail):
    """Split-bf16 launches on split-pair operands with M >= one full round of 256 x 256 tiles: the 8-phase kernel takes the
    full rounds.  Same term order per accumulator (w_hi.a_lo, w_lo.a_hi, w_hi.a_hi per 16 k) and the same epilogue order as
    the 128 x 128 kernel: BIT-IDENTICAL to the launch that stays off it (fmt bits 4, 12); and close to fp64."""
    l = lib.load()
    M = (256 * 256 // (N // 256)) + 256 * tail + 33
    g = torch.Generator().manual_seed(N * 5 + K)
    A = torch.randn(M, K, generator=g)
    Ap = _pack_split(A).to(DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    NG = 555
    gbuf = torch.randn(NG, 2 * N, generator=g).to(DEV)
    gi0 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    gi1 = torch.randint(0, NG, (M,), generator=g, dtype=torch.int32).to(DEV)
    hi = torch.empty(N * K + 128, dtype=torch.int16, device=DEV)
    lo = torch.empty_like(hi)
    lib.check(l.vlsat_k_split_bf16(W.data_ptr(), N * K, hi.data_ptr(), lo.data_ptr(), lib.stream_ptr()))
    fmt = 1 | (4 if c_pairs else 0)

    def run(f):
        Cb = torch.full((M, N), float("nan"), device=DEV)
        lib.check(l.vlsat_k_gemm_planes(Ap.data_ptr(), K, W.data_ptr(), hi.data_ptr(), lo.data_ptr(), K, Cb.data_ptr(), N, M, N, K,
                                        b.data_ptr(), 0, 0, 1.0,
                                        gbuf.data_ptr() if gather else 0, gi0.data_ptr() if gather else 0, 2 * N if gather else 0,
                                        gbuf.data_ptr() + 4 * N if gather else 0, gi1.data_ptr() if gather else 0, 2 * N if gather else 0,
                                        relu_a, act, 3, 0, -1, f, 1.0, lib.stream_ptr()))
        _sync()
        return Cb.cpu()
    got, flat = run(fmt), run(fmt | 16 | (1 << 12))
    assert torch.equal(got.view(torch.int32), flat.view(torch.int32)), f"{int((got.view(torch.int32) != flat.view(torch.int32)).sum())} words differ"
    val = _unpack_split(got) if c_pairs else got
    assert torch.isfinite(val).all()
    kw = dict(bias=b, act=act)
    if gather:
        kw.update(g0=gbuf[:, :N], gi0=gi0, g1=gbuf[:, N:], gi1=gi1)
    Af = _unpack_split(Ap.cpu())
    ref = _ref_gemm(torch.relu(Af) if relu_a else Af, W, **kw)
    assert float((val - ref).abs().max()) < 1e-4 * math.sqrt(K / 64) + 2e-5


# ---- split-K kernel of the small launches (gemm_splitk.hip) ---------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(9, 512, 512), (80, 3328, 512), (80, 512, 768), (40, 160, 512), (72, 26, 256), (600, 512, 1024),
                                   (1500, 1024, 512), (130, 1536, 128), (7, 40, 64)])
def test_gemm_splitk_matches_persistent_kernel(lib, M, N, K):
    """Small launches cut their k range over several CUs and reduce in the kernel.  Against the persistent kernel (other
    summation order: fp32 roundoff only), with every epilogue operand, and twice for run-to-run bit equality (the
    reduction order is fixed, whichever block arrives last)."""
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    rows = torch.rand(M, generator=g).to(DEV) + 0.5
    G0 = torch.randn(50, 2 * N, generator=g).to(DEV)
    gi0 = torch.randint(0, 50, (M,), generator=g).to(torch.int32).to(DEV)
    gi1 = torch.randint(0, 50, (M,), generator=g).to(torch.int32).to(DEV)
    cases = [dict(bias=bias), dict(bias=bias, resid=resid, resid_scale=0.5, act=1), dict(bias=bias, rowscale=rows),
             dict(g0=G0, gi0=gi0, g1=G0[:, N:], gi1=gi1, act=1, relu_a=1), dict(bias=bias, act=2, relu_a=1)]
    for kw in cases:
        ref = _ref_gemm(A, W, **kw)
        plain = _gemm(lib, A, W, **kw)
        kw4 = dict(kw, relu_a=kw.get("relu_a", 0) | 4)
        got = _gemm(lib, A, W, **kw4)
        again = _gemm(lib, A, W, **kw4)
        assert torch.equal(got, again), "split-K result changed between two identical launches"
        assert float((got - ref).abs().max()) < 2e-4 and float((got - plain).abs().max()) < 1e-4, kw.keys()


@pytest.mark.parametrize("prec,fmt", [(3, 0), (1, 0), (3, 5), (1, 5), (1, 37)])
def test_gemm_splitk_bf16_modes(lib, prec, fmt):
    """The same in the bf16 modes (fp32, split-pair and half-row operands) against the persistent kernel."""
    l = lib.load()
    g = torch.Generator().manual_seed(99 + fmt)
    for M, N, K in ((80, 512, 512), (600, 1024, 512), (300, 512, 1024)):
        A = torch.randn(M, K, generator=g)
        if fmt & 32:
            A = A.to(torch.bfloat16).float()
        W = (torch.randn(N, K, generator=g) / 16).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        hi = torch.empty(N * K + 128, dtype=torch.int16, device=DEV)
        lo = torch.empty_like(hi)
        lib.check(l.vlsat_k_split_bf16(W.data_ptr(), N * K, hi.data_ptr(), lo.data_ptr(), lib.stream_ptr()))
        if fmt & 32:
            Ad = _to_half_rows(A).to(DEV)
        elif fmt & 1:
            Ad = _pack_split(A).to(DEV)
        else:
            Ad = A.to(DEV)

        def run(extra):
            Cb = torch.zeros(M, N, device=DEV)
            lib.check(l.vlsat_k_gemm_planes(Ad.data_ptr(), K, W.data_ptr(), hi.data_ptr(), lo.data_ptr(), K, Cb.data_ptr(), N, M, N, K,
                                            b.data_ptr(), 0, 0, 1.0, 0, 0, 0, 0, 0, 0, 1, 1, prec, 0, -1, fmt | extra, 1.0,
                                            lib.stream_ptr()))
            _sync()
            return Cb.cpu()
        plain, split = run(0), run(64)
        if fmt & 4:
            unpack = (lambda t: _from_half_rows(t, N)) if fmt & 32 else _unpack_split
            plain, split = unpack(plain), unpack(split)
        tol = 2e-2 if fmt & 32 else 1e-4
        assert float((plain - split).abs().max()) < tol, (M, N, K, float((plain - split).abs().max()))
        assert torch.equal(split, (lambda t: (_from_half_rows(t, N) if fmt & 32 else _unpack_split(t)) if fmt & 4 else t)(run(64)))
