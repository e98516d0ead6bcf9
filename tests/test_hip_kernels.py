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
