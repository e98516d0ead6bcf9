"""Kernel-level parity of the non-GEMM kernels of the GCN block and the node attention, through the C ABI
(vlsat_k_edge_gate / vlsat_k_aggregate / vlsat_k_node_attn / vlsat_k_dist_bias) against the CPU oracle's
``edge_atten`` / ``aggre_index`` / ``mha`` / ``distance_bias`` (fp64) on random graphs -- every head geometry
MODEL.NUM_HEADS in {4, 8, 16} x MODEL.DIM_ATTEN in {128, 256, 512} builds, the three aggregators, both node-attention
kernels.  A parity failure in one gate geometry shows up here as that geometry, not as a whole-forward difference.
Reference: network_MMG.py:84-112,165-173,190-203; network_util.py:64-73; transformer/attention.py:41-78.
Needs an MI355X:  pytest -m gpu"""
import math

import pytest
import torch

import vlsat_amd  # noqa: F401
from oracle import vlsat_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F64 = torch.float64


@pytest.fixture(scope="module")
def lib():
    from vlsat_amd import lib as L
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path cannot run and there is no fallback")
    return L


def _dev(t):
    return t.to(torch.float32).contiguous().to(DEV)


def _random_edges(g, n, e, kind):
    """[2,E] int64: 'fc' = source-major fully connected without self loops (the reference's list,
    dataset_3dssg.py:264-266), 'random' = arbitrary pairs with duplicates and self loops, some nodes without out-edges."""
    if kind == "fc":
        s, d = torch.meshgrid(torch.arange(n), torch.arange(n), indexing="ij")
        keep = s != d
        return torch.stack([s[keep], d[keep]])
    src = torch.randint(0, max(1, n - 3), (e,), generator=g)      # the last three nodes never appear as a source
    dst = torch.randint(0, n, (e,), generator=g)
    return torch.stack([src, dst])


def _gate_weights(g, H, A, use_edge=True):
    """random weights of one MultiHeadedEdgeAttention (network_MMG.py:55-79) in the reference's shapes; nn_edge is shrunk
    (its output is not under test here)"""
    dk, dox = 512 // H, A // H
    cin = 2 * dk if use_edge else dk            # MLP([d_n+d_e, d_n+d_e, d_o]) or MLP([d_n, 2 d_n, d_o]) (:72-75)
    p = "g.edgeatten."

    def rnd(*shape, s):
        return torch.randn(*shape, generator=g, dtype=F64) * s
    return {p + "nn_edge.0.weight": rnd(8, 1536, s=0.02), p + "nn_edge.0.bias": rnd(8, s=0.1),
            p + "nn_edge.2.weight": rnd(512, 8, s=0.1), p + "nn_edge.2.bias": rnd(512, s=0.1),
            p + "proj_query.0.weight": rnd(512, 512, s=1 / math.sqrt(512)), p + "proj_query.0.bias": rnd(512, s=0.1),
            p + "proj_edge.0.weight": rnd(512, 512, s=1 / math.sqrt(512)), p + "proj_edge.0.bias": rnd(512, s=0.1),
            p + "proj_value.0.weight": rnd(A, 512, s=1 / math.sqrt(512)), p + "proj_value.0.bias": rnd(A, s=0.1),
            p + "nn.0.weight": rnd(2 * dk, cin, 1, s=1.5 / math.sqrt(cin)), p + "nn.0.bias": rnd(2 * dk, s=0.2),
            p + "nn.3.weight": rnd(dox, 2 * dk, 1, s=2.0 / math.sqrt(2 * dk)), p + "nn.3.bias": rnd(dox, s=0.2)}


GEOMS = [(4, 128), (4, 256), (4, 512), (8, 128), (8, 256), (8, 512), (16, 128), (16, 256), (16, 512)]
# (variant, tolerance relative to max(1, max |reference|)): fp32 kernels, split-bf16, single-rounded bf16
VARIANTS = {"fp32": (0, 2e-5), "valu": (1, 2e-5), "mfma": (2, 2e-5), "bf16x3": (3, 3e-4), "bf16": (4, 4e-2)}


def _gate_case(L, H, A, vname, kind, use_edge=True, seed=0, n=23, e=777):
    """Prepared operands exactly as DESIGN.md section 2 / include/vlsat.h describe them, built from reference-layout
    tensors in fp64 and rounded once to fp32; outputs compared in the REFERENCE layouts with oracle.edge_atten."""
    variant, tol = VARIANTS[vname]
    g = torch.Generator().manual_seed(1000 * H + A + seed)
    w = _gate_weights(g, H, A, use_edge)
    ei = _random_edges(g, n, e, kind)
    E = ei.shape[1]
    x = torch.randn(n, 512, generator=g, dtype=F64)
    ed = torch.randn(E, 512, generator=g, dtype=F64)
    ref_gated, _, ref_prob = O.edge_atten(x, ed, ei, w, "g", H)
    dk, dox = 512 // H, A // H
    p = "g.edgeatten."
    q = O.lin(x, w, p + "proj_query.0").view(n, dk, H)
    k = O.lin(ed, w, p + "proj_edge.0").view(E, dk, H).permute(0, 2, 1).reshape(E, H * dk)          # head-major
    v = O.lin(x, w, p + "proj_value.0").view(n, dox, H).permute(0, 2, 1).reshape(n, H * dox)        # head-major
    W0, b0 = w[p + "nn.0.weight"][:, :, 0], w[p + "nn.0.bias"]
    W3, b3 = w[p + "nn.3.weight"][:, :, 0], w[p + "nn.3.bias"]
    gq = (torch.einsum("oc,nch->nho", W0[:, :dk], q) + b0[None, None, :]).reshape(n, H * 2 * dk)    # query half of nn.0, per node
    w0k = W0[:, dk:] if use_edge else torch.zeros(2 * dk, dk, dtype=F64)
    node = torch.cat([gq, v, torch.zeros(n, 4, dtype=F64)], 1)          # [Gq | value | pad]: the pitch is not the width
    ld_node, gq_off, v_off = node.shape[1], 0, H * 2 * dk
    d_node, d_k = _dev(node), _dev(k)
    d_src, d_dst = ei[0].to(torch.int32).to(DEV), ei[1].to(torch.int32).to(DEV)
    d_w0k, d_w3, d_b3 = _dev(w0k), _dev(W3), _dev(b3)
    gated = torch.full((E, A), float("nan"), device=DEV)
    prob = torch.full((E, dox, H), float("nan"), device=DEV)
    l = L.load()
    L.check(l.vlsat_k_edge_gate(d_k.data_ptr(), d_node.data_ptr(), ld_node, gq_off, v_off, d_src.data_ptr(), d_dst.data_ptr(),
                                d_w0k.data_ptr(), d_w3.data_ptr(), d_b3.data_ptr(), gated.data_ptr(), prob.data_ptr(), E, H, dk, dox,
                                1 if use_edge else 0, variant, L.stream_ptr()))
    torch.cuda.synchronize()
    got = gated.cpu().double().view(E, H, dox).permute(0, 2, 1).reshape(E, A)      # head-major -> the reference's m*H + h
    scale = max(1.0, float(ref_gated.abs().max()))
    err = float((got - ref_gated).abs().max()) / scale
    gp = prob.cpu().double()
    perr = float((gp - ref_prob).abs().max())
    assert torch.isfinite(got).all() and err < tol and perr < tol, f"H={H} A={A} {vname} {kind}: gated {err:.2e} prob {perr:.2e} (tol {tol})"
    # the probabilities of every (edge, head) sum to one: the softmax ran over the d_o channels, not over the heads
    assert float((gp.sum(1) - 1).abs().max()) < max(tol, 1e-5)


@pytest.mark.parametrize("H,A", GEOMS)
@pytest.mark.parametrize("vname", ["fp32", "valu", "mfma"])
def test_edge_gate_fp32_vs_oracle_all_head_geometries(lib, H, A, vname):
    _gate_case(lib, H, A, vname, "random")


@pytest.mark.parametrize("H,A", GEOMS)
@pytest.mark.parametrize("vname", ["bf16x3", "bf16"])
def test_edge_gate_bf16_vs_oracle_all_head_geometries(lib, H, A, vname):
    if vname == "bf16x3" and H == 4:
        with pytest.raises(lib.VlsatError):          # two plane sets do not fit the LDS at d_k = 128: refused, not silently wrong
            _gate_case(lib, H, A, vname, "random")
        return
    _gate_case(lib, H, A, vname, "random")


@pytest.mark.parametrize("H,A", [(8, 256), (4, 512), (16, 128)])
def test_edge_gate_fully_connected_source_major_list(lib, H, A):
    """the reference's own edge list (fully connected, source-major): a wave's 32 rows share one or two source rows"""
    _gate_case(lib, H, A, "fp32", "fc", n=17)


@pytest.mark.parametrize("H,A", [(8, 256), (16, 256)])
@pytest.mark.parametrize("vname", ["fp32", "valu", "bf16x3"])
def test_edge_gate_without_edge_features(lib, H, A, vname):
    """MODEL.USE_GCN_EDGE = false (network_MMG.py:72-75,99-102): the gate MLP sees the projected query alone"""
    _gate_case(lib, H, A, vname, "random", use_edge=False, seed=5)


def test_edge_gate_ragged_sizes(lib):
    """edge counts that are not multiples of the kernels' 32-edge work units, down to one edge"""
    for e in (1, 31, 33, 65):
        _gate_case(lib, 8, 256, "fp32", "random", seed=e, n=9, e=e)
        _gate_case(lib, 8, 256, "bf16x3", "random", seed=e, n=9, e=e)


# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("aggr", ["max", "add", "mean"])
@pytest.mark.parametrize("n_ch", [128, 256, 512])
def test_aggregate_vs_oracle(lib, aggr, n_ch):
    """Aggre_Index (network_util.py:64-73) on a random index with duplicates and EMPTY segments (-> 0), into a strided
    output at a column offset (how the forward writes cat([x, agg]))."""
    g = torch.Generator().manual_seed(n_ch + len(aggr))
    n, e = 37, 1501
    idx = torch.randint(0, n - 5, (e,), generator=g)
    idx[idx == 7] = 8                                           # node 7: an empty segment in the middle as well
    x = torch.randn(e, n_ch, generator=g, dtype=F64)
    ei = torch.stack([idx, torch.zeros_like(idx)])
    ref = O.aggre_index(x, ei, n, aggr, "target_to_source")
    ldo, col0 = 512 + n_ch, 512
    out = torch.full((n, ldo), 7.0, device=DEV)
    d_x = _dev(x)
    l = lib.load()
    lib.check(l.vlsat_k_aggregate(d_x.data_ptr(), n_ch, idx.contiguous().data_ptr(), e, n, {"max": 0, "add": 1, "mean": 2}[aggr],
                                  out.data_ptr(), ldo, col0, lib.stream_ptr()))
    got = out.cpu().double()
    assert torch.equal(got[:, :col0], torch.full((n, col0), 7.0, dtype=F64)), "columns outside [col0, col0 + n_ch) were touched"
    ref32 = O.aggre_index(x.float(), ei, n, aggr, "target_to_source")
    if aggr == "max":
        assert torch.equal(got[:, col0:].float(), ref32), "max must be exact"
    err = float((got[:, col0:] - ref).abs().max())
    assert err < 1e-4, f"{aggr} {n_ch}: {err:.2e}"
    assert float(got[7, col0:].abs().max()) == 0.0 and float(got[n - 1, col0:].abs().max()) == 0.0, "empty segment must give 0"


def test_aggregate_no_edges(lib):
    out = torch.full((5, 256), 3.0, device=DEV)
    l = lib.load()
    lib.check(l.vlsat_k_aggregate(0, 256, 0, 0, 5, 0, out.data_ptr(), 256, 0, lib.stream_ptr()))
    assert float(out.abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------------
def _attn_weights(g, prefix, H):
    w = {}
    for nm in ("fc_q", "fc_k", "fc_v", "fc_o"):
        w[f"{prefix}.attention.{nm}.weight"] = torch.randn(512, 512, generator=g, dtype=F64) * (2.0 / math.sqrt(512))
        w[f"{prefix}.attention.{nm}.bias"] = torch.randn(512, generator=g, dtype=F64) * 0.1
    w[prefix + ".layer_norm.weight"] = 1 + 0.2 * torch.randn(512, generator=g, dtype=F64)
    w[prefix + ".layer_norm.bias"] = 0.1 * torch.randn(512, generator=g, dtype=F64)
    return w


def _bias_weights(g, H):
    p = "mmg.self_attn_fc."
    return {p + "0.weight": torch.randn(32, 4, generator=g, dtype=F64) * 0.7, p + "0.bias": torch.randn(32, generator=g, dtype=F64) * 0.3,
            p + "2.weight": 1 + 0.3 * torch.randn(32, generator=g, dtype=F64), p + "2.bias": 0.2 * torch.randn(32, generator=g, dtype=F64),
            p + "3.weight": torch.randn(32, 32, generator=g, dtype=F64) * 0.3, p + "3.bias": torch.randn(32, generator=g, dtype=F64) * 0.3,
            p + "5.weight": 1 + 0.3 * torch.randn(32, generator=g, dtype=F64), p + "5.bias": 0.2 * torch.randn(32, generator=g, dtype=F64),
            p + "6.weight": torch.randn(H, 32, generator=g, dtype=F64) * 0.4, p + "6.bias": torch.randn(H, generator=g, dtype=F64) * 0.2}


def _dist_bias_hip(L, desc, node_ptr, H, w):
    p = "mmg.self_attn_fc."
    d = {k: _dev(v) for k, v in w.items() if k.startswith(p)}
    sizes = [int(node_ptr[i + 1] - node_ptr[i]) for i in range(len(node_ptr) - 1)]
    out = torch.full((sum(H * n * n for n in sizes),), float("nan"), device=DEV)
    d_desc = _dev(desc)
    l = L.load()
    L.check(l.vlsat_k_dist_bias(d_desc.data_ptr(), desc.shape[1], node_ptr.data_ptr(), len(sizes), H,
                                d[p + "0.weight"].data_ptr(), d[p + "0.bias"].data_ptr(), d[p + "2.weight"].data_ptr(), d[p + "2.bias"].data_ptr(),
                                d[p + "3.weight"].data_ptr(), d[p + "3.bias"].data_ptr(), d[p + "5.weight"].data_ptr(), d[p + "5.bias"].data_ptr(),
                                d[p + "6.weight"].data_ptr(), d[p + "6.bias"].data_ptr(), out.data_ptr(), L.stream_ptr()))
    return out, sizes


@pytest.mark.parametrize("H", [4, 8, 16])
def test_dist_bias_vs_oracle(lib, H):
    """network_MMG.py:190-203 + self_attn_fc: orientation (key minus query), per-scene blocks, head-major output"""
    g = torch.Generator().manual_seed(H)
    sizes = [1, 9, 40, 3, 65]
    node_ptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int64)
    desc = torch.rand(sum(sizes), 11, generator=g, dtype=F64) * 4
    w = _bias_weights(g, H)
    out, _ = _dist_bias_hip(lib, desc, node_ptr, H, w)
    got = out.cpu().double()
    off = 0
    for s, n in enumerate(sizes):
        lo = int(node_ptr[s])
        ref = O.distance_bias(desc[lo:lo + n, :3], w)                    # [H, n, n]
        blk = got[off:off + H * n * n].view(H, n, n)
        err = float((blk - ref).abs().max())
        assert err < 2e-5, f"H={H} scene {s} (n={n}): {err:.2e}"
        off += H * n * n


@pytest.mark.parametrize("H", [4, 8, 16])
@pytest.mark.parametrize("lanes", [1, 16])
@pytest.mark.parametrize("cross", [False, True])
def test_node_attention_vs_oracle_mha(lib, H, lanes, cross):
    """MultiHeadAttention of the node stages (attention.py:41-126 as called from network_MMG.py:217-218) with the kernel as
    its attention core: projections, out-projection, residual and LayerNorm in plain torch fp64 around it, compared with
    oracle.mha scene by scene -- ragged scenes (1 .. 70 nodes: more than one 64-key chunk), distance bias from the HIP kernel."""
    g = torch.Generator().manual_seed(100 * H + lanes + (7 if cross else 0))
    sizes = [1, 12, 70, 5, 40]
    node_ptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int64)
    N = sum(sizes)
    prefix = "mmg.cross_attn.0" if cross else "mmg.self_attn.0"
    w = _attn_weights(g, prefix, H)
    w.update(_bias_weights(g, H))
    desc = torch.rand(N, 11, generator=g, dtype=F64) * 4
    xq = torch.randn(N, 512, generator=g, dtype=F64)
    xkv = torch.randn(N, 512, generator=g, dtype=F64) if cross else xq
    bias, _ = _dist_bias_hip(lib, desc, node_ptr, H, w)
    p = prefix + ".attention."
    dk = 512 // H
    q, k, v = O.lin(xq, w, p + "fc_q"), O.lin(xkv, w, p + "fc_k"), O.lin(xkv, w, p + "fc_v")
    qkv = _dev(torch.cat([q, k, v], 1))                              # one [N, 1536] buffer like the forward's
    o = torch.full((N, 512), float("nan"), device=DEV)
    l = lib.load()
    lib.check(l.vlsat_k_node_attn(qkv.data_ptr(), 1536, qkv.data_ptr() + 512 * 4, 1536, qkv.data_ptr() + 1024 * 4, 1536, o.data_ptr(), 512,
                                  bias.data_ptr(), node_ptr.data_ptr(), len(sizes), H, 1.0 / math.sqrt(dk), lanes, lib.stream_ptr()))
    att = o.cpu().double()
    assert torch.isfinite(att).all()
    out = torch.nn.functional.layer_norm(xq + O.lin(att, w, p + "fc_o"), (512,), w[prefix + ".layer_norm.weight"], w[prefix + ".layer_norm.bias"], 1e-5)
    for s, n in enumerate(sizes):
        lo = int(node_ptr[s])
        ref = O.mha(xq[lo:lo + n], xkv[lo:lo + n], w, prefix, H, O.distance_bias(desc[lo:lo + n, :3], w))
        err = float((out[lo:lo + n] - ref).abs().max())
        assert err < 5e-5, f"H={H} lanes={lanes} cross={cross} scene {s} (n={n}): {err:.2e}"


def test_node_attention_without_bias_is_plain_softmax_attention(lib):
    g = torch.Generator().manual_seed(3)
    sizes = [33, 64, 7]
    node_ptr = torch.tensor([0, 33, 97, 104], dtype=torch.int64)
    N, H, dk = 104, 8, 64
    q, k, v = (torch.randn(N, 512, generator=g, dtype=F64) for _ in range(3))
    dq, dkk, dv = _dev(q), _dev(k), _dev(v)
    for lanes in (1, 16):
        o = torch.full((N, 512), float("nan"), device=DEV)
        l = lib.load()
        lib.check(l.vlsat_k_node_attn(dq.data_ptr(), 512, dkk.data_ptr(), 512, dv.data_ptr(), 512, o.data_ptr(), 512, 0, node_ptr.data_ptr(), 3, H,
                                      0.125, lanes, lib.stream_ptr()))
        got = o.cpu().double()
        for s, n in enumerate(sizes):
            lo = int(node_ptr[s])
            qs, ks, vs = (t[lo:lo + n].view(n, H, dk).permute(1, 0, 2) for t in (q, k, v))
            ref = (torch.softmax(qs @ ks.transpose(1, 2) * 0.125, -1) @ vs).permute(1, 0, 2).reshape(n, 512)
            assert float((got[lo:lo + n] - ref).abs().max()) < 1e-5
