"""debug probe: half-row edge attention through the C ABI on small and > 4 GiB tensors"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import vlsat_amd  # noqa
from vlsat_amd import lib as L
DEV = "cuda:0"
l = L.load()
S = 512
sc = 0.125 * 1.4426950408889634
for T in (4 * S, (1 << 21)):
    rows = T + S
    g = torch.Generator(device=DEV).manual_seed(11)
    def half_rows(scale=1.0):
        x = torch.zeros(rows, 512, dtype=torch.float32, device=DEV)
        v = (torch.randn(rows, 512, generator=g, device=DEV, dtype=torch.float32) * scale).to(torch.bfloat16)
        x.view(torch.bfloat16).view(rows, 1024)[:, :512] = v
        return x, v
    q, qv = half_rows(sc)
    k, kv = half_rows()
    v, vv = half_rows()
    o = torch.zeros(rows, 512, dtype=torch.float32, device=DEV)
    tok = torch.arange(0, rows + 1, S, dtype=torch.int64)
    L.check(l.vlsat_k_flash_attn_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), 512, tok.data_ptr(), len(tok) - 1, 8, 0.125, 1, 3, L.stream_ptr()))
    torch.cuda.synchronize()
    got = o.view(torch.bfloat16).view(rows, 1024)[:, :512].float()
    for a in (0, rows - S):
        qq = (qv[a:a + S].double() / sc).view(S, 8, 64).permute(1, 0, 2)
        kk = kv[a:a + S].double().view(S, 8, 64).permute(1, 2, 0)
        vh = vv[a:a + S].double().view(S, 8, 64).permute(1, 0, 2)
        ref = (torch.softmax(qq @ kk * 0.125, -1) @ vh).permute(1, 0, 2).reshape(S, 512).float()
        print(T, a, "err", float((got[a:a + S] - ref).abs().max()), "|got|", float(got[a:a + S].abs().max()), "|ref|", float(ref.abs().max()))
