import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
from test_gpu_parity import make_engine, rand_params
d, s, n, k = 256, 4096, 512, 32
for tiny in (1e-5, 1e-8, 1e-9, 3e-10):
    g = torch.Generator().manual_seed(31)
    p = rand_params(d, s, seed=32); p["b_enc"] = torch.zeros(s)
    big = torch.randn(224, d, generator=g); small = tiny * torch.randn(32, d, generator=g)
    x = torch.cat([big, -big, small, -small], dim=0).contiguous()
    out = {}
    for mode in ("f32", "f16r"):
        eng = make_engine(d, s, k, k_aux=0, max_batch=n, encoder=mode); eng.load_params(p)
        eng.step_forward(x.cuda(), training=False)
        idx, val, _ = eng.last_codes(n); st = eng.read_stats()
        out[mode] = idx.cpu(); print(mode, "dense_route", st.dense_route, "cand_max", st.cand_max, "overflow rows", st.n_overflow_rows)
    h = x.double() @ p["W_enc"].double()
    ex = h.topk(k, dim=1).indices.sort(dim=1).values
    for mode in out:
        got = out[mode].long().sort(dim=1).values
        print(tiny, mode, "tiny rows differing from fp64 top-k:", int((got[448:] != ex[448:]).any(dim=1).sum()), "big rows:", int((got[:448] != ex[:448]).any(dim=1).sum()))
