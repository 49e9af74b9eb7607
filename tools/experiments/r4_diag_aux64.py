import sys, math, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import sae_ref as R
from test_gpu_parity import make_engine, rand_params
n_dead = int(sys.argv[1]); asm = int(sys.argv[2]); base = int(sys.argv[3]) if len(sys.argv) > 3 else 80
d, s, k, n, k_aux, thr = 128, 1024, 8, 200, 64, 100_000
p = rand_params(d, s, seed=base + n_dead)
gen = torch.Generator().manual_seed(base + 1 + n_dead)
cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k, k_aux=k_aux, dead_threshold_tokens=thr)
toks = torch.zeros(s, dtype=torch.int64)
dead = torch.randperm(s, generator=torch.Generator().manual_seed(base + 2))[:n_dead]
toks[dead] = thr
p["b_enc"][dead] = -100.0
eng = make_engine(d, s, k, k_aux=k_aux, thr=thr, max_batch=n, aux_small_max=asm)
eng.load_params(p); eng.set_tracker(toks)
for i in range(9):
    x = torch.randn(n, d, generator=gen)
    state = R.TrainState(params={k_: v.cpu().clone() for k_, v in eng.param_views().items()},
        m={k_: eng.view(k_, eng.adam_m).cpu().clone() for k_ in R.PARAM_ORDER},
        v={k_: eng.view(k_, eng.adam_v).cpu().clone() for k_ in R.PARAM_ORDER},
        toks_since_active=eng.toks_since_active.cpu().clone(), adam_steps=eng.adam_steps, lr=1e-3)
    ref = R.train_step(state, x, cfg)
    eng.train_step(x.cuda(), 1e-3, 1.0)
    st = eng.read_stats()
    out = []
    for key in ("W_dec", "W_enc"):
        a, b = eng.view(key).cpu(), state.params[key]
        bad = ~torch.isclose(a, b, rtol=1e-4, atol=2e-6)
        rows = bad.any(dim=1 if key == "W_dec" else 0).nonzero().flatten()
        isdead = [int(r) in set(dead.tolist()) for r in rows.tolist()]
        out.append(f"{key}: {int(bad.sum())} off, max|d| {(a-b).abs().max():.2e}, rows {len(rows)} (dead among them {sum(isdead)})")
    print(i, "route", eng.aux_route(), f"aux {st.aux:.6f}/{ref['aux']:.6f} gn {st.grad_norm:.6f}/{ref['grad_norm']:.6f}", *out)
