"""Diagnose test_growing_dead_set_switches_to_the_dense_route_in_time: which elements differ from the oracle, their gradient
magnitude and whether the row's selection differs."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import sae_ref as R
from test_gpu_parity import make_engine, rand_params
d, s, k, n, k_aux, thr = 128, 1024, 8, 200, 64, 100_000
p = rand_params(d, s, seed=290)
gen = torch.Generator().manual_seed(291)
cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k, k_aux=k_aux, dead_threshold_tokens=thr)
toks = torch.zeros(s, dtype=torch.int64)
late = torch.randperm(s, generator=torch.Generator().manual_seed(292))[:80]
toks[late] = thr - 6 * n
p["b_enc"][late] = -100.0
eng = make_engine(d, s, k, k_aux=k_aux, thr=thr, max_batch=n)
eng.load_params(p); eng.set_tracker(toks)
lateset = set(late.tolist())
for i in range(9):
    x = torch.randn(n, d, generator=gen)
    state = R.TrainState(params={k_: v.cpu().clone() for k_, v in eng.param_views().items()},
        m={k_: eng.view(k_, eng.adam_m).cpu().clone() for k_ in R.PARAM_ORDER},
        v={k_: eng.view(k_, eng.adam_v).cpu().clone() for k_ in R.PARAM_ORDER},
        toks_since_active=eng.toks_since_active.cpu().clone(), adam_steps=eng.adam_steps, lr=1e-3)
    v_before = {k_: state.v[k_].clone() for k_ in R.PARAM_ORDER}
    ref = R.train_step(state, x, cfg)
    eng.train_step(x.cuda(), 1e-3, 1.0)
    st = eng.read_stats()
    print(i, "route", eng.aux_route(), f"n_dead {st.n_dead} aux {st.aux:.6g}/{ref['aux']:.6g} gn {st.grad_norm:.6f}/{ref['grad_norm']:.6f}")
    for key in R.PARAM_ORDER:
        a, b = eng.view(key).cpu(), state.params[key]
        bad = ~torch.isclose(a, b, rtol=1e-4, atol=2e-6)
        if bad.any():
            g = ref["grads"][key]
            ma, mb = eng.view(key, eng.adam_m).cpu(), state.m[key]
            for ix in bad.nonzero().tolist()[:12]:
                t = tuple(ix)
                row = t[0] if key == "W_dec" else t[-1]
                print(f"   {key}{t}: hip {a[t]:.8g} ref {b[t]:.8g} d {a[t]-b[t]:.3g} | ref grad {g[t]:.4g} v_before {v_before[key][t]:.4g} m hip {ma[t]:.6g} ref {mb[t]:.6g} | dead row {row in lateset}")
