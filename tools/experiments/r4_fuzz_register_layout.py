"""Random shapes through the register-layout kernels (one teacher-forced train step each) against the oracle."""
import math, random, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import sae_ref as R
from test_gpu_parity import make_engine, rand_params
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    d = random.choice([256, 512, 768, 1024]); k = random.choice([1, 2, 5, 16, 31, 32]); n = random.choice([1, 3, 17, 64, 129, 333])
    s = random.choice([2, 4]) * d; P = random.choice([1, 1, 2, 7, 16]); nd = random.choice([0, 0, 1, 3, 8])
    thr = 1000
    p = rand_params(d, s, seed=trial)
    toks = torch.zeros(s, dtype=torch.int64)
    if nd:
        dead = torch.randperm(s, generator=torch.Generator().manual_seed(trial))[:nd]
        p["b_enc"][dead] = -5.0; toks[dead] = thr
    cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k, k_aux=64 if nd else 0, n_prefixes=P, dead_threshold_tokens=thr, grad_clip=1.0)
    eng = make_engine(d, s, k, k_aux=64 if nd else 0, thr=thr, max_batch=n)
    eng.load_params(p); eng.set_tracker(toks)
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(1000 + trial)) + 0.2
    state = R.TrainState.create({kk: v.clone() for kk, v in p.items()}); state.toks_since_active = toks.clone(); state.lr = 1e-3
    torch.manual_seed(5000 + trial)
    prefixes = R.sample_prefixes(s, P) if P > 1 else None
    torch.manual_seed(5000 + trial)
    ref = R.train_step(state, x, cfg)
    eng.set_prefixes(prefixes)
    eng.train_step(x.cuda(), 1e-3, 1.0)
    st = eng.read_stats()
    ok = math.isclose(st.mse, ref["mse"], rel_tol=1e-4) and math.isclose(st.aux, ref["aux"], rel_tol=1e-3, abs_tol=1e-10) and math.isclose(st.grad_norm, ref["grad_norm"], rel_tol=2e-4)
    worst = 0.0
    for key in R.PARAM_ORDER:
        a, b = eng.view(key).cpu(), state.params[key]
        badm = ~torch.isclose(a, b, rtol=1e-4, atol=2e-6)
        # (one element of a short vector may sit at a gradient of ~1e-8, where Adam's first update g / (|g| + eps) amplifies the last
        # bits of g: seed 0, trial 18 -- b_enc[303], the same on every build; r4_diag_fuzz_case.py)
        if badm.sum().item() > 1: worst = max(worst, badm.float().mean().item())
    ok = ok and worst <= 1e-3
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} d={d} s={s} k={k} n={n} P={P} nd={nd}: mse {st.mse:.6f}/{ref['mse']:.6f} aux {st.aux:.5f}/{ref['aux']:.5f} gn {st.grad_norm:.5f}/{ref['grad_norm']:.5f} off {worst:.1e} route {eng.aux_route()}")
print("bad:", bad)
