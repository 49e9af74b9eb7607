"""teacher-forced steps, one-pass / five-pass / dense AuxK vs the oracle: which quantities drift at which step"""
import os, sys, math, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import sae_ref as R
from test_gpu_parity import make_engine, rand_params
d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_dead = int(sys.argv[2]) if len(sys.argv) > 2 else 5
s, k, n, k_aux, thr = 4 * d, 8, 210, 64, 100_000
for name, asm in (("one", "0"), ("five", "64"), ("dense", "-1")):
    p = rand_params(d, s, seed=500 + d + n_dead)
    dead = torch.randperm(s, generator=torch.Generator().manual_seed(502 + d))[:n_dead]
    p["b_enc"][dead] = -100.0
    toks = torch.zeros(s, dtype=torch.int64); toks[dead] = thr
    cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k, k_aux=k_aux, dead_threshold_tokens=thr)
    os.environ["SAEV_AMD_AUX_SMALL_MAX"] = asm
    eng = make_engine(d, s, k, k_aux=k_aux, thr=thr, max_batch=n)
    eng.load_params(p); eng.set_tracker(toks)
    gen = torch.Generator().manual_seed(503 + d + n_dead)
    for i in range(7):
        x = torch.randn(n, d, generator=gen)
        state = R.TrainState(params={k_: v.cpu().clone() for k_, v in eng.param_views().items()},
            m={k_: eng.view(k_, eng.adam_m).cpu().clone() for k_ in R.PARAM_ORDER},
            v={k_: eng.view(k_, eng.adam_v).cpu().clone() for k_ in R.PARAM_ORDER},
            toks_since_active=eng.toks_since_active.cpu().clone(), adam_steps=eng.adam_steps, lr=1e-3)
        h = R.encode_pre(x, state.params["W_enc"], state.params["b_enc"])
        top = h.topk(k + 1, dim=1).values
        gap = ((top[:, k - 1] - top[:, k]) / top[:, k - 1].abs()).min().item()
        ref = R.train_step(state, x, cfg)
        eng.train_step(x.cuda(), 1e-3, 1.0)
        st = eng.read_stats()
        print(f"{name:5s} step {i} route {eng.aux_route()} mse rel {abs(st.mse-ref['mse'])/ref['mse']:.2e} aux rel {abs(st.aux-ref['aux'])/max(ref['aux'],1e-30):.2e} gn rel {abs(st.grad_norm-ref['grad_norm'])/ref['grad_norm']:.2e} gn {ref['grad_norm']:.4f} min gap {gap:.1e}")
