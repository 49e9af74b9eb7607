"""one-pass vs five-pass AuxK vs the oracle: gradient differences per parameter (d = 1024, 5 dead latents)."""
import os, sys, math, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import sae_ref as R
from test_gpu_parity import make_engine, rand_params
d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_dead = int(sys.argv[2]) if len(sys.argv) > 2 else 5
s, k, n, k_aux, thr = 4 * d, 8, 210, 64, 100_000
p = rand_params(d, s, seed=500 + d + n_dead)
dead = torch.randperm(s, generator=torch.Generator().manual_seed(502 + d))[:n_dead]
p["b_enc"][dead] = -100.0
toks = torch.zeros(s, dtype=torch.int64); toks[dead] = thr
cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k, k_aux=k_aux, dead_threshold_tokens=thr)
x = torch.randn(n, d, generator=torch.Generator().manual_seed(503 + d + n_dead))
grads = {}
for name, asm in (("one", "0"), ("five", "64"), ("dense", "-1")):
    os.environ["SAEV_AMD_AUX_SMALL_MAX"] = asm
    eng = make_engine(d, s, k, k_aux=k_aux, thr=thr, max_batch=n)
    eng.load_params(p); eng.set_tracker(toks)
    eng.step_forward(x.cuda()); eng.step_dead(n); eng.step_backward(); torch.cuda.synchronize()
    grads[name] = {kk: v.cpu().clone() for kk, v in eng.grad_views().items()}
    st = eng.read_stats(); print(name, "route", eng.aux_route(), "aux", st.aux, "mse", st.mse)
state = R.TrainState.create({kk: v.clone() for kk, v in p.items()}); state.toks_since_active = toks.clone(); state.lr = 0.0
ref = R.train_step(state, x, cfg)
print("oracle aux", ref["aux"], "mse", ref["mse"], "gn", ref["grad_norm"])
for key in R.PARAM_ORDER:
    g = ref["grads"][key]
    # ref grads: after rpg? compare raw where possible
    for name in grads:
        a = grads[name][key]
        print(f"{key:6s} {name:5s} max|d| {(a - g).abs().max():.3e}  rel fro {((a - g).norm() / g.norm()):.3e}  |g| {g.norm():.4e}")
dl = dead.tolist()
for name in grads:
    a, g = grads[name]["W_dec"][dl], ref["grads"]["W_dec"][dl]
    print(name, "dead W_dec rows rel", ((a - g).norm() / g.norm()).item(), "W_enc cols rel", ((grads[name]["W_enc"][:, dl] - ref["grads"]["W_enc"][:, dl]).norm() / ref["grads"]["W_enc"][:, dl].norm()).item(),
          "b_enc dead", (grads[name]["b_enc"][dl] - ref["grads"]["b_enc"][dl]).abs().max().item(), ref["grads"]["b_enc"][dl].abs().max().item())
