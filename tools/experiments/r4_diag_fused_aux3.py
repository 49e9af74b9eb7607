"""whose clip norm is right?  fp64 recomputation from the HIP path's raw gradients vs the oracle's and the HIP path's own"""
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
eng = make_engine(d, s, k, k_aux=k_aux, thr=thr, max_batch=n)
eng.load_params(p); eng.set_tracker(toks)
eng.step_forward(x.cuda()); eng.step_dead(n); eng.step_backward(); torch.cuda.synchronize()
g = {kk: v.cpu().double() for kk, v in eng.grad_views().items()}
w = eng.view("W_dec").cpu().double()
print("row norms of W_dec: min %.9f max %.9f" % (w.norm(dim=1).min(), w.norm(dim=1).max()))
par = (g["W_dec"] * w).sum(1, keepdim=True)
proj = g["W_dec"] - par * w
tot64 = math.sqrt(float((proj**2).sum() + (g["W_enc"]**2).sum() + (g["b_enc"]**2).sum() + (g["b_dec"]**2).sum()))
eng2 = make_engine(d, s, k, k_aux=k_aux, thr=thr, max_batch=n)
eng2.load_params(p); eng2.set_tracker(toks)
eng2.train_step(x.cuda(), 0.0, 1.0)
state = R.TrainState.create({kk: v.clone() for kk, v in p.items()}); state.toks_since_active = toks.clone(); state.lr = 0.0
ref = R.train_step(state, x, cfg)
print(f"fp64 from HIP raw grads {tot64:.7f}   HIP reported {eng2.read_stats().grad_norm:.7f}   oracle {ref['grad_norm']:.7f}")
# the oracle's pieces in fp64 from its own projected gradient
rg = {kk: v.double() for kk, v in ref["grads"].items()}
print("oracle pieces: W_dec %.7f  HIP-fp64 W_dec %.7f" % (float((rg["W_dec"]**2).sum()), float((proj**2).sum())))
dl = dead.tolist()
print("dead rows: oracle %.7f  HIP-fp64 %.7f ; live rows: oracle %.7f  HIP-fp64 %.7f" % (
    float((rg["W_dec"][dl]**2).sum()), float((proj[dl]**2).sum()),
    float((rg["W_dec"]**2).sum() - (rg["W_dec"][dl]**2).sum()), float((proj**2).sum() - (proj[dl]**2).sum())))
# is the oracle's projection done with the normalised rows?
wn = R.normalize_w_dec(p["W_dec"].clone()).double()
print("max |W_dec(engine) - normalised(p)|", (w - wn).abs().max().item())
