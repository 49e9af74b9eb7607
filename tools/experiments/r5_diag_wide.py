"""Which number is off at the wide shapes: grad_norm of the fused step (light finalize), of fin_route = 1, of the phases (gradient
buffer, fp64 norm), and the oracle's."""
import math, os, sys
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import sae_ref as R
from test_gpu_parity import make_engine, rand_params

for d, k in ((1024, 48), (1280, 32), (1280, 64), (768, 64)):
    s, n = 4 * d, 300
    p = rand_params(d, s, seed=300 + d + k)
    gen = torch.Generator().manual_seed(301 + d + k)
    cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k, k_aux=0, n_prefixes=1, grad_clip=1.0)
    x = torch.randn(n, d, generator=gen) + 0.3
    state = R.TrainState.create({k_: v.clone() for k_, v in p.items()})
    state.lr = 0.0
    ref = R.train_step(state, x, cfg)
    out = {}
    for name, env in (("light", {}), ("fin1", {"SAEV_AMD_FIN": "1"}), ("slices_a", {"SAEV_AMD_DW": "slices_a"})):
        os.environ.update(env)
        eng = make_engine(d, s, k, k_aux=0, max_batch=n)
        for key in env: os.environ.pop(key)
        eng.load_params(p)
        eng.train_step(x.cuda(), 0.0, 1.0)
        out[name] = eng.read_stats().grad_norm
        eng.close()
    eng = make_engine(d, s, k, k_aux=0, max_batch=n)
    eng.load_params(p)
    eng.step_forward(x.cuda()); eng.step_dead(n); eng.step_backward()
    g = eng.grad_views()
    W = eng.view("W_dec").double()
    gd = g["W_dec"].double()
    proj = gd - (gd * W).sum(1, keepdim=True) / (W * W).sum(1, keepdim=True) * W
    n64 = math.sqrt((proj ** 2).sum().item() + (g["W_enc"].double() ** 2).sum().item() + (g["b_enc"].double() ** 2).sum().item() + (g["b_dec"].double() ** 2).sum().item())
    eng.step_tail(0.0, 1.0)
    out["phases"] = eng.read_stats().grad_norm
    print(d, k, "oracle", ref["grad_norm"], "fp64 of our gradient", n64, out, flush=True)
