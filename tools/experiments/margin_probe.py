"""How much of the f16r margin is the *largest* encoder-column norm?  Trains the bench loop for N steps, then reports the
spread of ||W_enc[:, s]|| (the row margin uses its maximum), and per row the number of first-pass survivors a per-latent
margin would keep against what the max-norm margin keeps (computed on the host from exact fp32 pre-activations).
   PYTHONPATH=$PWD python tools/experiments/margin_probe.py [steps]"""
import math
import sys

import torch

sys.path.insert(0, ".")
from saev_amd.engine import EngineConfig, SaeEngine  # noqa: E402

D, S, B, K = 1024, 32768, 16384, 32
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0")
eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=B, dead_threshold_tokens=10_000_000), dev)
g = torch.Generator(device=dev).manual_seed(42)
W = (torch.rand(S, D, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D)
W /= W.norm(dim=1, keepdim=True)
eng.view("W_dec").copy_(W)
eng.view("W_enc").copy_(W.t())
del W
g = torch.Generator(device=dev).manual_seed(17)
mu = torch.randn(D, device=dev, generator=torch.Generator(device=dev).manual_seed(17))
pool = torch.randn(64 * B, D, device=dev, generator=g) + mu
perm = torch.randperm(pool.shape[0], device=dev, generator=g)
x = torch.empty(B, D, device=dev)
for i in range(steps):
    rows = perm[(i % 64) * B : (i % 64 + 1) * B]
    eng.gather_rows(pool, rows, out=x)
    eng.train_step(x, 4e-4 * min(1.0, i / 500), 1.0)
    if i in (0, 100, 500, 1000, steps - 1):
        We = eng.view("W_enc")
        nrm = We.norm(dim=0)
        q = torch.quantile(nrm, torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev))
        xs = x[:512]
        xc = xs - x.mean(dim=0)
        h = xs @ We + eng.view("b_enc")
        kth = h.topk(K, dim=1).values[:, -1:]
        c = 1.05 * (2.0 ** -10 + D * 2.0 ** -22)
        e_row = c * xc.norm(dim=1, keepdim=True)            # x ||w||: per-latent error bound
        n_max = ((h + 2 * e_row * nrm.max()) >= kth).sum(dim=1).float().mean().item()   # crude: within 2 E of the cut
        n_lat = ((h + 2 * e_row * nrm[None, :]) >= kth).sum(dim=1).float().mean().item()
        print(f"step {i}: ||w|| median {q[0]:.3f} p90 {q[1]:.3f} p99 {q[2]:.3f} p99.9 {q[3]:.3f} max {nrm.max():.3f}; "
              f"entries within the margin of the cut: max-norm margin {n_max:.1f}, per-latent margin {n_lat:.1f}", flush=True)
