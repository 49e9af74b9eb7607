"""300 steps of configs[1] from the same init on the same batches with the weight gradients from whole rows (SAEV_AMD_DW=rows) and
from column slices (default): the two trajectories differ by rounding only (dval is summed in another order)."""
import math, os, pathlib, sys
import torch
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from saev_amd.engine import EngineConfig, SaeEngine

D, S, K, B = 1024, 32768, 32, 16384
dev = torch.device("cuda", 0)
def make(route):
    os.environ["SAEV_AMD_DW"] = route
    eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=B), dev)
    g = torch.Generator(device=dev).manual_seed(42)
    W = (torch.rand(S, D, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D)
    W /= W.norm(dim=1, keepdim=True)
    eng.view("W_dec").copy_(W); eng.view("W_enc").copy_(W.t())
    return eng
engs = {r: make(r) for r in ("rows", "slices")}
g = torch.Generator(device=dev).manual_seed(7)
pool = torch.randn(8 * B, D, device=dev, generator=g) + 2.0 * torch.randn(D, device=dev, generator=g)
for i in range(300):
    x = pool[torch.randint(0, pool.shape[0], (B,), device=dev, generator=g)]
    lr = 4e-4 * min(1.0, (i + 1) / 100)
    for e in engs.values(): e.train_step(x, lr, 1.0)
    if i % 50 == 49 or i == 299:
        a, b = engs["rows"].read_stats(), engs["slices"].read_stats()
        dp = (engs["rows"].params - engs["slices"].params).abs().max().item()
        print(f"step {i + 1}: mse rows {a.mse:.7f} slices {b.mse:.7f} rel {abs(a.mse - b.mse) / a.mse:.2e}  grad_norm {a.grad_norm:.6f} / {b.grad_norm:.6f}  max |dparam| {dp:.2e}")
