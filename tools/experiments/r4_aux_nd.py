"""configs[1] with N latents forced dead: ms/step on the few-dead-latents route and on the dense route (SAEV_AMD_AUX_SMALL_MAX=-1).
   python tools/experiments/r4_aux_nd.py 30 [steps]"""
import math, pathlib, sys, time
import torch
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from saev_amd.engine import EngineConfig, SaeEngine

ND = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
D, S, K, B, thr = 1024, 32768, 32, 16384, 10_000_000
dev = torch.device("cuda", 0)
eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=B, dead_threshold_tokens=thr), dev)
g = torch.Generator(device=dev).manual_seed(42)
W = (torch.rand(S, D, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D)
W /= W.norm(dim=1, keepdim=True)
eng.view("W_dec").copy_(W); eng.view("W_enc").copy_(W.t())
x = torch.randn(B, D, device=dev, generator=g) + torch.randn(D, device=dev, generator=g)
toks = torch.zeros(S, dtype=torch.int64, device=dev)
if ND > 0:
    sel = torch.randperm(S, device=dev, generator=g)[:ND]
    eng.view("b_enc")[sel] = -100.0
    toks[sel] = thr
eng.set_tracker(toks)
for i in range(8): eng.train_step(x, 1e-4, 1.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N): eng.train_step(x, 1e-4, 1.0)
torch.cuda.synchronize()
st = eng.read_stats()
print(f"n_dead {ND:5d}: {(time.perf_counter() - t0) / N * 1e3:.3f} ms/step  (n_dead {st.n_dead} aux {st.aux:.4f} route {eng.aux_route()} readbacks {eng.dead_readbacks()})")
