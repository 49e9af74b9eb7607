"""configs[1] with 1 000 latents forced dead (bench.py's auxk_active[1]) on its own, for a per-kernel profile:
   rocprofv3 --kernel-trace -d /tmp/p -o run -- python tools/experiments/r3_aux1000.py"""
import math, pathlib, sys, time
import torch
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from saev_amd.engine import EngineConfig, SaeEngine

D, S, K, B, thr = 1024, 32768, 32, 16384, 10_000_000
dev = torch.device("cuda", 0)
eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=B, dead_threshold_tokens=thr), dev)
g = torch.Generator(device=dev).manual_seed(42)
W = (torch.rand(S, D, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D)
W /= W.norm(dim=1, keepdim=True)
eng.view("W_dec").copy_(W); eng.view("W_enc").copy_(W.t())
x = torch.randn(B, D, device=dev, generator=g) + torch.randn(D, device=dev, generator=g)
sel = torch.randperm(S, device=dev, generator=g)[:1000]
eng.view("b_enc")[sel] = -100.0
toks = torch.zeros(S, dtype=torch.int64, device=dev); toks[sel] = thr
eng.set_tracker(toks)
for i in range(5): eng.train_step(x, 1e-4, 1.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 20
for i in range(N): eng.train_step(x, 1e-4, 1.0)
torch.cuda.synchronize()
st = eng.read_stats()
print(f"{(time.perf_counter() - t0) / N * 1e3:.3f} ms/step  n_dead {st.n_dead} aux {st.aux:.4f} route {eng.aux_route()} readbacks {eng.dead_readbacks()}")
