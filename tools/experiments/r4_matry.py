"""configs[1] with the reference's default objective (10 Matryoshka prefixes sampled per step) next to P = 1, same data:
   rocprofv3 --kernel-trace -d /tmp/p -o run -- python tools/experiments/r4_matry.py 10"""
import math, pathlib, sys, time
import torch
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from saev_amd.engine import EngineConfig, SaeEngine
from saev_amd.nn.objectives import sample_prefixes

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
D, S, K, B = 1024, 32768, 32, 16384
dev = torch.device("cuda", 0)
eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=B, dead_threshold_tokens=10_000_000), dev)
g = torch.Generator(device=dev).manual_seed(42)
W = (torch.rand(S, D, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D)
W /= W.norm(dim=1, keepdim=True)
eng.view("W_dec").copy_(W); eng.view("W_enc").copy_(W.t())
mu = torch.randn(D, device=dev, generator=g)
pool = torch.randn(8 * B, D, device=dev, generator=g) + mu
torch.manual_seed(7)
def one(i):
    if P > 1:
        eng.set_prefixes(sample_prefixes(S, P))
    eng.train_step(pool[(i % 8) * B:(i % 8 + 1) * B], 4e-4 * min(1.0, i / 500), 1.0)
PRE = int(sys.argv[3]) if len(sys.argv) > 3 else 10  # steps before the timed ones (1 500: past the dead-latent threshold, the steady state)
for i in range(PRE): one(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(PRE, PRE + N): one(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N * 1e3
t0 = time.perf_counter()
for i in range(200): sample_prefixes(S, max(P, 2))
host = (time.perf_counter() - t0) / 200 * 1e3
print(f"P={P} after {PRE} steps: {dt:.3f} ms/step  (host: sample_prefixes {host:.3f} ms per call)  mse {eng.read_stats().mse:.4f}")
