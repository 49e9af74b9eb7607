"""Step time of the bench.py workload in windows of 200 steps over a long run (does it drift, and why?)."""
import math
import sys
import time

import torch

sys.path.insert(0, ".")
from saev_amd.engine import EngineConfig, SaeEngine  # noqa: E402

D, S, B, K = 1024, 32768, 16384, 32
k_aux = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_win = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, k_aux=k_aux, max_batch=B), dev)
g = torch.Generator(device=dev).manual_seed(42)
W = (torch.rand(S, D, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D)
W /= W.norm(dim=1, keepdim=True)
eng.view("W_dec").copy_(W)
eng.view("W_enc").copy_(W.t())
g = torch.Generator(device=dev).manual_seed(17)
mu = torch.randn(D, device=dev, generator=torch.Generator(device=dev).manual_seed(17))
pool = torch.randn(64 * B, D, device=dev, generator=g) + mu
perm = torch.randperm(pool.shape[0], device=dev, generator=g)
x = torch.empty(B, D, device=dev)
step = 0
eng.enable_kernel_timing(True)
for w in range(n_win):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    routes = [0, 0, 0, 0]
    for _ in range(200):
        rows = perm[(step % 64) * B : (step % 64 + 1) * B]
        eng.gather_rows(pool, rows, out=x)
        eng.train_step(x, 4e-4 * min(1.0, step / 500), 1.0)
        routes[eng.aux_route()] += 1  # (host-side decision of the step just enqueued: no synchronisation)
        step += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200 * 1e3
    st = eng.read_stats()
    print(f"steps {step - 200:5d}-{step:5d}: {dt:.3f} ms/step  encoder {eng.encoder_ms():.3f} ms  mse {st.mse:.4f} n_dead {st.n_dead} "
          f"cand_max {st.cand_max} dense_route {st.dense_route}  aux routes none/small/small+read/dense {routes}", flush=True)
