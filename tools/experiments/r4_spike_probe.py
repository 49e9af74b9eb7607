"""Loss spikes over a long run (bench workload, lr 4e-4 after a 500-step ramp): per window of 200 steps the largest MSE and dead
count seen (statistics read every 4th step) -- are the spikes the training dynamics or a route?  argv: k_aux windows"""
import math, sys, time, torch
sys.path.insert(0, ".")
from saev_amd.engine import EngineConfig, SaeEngine
D, S, B, K = 1024, 32768, 16384, 32
k_aux = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_win = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = torch.device("cuda:0")
eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, k_aux=k_aux, max_batch=B), dev)
g = torch.Generator(device=dev).manual_seed(42)
W = (torch.rand(S, D, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D)
W /= W.norm(dim=1, keepdim=True)
eng.view("W_dec").copy_(W); eng.view("W_enc").copy_(W.t())
g = torch.Generator(device=dev).manual_seed(17)
mu = torch.randn(D, device=dev, generator=torch.Generator(device=dev).manual_seed(17))
pool = torch.randn(64 * B, D, device=dev, generator=g) + mu
perm = torch.randperm(pool.shape[0], device=dev, generator=g)
x = torch.empty(B, D, device=dev)
step = 0
for w in range(n_win):
    mx, mn, dmax, amax, gmax = 0.0, 1e9, 0, 0.0, 0.0
    routes = [0, 0, 0, 0]
    for _ in range(200):
        rows = perm[(step % 64) * B : (step % 64 + 1) * B]
        eng.gather_rows(pool, rows, out=x)
        eng.train_step(x, 4e-4 * min(1.0, step / 500), 1.0)
        routes[eng.aux_route()] += 1
        step += 1
        if step % 4 == 0:
            st = eng.read_stats()
            mx = max(mx, st.mse); mn = min(mn, st.mse); dmax = max(dmax, st.n_dead); amax = max(amax, st.aux); gmax = max(gmax, st.grad_norm)
    print(f"steps {step - 200:5d}-{step:5d}: mse min {mn:.4f} max {mx:.4f}  n_dead max {dmax}  aux max {amax:.4f}  grad norm max {gmax:.3f}  routes {routes}", flush=True)
