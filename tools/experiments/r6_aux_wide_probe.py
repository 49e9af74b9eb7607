"""What a LOOSE bound of the dead count costs at configs[1]'s shape.  20 latents are dead; 80 more sit three steps short
of the threshold, stay quiet for two steps (bias -30) and fire on the third (bias +30 for that step): the tracker records of
steps 0 and 1 count 100 latents "dead or within four steps of it", so steps 4 and 5 -- the first that size their auxiliary work by
a record -- run under a bound of 100 with 20 dead latents.  Step times (HIP events) with the matrix-core kernels up to 128 dead
latents (default: the device-side count picks the one-block kernels) and up to 64 as in round 5 (aux_wide_route=1: dense algebra)."""
import sys, pathlib, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import bench
from saev_amd.engine import EngineConfig, SaeEngine

dev = torch.device("cuda:0")
B, D, S, K = bench.BATCH, bench.D_MODEL, bench.D_SAE, bench.TOP_K
thr = 10_000_000
g = torch.Generator(device=dev).manual_seed(5)
pool = bench.synthetic_pool(dev, "mean", 8 * B, D)
perm = torch.randperm(S, device=dev, generator=g)
dead, sleepy = perm[:20], perm[20:100]
for wide_off in (0, 1, 0, 1):
    eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=B, dead_threshold_tokens=thr, aux_wide_route=wide_off), dev)
    gw = torch.Generator(device=dev).manual_seed(42)
    W = (torch.rand(S, D, device=dev, generator=gw) * 2 - 1) * (6.0 / D) ** 0.5
    W /= W.norm(dim=1, keepdim=True)
    eng.view("W_dec").copy_(W); eng.view("W_enc").copy_(W.t())
    eng.view("b_enc")[dead] = -30.0
    eng.view("b_enc")[sleepy] = -30.0
    eng.params_touched()
    for i in range(6):  # (settle: images, streamed preparation)
        eng.train_step(pool[i * B:(i + 1) * B], 0.0, 1.0)
    toks = torch.zeros(S, dtype=torch.int64)
    toks[dead.cpu()] = thr
    toks[sleepy.cpu()] = thr - 3 * B
    eng.set_tracker(toks)
    N = 16
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
    routes, deads = [], []
    for i in range(N):
        x = pool[(i % 8) * B:(i % 8 + 1) * B]
        if i in (2, 3):
            eng.view("b_enc")[sleepy] = 30.0 if i == 2 else -30.0
            eng.params_touched()
        ev[i][0].record()
        eng.train_step(x, 0.0, 1.0)
        ev[i][1].record()
        routes.append(eng.aux_route())
    torch.cuda.synchronize()
    deads = [eng.read_stats().n_dead] * N
    ms = [a.elapsed_time(b) for a, b in ev]
    print(f"aux_wide_route {wide_off}: steps 4 / 5 (bound 100): {ms[4]:.3f} / {ms[5]:.3f} ms, routes {routes[4:6]}; "
          f"steps 8-15 (bound = count; {deads[10]} dead at the end): {sum(ms[8:]) / 8:.3f} ms, routes {sorted(set(routes[8:]))}; read-backs {eng.dead_readbacks()}")
    eng.close()
