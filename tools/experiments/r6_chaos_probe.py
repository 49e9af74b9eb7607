"""How much of a sustained-segment difference is the trajectory and how much the rule?  Three engines at configs[1]'s shape on the same
batches: B0 (matrix-core AuxK up to 64 dead latents, round 5), B1 = B0 with ONE parameter element moved by one ulp at step 700, A
(matrix-core AuxK up to 128).  From step 1500 on: mean step time (HIP events, order rotated every step) and the dead count every 100 steps."""
import sys, pathlib, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import bench
from saev_amd.engine import EngineConfig, SaeEngine

dev = torch.device("cuda:0")
B, D, S, K = bench.BATCH, bench.D_MODEL, bench.D_SAE, bench.TOP_K
PRE, N = 1500, int(sys.argv[1]) if len(sys.argv) > 1 else 2000
pool = bench.synthetic_pool(dev, "mean", 64 * B, D)
perm = torch.randperm(pool.shape[0], device=dev, generator=torch.Generator(device=dev).manual_seed(17))
def make(wide_off):
    e = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=B, dead_threshold_tokens=10_000_000, aux_wide_route=wide_off), dev)
    g = torch.Generator(device=dev).manual_seed(42)
    W = (torch.rand(S, D, device=dev, generator=g) * 2 - 1) * (6.0 / D) ** 0.5
    W /= W.norm(dim=1, keepdim=True)
    e.view("W_dec").copy_(W); e.view("W_enc").copy_(W.t())
    return e
if len(sys.argv) > 2 and sys.argv[2] == "six":  # three trajectories per rule: every engine but the first of its rule gets its own ulp
    engs = {"B0 (up to 64)": make(1), "B1 (up to 64, ulp)": make(1), "B2 (up to 64, ulp)": make(1),
            "A0 (up to 128)": make(0), "A1 (up to 128, ulp)": make(0), "A2 (up to 128, ulp)": make(0)}
else:
    engs = {"B0 (up to 64)": make(1), "B1 (up to 64, one ulp at step 700)": make(1), "A (up to 128)": make(0)}
names = list(engs)
x = torch.empty(B, D, device=dev)
lr = lambda i: 4e-4 * min(1.0, i / 500)
tot = {n: 0.0 for n in names}
dead = {n: [] for n in names}
routes = {n: {} for n in names}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
for i in range(PRE + N):
    rows = perm[(i % 64) * B:(i % 64 + 1) * B]
    engs[names[0]].gather_rows(pool, rows, out=x)
    if i == 700:
        for j, nme in enumerate(names):
            if "ulp" in nme:
                w = engs[nme].view("W_enc")
                w[3 + j, 5] = torch.nextafter(w[3 + j, 5], w[3 + j, 5] + 1)
    order = names[i % len(names):] + names[:i % len(names)]
    ev[0].record()
    for j, n in enumerate(order):
        engs[n].train_step(x, lr(i), 1.0)
        ev[j + 1].record()
    if i >= PRE:
        torch.cuda.synchronize()
        for j, n in enumerate(order):
            tot[n] += ev[j].elapsed_time(ev[j + 1])
            r = engs[n].aux_route(); routes[n][r] = routes[n].get(r, 0) + 1
        if i % 100 == 0:
            for n in names: dead[n].append(engs[n].read_stats().n_dead)
for n in names:
    print(f"{n:40s} {tot[n] / N:.4f} ms per step over {N} steps; routes {dict(sorted(routes[n].items()))}; mean n_dead (every 100 steps) {sum(dead[n]) / max(1, len(dead[n])):.1f}; n_dead every 100 steps {dead[n]}")
