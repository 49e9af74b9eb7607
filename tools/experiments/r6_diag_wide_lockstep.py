"""The two AuxK rules in lock step at configs[1]'s shape: engine A (matrix-core kernels up to 128 dead latents) and engine B (round 5: up to
64) start from the same trained state; before every step B takes A's parameters and Adam moments, then both take the step on the same
batch.  Any step on which they disagree beyond rounding is a bug in one of the routes; none is expected."""
import sys, pathlib, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import bench
from saev_amd.engine import EngineConfig, SaeEngine

dev = torch.device("cuda:0")
B, D, S, K = bench.BATCH, bench.D_MODEL, bench.D_SAE, bench.TOP_K
PRE, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500, int(sys.argv[2]) if len(sys.argv) > 2 else 600
pool = bench.synthetic_pool(dev, "mean", 64 * B, D)
perm = torch.randperm(pool.shape[0], device=dev, generator=torch.Generator(device=dev).manual_seed(17))
def make(wide_off):
    e = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=B, dead_threshold_tokens=10_000_000, aux_wide_route=wide_off), dev)
    g = torch.Generator(device=dev).manual_seed(42)
    W = (torch.rand(S, D, device=dev, generator=g) * 2 - 1) * (6.0 / D) ** 0.5
    W /= W.norm(dim=1, keepdim=True)
    e.view("W_dec").copy_(W); e.view("W_enc").copy_(W.t())
    return e
A, Bq = make(0), make(1)
x = torch.empty(B, D, device=dev)
lr = lambda i: 4e-4 * min(1.0, i / 500)
for i in range(PRE):
    rows = perm[(i % 64) * B:(i % 64 + 1) * B]
    A.gather_rows(pool, rows, out=x)
    A.train_step(x, lr(i), 1.0)
    Bq.train_step(x, lr(i), 1.0)     # (free-running so far: both build their own tracker records)
print("after pretrain: n_dead", A.read_stats().n_dead, Bq.read_stats().n_dead, "tracker equal", torch.equal(A.toks_since_active, Bq.toks_since_active))
worst = 0.0
hist = {}
tA = tB = 0.0
tpair = {}
evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for i in range(PRE, PRE + N):
    Bq.params.copy_(A.params); Bq.adam_m.copy_(A.adam_m); Bq.adam_v.copy_(A.adam_v)
    Bq.toks_since_active.copy_(A.toks_since_active)
    rows = perm[(i % 64) * B:(i % 64 + 1) * B]
    A.gather_rows(pool, rows, out=x)
    evs[0].record()
    A.train_step(x, lr(i), 1.0)
    evs[1].record()
    Bq.train_step(x, lr(i), 1.0)
    evs[2].record()
    torch.cuda.synchronize()
    ra, rb = A.aux_route(), Bq.aux_route()
    hist[(ra, rb)] = hist.get((ra, rb), 0) + 1
    a_ms, b_ms = evs[0].elapsed_time(evs[1]), evs[1].elapsed_time(evs[2])
    tA += a_ms; tB += b_ms
    t = tpair.setdefault((ra, rb), [0.0, 0.0]); t[0] += a_ms; t[1] += b_ms
    if (ra, rb) != (1, 1) or i % 50 == 0:
        dp = (A.params - Bq.params).abs().max().item()
        sa, sb = A.read_stats(), Bq.read_stats()
        worst = max(worst, dp)
        if dp > 1e-6 or sa.n_dead != sb.n_dead or i % 50 == 0:
            print(f"step {i}: routes {ra} {rb} n_dead {sa.n_dead} {sb.n_dead} aux {sa.aux:.6g} {sb.aux:.6g} max |dp| {dp:.3g} tracker equal {torch.equal(A.toks_since_active, Bq.toks_since_active)}")
print("route pairs (A, B):", hist, "worst |dp|", worst)
print(f"mean step time on identical states: A (up to 128) {tA / N:.4f} ms, B (up to 64) {tB / N:.4f} ms")
for key, (a_, b_) in sorted(tpair.items()):
    print(f"   routes {key}: {hist[key]} steps, A {a_ / hist[key]:.4f} ms, B {b_ / hist[key]:.4f} ms")
