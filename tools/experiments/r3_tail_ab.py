"""A/B of the two tails on one GPU: saev_train_step (projection inside Adam, clip norm from the rows' statistics) against the
phase calls (rpg pass, flat Adam) on identical state.  Prints, per tensor, how many elements of params / m / v differ."""
import sys, pathlib, torch
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from saev_amd.engine import EngineConfig, SaeEngine

def run(fused, clip, steps=3, nd=True):
    torch.manual_seed(0)
    d, s, k, b = 256, 1024, 16, 512
    eng = SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=k, k_aux=32, dead_threshold_tokens=1000, max_batch=b), "cuda:0")
    g = torch.Generator(device="cuda").manual_seed(1)
    W = torch.randn(s, d, device="cuda", generator=g); W /= W.norm(dim=1, keepdim=True)
    eng.view("W_dec").copy_(W); eng.view("W_enc").copy_(W.t())
    if nd:
        toks = torch.zeros(s, dtype=torch.int64); toks[::9] = 1000; eng.set_tracker(toks)
    x = torch.randn(b, d, device="cuda", generator=g)
    out = []
    for i in range(steps):
        if fused:
            eng.train_step(x, 1e-3 * i, clip)
        else:
            eng.step_forward(x, training=True); eng.step_dead(b); eng.step_backward(); eng.step_tail(1e-3 * i, clip)
        torch.cuda.synchronize()
        out.append((eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.read_stats().grad_norm, eng))
    return out

for clip in (1.0, 0.02):
    a, b = run(True, clip), run(False, clip)
    for i, (u, v) in enumerate(zip(a, b)):
        eng = u[4]
        print(f"clip {clip} step {i}: grad_norm {u[3]!r} vs {v[3]!r}")
        for nm, j in (("params", 0), ("m", 1), ("v", 2)):
            for t in ("W_dec", "b_dec", "W_enc", "b_enc"):
                x1, x2 = eng.view(t, u[j]), eng.view(t, v[j])
                nd = (x1 != x2).sum().item()
                if nd:
                    print(f"   {nm}.{t}: {nd} of {x1.numel()} differ, max abs {(x1 - x2).abs().max().item():.3e}")
