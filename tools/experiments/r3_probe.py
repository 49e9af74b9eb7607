"""Round-3 probe of the bench workload (configs[1]): per-latent firing histogram of the (row, latent) pairs at a few
points of the run (what a hot/cold split of dw_rows could save), and how much of a sustained step the host spends
enqueueing (is the 7 % between the kernel sum and the wall host time or read-backs?).

    python tools/experiments/r3_probe.py [--steps-at 25,700,1500,2600]
"""
import argparse
import json
import math
import pathlib
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from saev_amd.engine import EngineConfig, SaeEngine  # noqa: E402

D, S, K, B = 1024, 32768, 32, 16384


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps-at", default="25,700,1500,2600")
    ap.add_argument("--series", action="store_true", help="per-100-step wall time / read-backs over steps 600-2700 instead of the histograms")
    args = ap.parse_args()
    marks = [int(v) for v in args.steps_at.split(",")]
    dev = torch.device("cuda", 0)
    eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=B, dead_threshold_tokens=10_000_000), dev)
    g = torch.Generator(device=dev).manual_seed(42)
    W = (torch.rand(S, D, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D)
    W /= W.norm(dim=1, keepdim=True)
    eng.view("W_dec").copy_(W)
    eng.view("W_enc").copy_(W.t())
    del W
    g = torch.Generator(device=dev).manual_seed(17)
    mu = torch.randn(D, device=dev, generator=torch.Generator(device=dev).manual_seed(17))
    pool = torch.randn(64 * B, D, device=dev, generator=g) + mu
    perm = torch.randperm(pool.shape[0], device=dev, generator=g)
    x = torch.empty(B, D, device=dev)
    lr = lambda i: 4e-4 * min(1.0, i / 500)  # noqa: E731

    def one(i):
        rows = perm[(i % 64) * B:(i % 64 + 1) * B]
        eng.gather_rows(pool, rows, out=x)
        eng.train_step(x, lr(i), 1.0)

    out = {"hist": [], "series": []}
    i = 0
    if args.series:
        while i < 600:
            one(i); i += 1
        for blk in range(21):
            torch.cuda.synchronize()
            rb0 = eng.dead_readbacks()
            t0 = time.perf_counter()
            for j in range(100):
                one(i); i += 1
            torch.cuda.synchronize()
            out["series"].append({"first": i - 100, "ms": (time.perf_counter() - t0) * 10, "readbacks": eng.dead_readbacks() - rb0,
                                  "n_dead": eng.read_stats().n_dead, "route": eng.aux_route()})
        marks = []
    for m in marks:
        while i < m:
            one(i)
            i += 1
        torch.cuda.synchronize()
        idx, val, _ = eng.last_codes(B)
        cnt = torch.bincount(idx.reshape(-1).long(), minlength=S).cpu()
        srt, _ = cnt.sort(descending=True)
        tot = int(cnt.sum())
        rec = {"step": m, "pairs": tot, "latents_used": int((cnt > 0).sum()), "max": int(srt[0]),
               "top": [int(v) for v in srt[:16]]}
        for thr in (B // 2, B // 4, B // 8, B // 16, B // 32, 512, 64):
            sel = cnt > thr
            rec[f"gt_{thr}"] = {"latents": int(sel.sum()), "pair_share": float(cnt[sel].sum()) / tot}
        # bytes the gather kernel moves for the hot latents vs one shared streaming pass
        out["hist"].append(rec)
        # host enqueue time per step vs wall, 100 steps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host = 0.0
        for j in range(100):
            h0 = time.perf_counter()
            one(i)
            host += time.perf_counter() - h0
            i += 1
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        rec["host_ms_per_step"] = host / 100 * 1e3
        rec["wall_ms_per_step"] = wall / 100 * 1e3
        rec["readbacks_total"] = eng.dead_readbacks()
        rec["aux_route"] = eng.aux_route()
        rec["n_dead"] = eng.read_stats().n_dead
    # pure host cost of one step: enqueue into an EMPTY queue (a full queue makes every launch wait for the GPU)
    hs = []
    for j in range(20):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        one(i)
        hs.append((time.perf_counter() - h0) * 1e3)
        i += 1
    out["host_enqueue_ms_empty_queue"] = {"median": sorted(hs)[10], "min": min(hs), "max": max(hs)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
