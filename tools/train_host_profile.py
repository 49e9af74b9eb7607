"""Where the host spends a step of framework.train.train(): cProfile over the train loop at configs[1]'s shape
on a resident pool, against the bare engine loop.  Prints the loop's wall time per step, the host time outside the library call
(everything Python does per step: loader, limiter, schedule, bookkeeping) and the top functions by own time.

    python tools/train_host_profile.py [--steps 300]
"""
import argparse
import cProfile
import pathlib
import pstats
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    import saev_amd.utils.scheduling as sched
    from saev_amd import data, nn
    from saev_amd.framework import train as T
    from saev_amd.nn import modeling, objectives

    dev = torch.device("cuda:0")
    B, D, S, K = bench.BATCH, bench.D_MODEL, bench.D_SAE, bench.TOP_K
    pool = bench.synthetic_pool(dev, "mean", 32 * B, D)
    dcfg = data.ShuffledConfig(batch_size=B, seed=17)
    cfg = T.Config(train_data=dcfg, val_data=dcfg, n_train=args.steps * B, n_val=B,
                   sae=nn.SparseAutoencoderConfig(d_model=D, d_sae=S, reinit_blend=0.0, activation=modeling.TopK(top_k=K)),
                   objective=objectives.Matryoshka(n_prefixes=1), log_every=10**9, track=False, runs_root=pathlib.Path("/tmp/saev_prof_runs"))
    t = {}
    orig_iter = sched.BatchLimiter.__iter__
    prof = cProfile.Profile()

    def timed_iter(self, _orig=orig_iter):
        torch.cuda.synchronize()
        t["t0"] = time.perf_counter()
        prof.enable()
        yield from _orig(self)

    sched.BatchLimiter.__iter__ = timed_iter
    try:
        saes, objs, run, n_steps = T.train([cfg], train_pool=pool)
    finally:
        sched.BatchLimiter.__iter__ = orig_iter
    torch.cuda.synchronize()
    prof.disable()
    dt = time.perf_counter() - t["t0"]
    st = pstats.Stats(prof)
    total = sum(v[2] for v in st.stats.values())  # own time of everything
    lib = sum(v[2] for k, v in st.stats.items() if "saev_train_step" in k[2] or "_CFuncPtr" in k[2] or "CFunctionType" in k[2])
    print(f"train(): {n_steps} steps, {dt / n_steps * 1e3:.3f} ms per step wall; profiled own time {total / n_steps * 1e3:.3f} ms per step "
          "(cProfile inflates Python frames several-fold: read the ranking, not the sum)")
    st.sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
