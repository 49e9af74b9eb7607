"""Throughput of the shuffled activation feed by itself (no training): rows/s and GB/s delivered as device batches.

    python tools/bench_feed.py --gb 8 --d-model 1024 --threads 1 4 8

Writes a synthetic protocol-2.1 cache (page-cache hot, so this measures the host copy + PCIe + gather path, not the
disk), then times one epoch in the streaming-reservoir mode per thread count and one in the resident mode."""
import argparse
import dataclasses
import shutil
import tempfile
import time

import numpy as np
import torch

from saev_amd import data
from saev_amd.engine import EngineConfig, SaeEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=8.0)
    ap.add_argument("--d-model", type=int, default=1024)
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 4, 8])
    ap.add_argument("--root", default=None)
    a = ap.parse_args()
    D, T = a.d_model, a.tokens
    n_ex = int(a.gb * 1e9 / (4 * D * (T + 1)))
    root = tempfile.mkdtemp(prefix="feed_", dir=a.root)
    try:
        rng = np.random.default_rng(0)
        acts = rng.standard_normal((n_ex, 1, T + 1, D), dtype=np.float32)
        d = data.write_shards(root, acts, layers=(23,), cls_token=True, max_tokens_per_shard=(T + 1) * 1024)
        del acts
        dev = torch.device("cuda:0")
        eng = SaeEngine(EngineConfig(d_model=D, d_sae=1024, top_k=8, k_aux=0, max_batch=a.batch), dev)
        cfg = data.ShuffledConfig(shards=d, layer=23, batch_size=a.batch, buffer_size=64)
        n_rows = n_ex * T

        def run(dl, label):
            dl.engine = eng
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for b in dl:
                n += b["act"].shape[0]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert n == n_rows
            print(f"{label}: {n / dt / 1e6:.2f} M rows/s, {n * D * 4 / dt / 1e9:.2f} GB/s ({dt:.2f} s for {n} rows)", flush=True)
            if getattr(dl, "reservoir", None) is not None:
                print("    phases [s, summed over threads]: " + ", ".join(f"{k} {v:.2f}" for k, v in dl.reservoir.phase_s.items()), flush=True)

        for nt in a.threads:
            dl = data.ShuffledDataLoader(dataclasses.replace(cfg, n_threads=nt), device=dev, resident=False)
            run(dl, f"streaming reservoir, {nt} reader thread(s)")
        t0 = time.perf_counter()
        dl = data.ShuffledDataLoader(cfg, device=dev, resident=True)
        print(f"resident pool load: {time.perf_counter() - t0:.2f} s")
        run(dl, "resident pool (HIP gather only)")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
