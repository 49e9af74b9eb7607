"""Time the train step on an arbitrary shape (defaults: BASELINE.json configs[3], the ViT-g/14 shape).

    python tools/probe_shape.py --d-model 1280 --d-sae 81920 --top-k 64 --batch 16384
"""
import argparse
import time

import torch

from saev_amd.engine import EngineConfig, SaeEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d-model", type=int, default=1280)
    ap.add_argument("--d-sae", type=int, default=81920)
    ap.add_argument("--top-k", type=int, default=64)
    ap.add_argument("--k-aux", type=int, default=0)
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--encoder", default=None)
    ap.add_argument("--n-dead", type=int, default=0, help="force this many latents dead (very negative b_enc, tracker at threshold)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    kw = {} if a.encoder is None else {"encoder": a.encoder}
    cfg = EngineConfig(d_model=a.d_model, d_sae=a.d_sae, top_k=a.top_k, k_aux=a.k_aux, max_batch=a.batch, **kw)
    eng = SaeEngine(cfg, dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    bound = 1.0 / a.d_model**0.5
    pv = eng.param_views()
    pv["W_enc"].copy_(((torch.rand(a.d_model, a.d_sae, generator=g) * 2 - 1) * bound).to(dev))
    pv["W_dec"].copy_(pv["W_enc"].t())
    x = torch.randn(a.batch, a.d_model, generator=g).to(dev)
    if a.n_dead:
        dead = torch.randperm(a.d_sae, generator=g)[: a.n_dead].to(dev)
        pv["b_enc"][dead] = -1e3  # never selected by TopK, so they stay dead; AuxK still sees their pre-activations
        toks = torch.zeros(a.d_sae, dtype=torch.int64)
        toks[dead.cpu()] = cfg.dead_threshold_tokens
        eng.set_tracker(toks)
    eng.enable_kernel_timing(True)
    for i in range(3):
        eng.train_step(x, lr=1e-4, max_norm=1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        eng.train_step(x, lr=1e-4, max_norm=1.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    st = eng.read_stats()
    print(
        f"D={a.d_model} S={a.d_sae} k={a.top_k} B={a.batch}: {ms:.2f} ms/step, {a.batch / ms * 1e3:.3e} acts/s, "
        f"encoder {eng.encoder_ms():.2f} ms, cand_max {st.cand_max}, overflow rows {st.n_overflow_rows}, mse {st.mse:.4f}, "
        f"n_dead {st.n_dead}, aux {st.aux:.5f}"
    )


if __name__ == "__main__":
    main()
