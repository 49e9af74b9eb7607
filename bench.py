"""Headline benchmark: activations/sec of the TopK-SAE train step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): d_in = 1024, 32x expansion (d_sae = 32768), k = 32,
batch 16384 activations per GPU, fp32, synthetic N(mu, 1) activations held in a device-resident pool
of 64 batches (the reference's reservoir capacity, data/shuffled.py:45-63).  One step = draw a batch
from the pool (row gather) + the full train step (renorm, encode+TopK, sparse decode, MSE, AuxK
bookkeeping, backward, rpg, clip, Adam).  With N > 1 ranks (launched by torch.distributed.run) the
batch is per-GPU (weak scaling) and each step exchanges the 268 MB gradient over RCCL -- reduce-scatter, 1/N of the
optimizer tail per rank, all-gather of the parameters (`--tail`, framework/ddp.py) -- plus the fired-latent flags.

Timed region ("steady-state train step"): the loop first trains `--pretrain-steps` steps (default 1 500: past the
reference's dead-latent threshold of 10 M tokens = 611 of these steps, past the 500-step lr warm-up, and past the ~1 000 steps
over which latent usage spreads and the gather kernels lose their L2 hits -- the step time is flat from there on), then runs the
contract's W untimed warm-up steps and times exactly K steps.  `value` is therefore the step a training run spends its
life in -- tracker consulted, AuxK on whatever is dead, latent usage spread over ~27 000 latents.  The first 5 + 20 steps
from random init (what rounds 1-3 reported; ~18 % faster because 19 000 latents are still unused and AuxK is off) are
timed on the way and reported as `from_random_init`.

Prints ONE JSON line on rank 0 (contract in the task description), including
  "roofline":     the encoder's first-pass MFMA kernel: achieved TFLOP/s = algorithmic 2*B*D*S flops / mean kernel
                  duration over the timed steps (HIP events recorded on the launch stream by the library).  The default
                  encoder (f16r) forms every product ONCE on v_mfma_f32_16x16x32_f16 with a rigorous error margin and
                  recomputes the survivors exactly in fp32 (separate kernels inside ms_per_step), so it is priced
                  against the dense f16 MFMA peak (2.5 PFLOP/s); --encoder f16x3 (three products per fp32 product)
                  reports `executed_tflops` = 3x beside it; --encoder f32 is priced against the 157.3 TFLOP/s fp32
                  matrix peak;
  "cpu_baseline": the CPU oracle (oracle/sae_ref.py, a restatement of the reference's PyTorch-CPU
                  step) timed on this box's host cores as BASELINE.md section 3 prescribes: all cores and 8 threads,
                  one warm-up + five timed steps each (~110 s of CPU work together);
  with N > 1:     "collectives": RCCL bus bandwidth of the step's own collectives at their real sizes.
"""

import argparse
import json
import math
import os
import pathlib
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

D_MODEL, D_SAE, TOP_K, BATCH = 1024, 32768, 32, 16384
POOL_BATCHES = 64
F32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
F16_MFMA_PEAK_TFLOPS = 2500.0  # same table, "Peak BF16/FP16 MFMA" dense


def cpu_baseline(n_rows: int = BATCH, steps: int = 5, rows_8t: int = 4096):
    """BASELINE.md section 3's procedure on the GPU box's host: the CPU oracle's train step (oracle/sae_ref.py: dense GEMMs + autograd,
    as the reference does) with ``torch.set_num_threads(n)`` for n = all host cores and n = 8, one warm-up + `steps` timed steps
    each.  All cores: the benchmark's own 16 384-row batch (about 11 s per step: ~70 s).  8 threads: `rows_8t`-row batches of the
    same data (a full batch takes ~25 s per step there; the dense products scale with the rows, the Adam tail -- ~1 % of the
    step -- does not, so the quarter batch understates the 8-thread rate by about that much): ~40 s."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import sae_ref as R

    cfg = R.RefConfig(d_model=D_MODEL, d_sae=D_SAE, top_k=TOP_K)
    x = torch.randn(n_rows, D_MODEL, generator=torch.Generator().manual_seed(17))
    all_cores = torch.get_num_threads()

    def run(threads, xb):
        torch.set_num_threads(threads)
        gen = torch.Generator().manual_seed(42)
        state = R.TrainState.create(R.init_params(cfg, gen))
        sched = R.WarmupCosine(0.0, cfg.n_lr_warmup, cfg.lr, 1000, 0.0)
        R.train_step(state, xb, cfg, sched)  # the warm-up step (threads, allocator, first-touch of the 2 GB of (B, S) temporaries)
        per_step = []
        for _ in range(steps):
            t0 = time.perf_counter()
            R.train_step(state, xb, cfg, sched)
            per_step.append(time.perf_counter() - t0)
        dt = sum(per_step)
        return {"threads": threads, "rows_per_step": xb.shape[0], "warmup_steps": 1, "timed_steps": steps,
                "seconds_per_step": dt / steps, "seconds_per_step_min": min(per_step), "seconds_per_step_max": max(per_step),
                "activations_per_sec": xb.shape[0] * steps / dt}

    try:
        full = run(all_cores, x)
        eight = run(min(8, all_cores), x[:rows_8t].contiguous())
    finally:
        torch.set_num_threads(all_cores)
    return {
        "value": full["activations_per_sec"], "unit": "activations/sec", "cores": all_cores, "kind": "port",
        "sample": f"1 warm-up + {steps} timed steps of {n_rows} rows (the benchmark's batch) at {all_cores} threads; the 8-thread row: "
                  f"1 warm-up + {steps} timed steps of {eight['rows_per_step']} rows; d_model={D_MODEL}, d_sae={D_SAE}, k={TOP_K}, fp32 "
                  "PyTorch-CPU oracle (dense GEMMs + autograd, as the reference does) -- BASELINE.md section 3",
        "seconds_per_step": full["seconds_per_step"],
        "all_cores": full, "threads_8": eight,
    }


def mse_vs_oracle(eng, x, n_rows: int = 2048):
    """The "+ recon-MSE" half of the metric on the driver-visible line: the bench's own parameters (as they are after the
    timed steps) and `n_rows` rows of its own data through the HIP forward and through the CPU oracle's forward
    (eval mode: no renormalisation, no tracker); north_star asks for 1e-4 relative."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import sae_ref as R

    xs = x[:n_rows].contiguous()
    eng.set_prefixes(None)
    eng.step_forward(xs, training=False)
    got = eng.read_stats().mse
    cfg = R.RefConfig(d_model=D_MODEL, d_sae=D_SAE, top_k=TOP_K)
    params = {k: v.detach().cpu().clone() for k, v in eng.param_views().items()}
    with torch.no_grad():
        ref = R.objective_forward(params, xs.cpu(), cfg, toks_since_active=None, training=False).mse.item()
    return {"rows": n_rows, "mse_hip": got, "mse_oracle": ref, "mse_rel_err_vs_oracle": abs(got - ref) / ref,
            "mse_check_note": f"forward MSE of the HIP path and of the CPU oracle on {n_rows} rows of the bench's own batch with the bench's own "
                              "parameters, both reported in fp32: 0.0 means equal after rounding to fp32, not a full-batch statement "
                              "(tests/test_gpu_fullsize.py holds the full-size checks)"}


def other_config_record(dev, *, name, d, s, k, b, encoder, n_prefixes=1, steps=10, warmup=5):
    """One of BASELINE.json's other configurations at its own shape on this GPU: ms per step and the encoder kernel's
    share of its MFMA peak (same measurement as the headline: HIP events on the launch stream).  Synthetic N(mu, 1)
    activations, random-init weights, pool of 8 batches, lr ramp as in the headline."""
    import dataclasses

    from saev_amd.engine import EngineConfig, SaeEngine
    from saev_amd.nn.objectives import sample_prefixes

    ecfg = EngineConfig(d_model=d, d_sae=s, top_k=k, max_batch=b, aux_dead_cap=4096)
    if encoder:
        ecfg = dataclasses.replace(ecfg, encoder=encoder)
    eng = SaeEngine(ecfg, dev)
    g = torch.Generator(device=dev).manual_seed(42)
    W = (torch.rand(s, d, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / d)
    W /= W.norm(dim=1, keepdim=True)
    eng.view("W_dec").copy_(W)
    eng.view("W_enc").copy_(W.t())
    del W
    mu = torch.randn(d, device=dev, generator=g)
    pool = torch.randn(8 * b, d, device=dev, generator=g) + mu
    torch.manual_seed(7)  # the Matryoshka cut points come from torch's global CPU generator, as in the reference

    def one(i):
        if n_prefixes > 1:
            eng.set_prefixes(sample_prefixes(s, n_prefixes))
        eng.train_step(pool[(i % 8) * b:(i % 8 + 1) * b], 4e-4 * min(1.0, i / 500), 1.0)

    for i in range(warmup):
        one(i)
    eng.enable_kernel_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        one(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    enc_ms = eng.encoder_ms()
    st = eng.read_stats()
    peak = F32_MFMA_PEAK_TFLOPS if eng.cfg.encoder == "f32" else F16_MFMA_PEAK_TFLOPS
    rec = {"config": name, "d_model": d, "d_sae": s, "top_k": k, "batch": b, "encoder": eng.cfg.encoder, "n_prefixes": n_prefixes,
           "steps": steps, "ms_per_step": dt / steps * 1e3, "activations_per_sec": b * steps / dt,
           "encoder_kernel_ms": enc_ms, "encoder_tflops": 2.0 * b * d * s / (enc_ms * 1e-3) / 1e12 if enc_ms > 0 else None,
           "encoder_frac_of_peak": (2.0 * b * d * s / (enc_ms * 1e-3) / 1e12 / peak) if enc_ms > 0 else None,
           "mse_last": st.mse, "n_overflow_rows": st.n_overflow_rows, "cand_max": st.cand_max, "dense_route": st.dense_route}
    eng.close()
    del eng, pool
    torch.cuda.empty_cache()
    return rec


def sweep_group_record(dev, n_saes=4, steps=20, warmup=8):
    """The reference's main training mode (framework/train.py:334-348; grouping :669-695): `n_saes` SAEs of configs[1]'s shape on
    the SAME batches, one after the other per batch, the first building what a step derives from x alone for all of them
    (saev_share_x).  Reports the cost of a batch through the whole group, per SAE, beside one SAE alone on the same loop
    (from random init, so both sides are in the same regime: compare with `single_ms_per_step`, not with the headline)."""
    import dataclasses

    from saev_amd.engine import EngineConfig, SaeEngine

    ecfg = EngineConfig(d_model=D_MODEL, d_sae=D_SAE, top_k=TOP_K, max_batch=BATCH)
    pool = synthetic_pool(dev, "mean", 8 * BATCH, D_MODEL)
    x = torch.empty(BATCH, D_MODEL, device=dev)

    def run(n):
        engs = []
        for j in range(n):
            e = SaeEngine(ecfg, dev)
            g = torch.Generator(device=dev).manual_seed(42 + j)
            W = (torch.rand(D_SAE, D_MODEL, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D_MODEL)
            W /= W.norm(dim=1, keepdim=True)
            e.view("W_dec").copy_(W)
            e.view("W_enc").copy_(W.t())
            del W
            if j:
                e.share_x(engs[0])
            engs.append(e)
        rows = torch.arange(BATCH, device=dev)

        def one(i):
            lr = 4e-4 * min(1.0, i / 500)
            r = rows + (i % 8) * BATCH
            if n == 1:
                engs[0].train_step_gather(pool, r, lr, 1.0, out=x)
                return
            engs[0].gather_rows(pool, r, out=x)
            for e in engs:
                e.train_step(x, lr, 1.0)

        for i in range(warmup):
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            one(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        for e in reversed(engs):
            e.close()
        del engs
        torch.cuda.empty_cache()
        return dt

    single = run(1)
    group = run(n_saes)
    return {"n_saes": n_saes, "steps": steps, "group_ms_per_batch": group, "ms_per_sae": group / n_saes, "single_ms_per_step": single,
            "per_sae_over_single": group / n_saes / single,
            "note": "n_saes SAEs of configs[1]'s shape trained on the same batches (saev_share_x), first steps from random init; "
                    "`single_ms_per_step` is one SAE alone on the same loop"}


def vendor_gemm_record(dev, b=BATCH, d=D_MODEL, s=D_SAE, launches=200):
    """Anchor for the encoder's first pass (measurement only -- nothing in the product path calls a library GEMM): what the vendor
    fp16 GEMM (torch.matmul -> hipBLASLt) reaches on THIS box for the encoder's own contraction, `launches` back-to-back launches
    timed with HIP events so that the power controller has settled; random operands at the scale of the encoder's images, both
    layouts of W, and all-zero operands (same instruction stream, clock not power-limited: the gap is the power limit)."""
    out = {"shape": [b, d, s], "launches": launches, "dtype": "fp16 operands, fp32 accumulate, fp16 output (1 GB written; the encoder's fused "
                                                              "TopK epilogue writes ~8 KB of candidates per row instead)"}
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.randn(b, d, device=dev, generator=g) * 4096).half()
    w = (torch.randn(d, s, device=dev, generator=g) * 256).half()
    y = torch.empty(b, s, device=dev, dtype=torch.float16)
    for name, xo, wo in (("random_nn", x, w), ("random_nt", x, w.t().contiguous().t()), ("zeros_nt", torch.zeros_like(x), torch.zeros_like(w).t().contiguous().t())):
        for _ in range(20):
            torch.matmul(xo, wo, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(launches):
            torch.matmul(xo, wo, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / launches
        out[name] = {"ms": ms, "tflops": 2.0 * b * d * s / ms / 1e9}
    del x, w, y
    torch.cuda.empty_cache()
    out["tflops"] = max(out["random_nn"]["tflops"], out["random_nt"]["tflops"])
    return out


def synthetic_pool(dev, kind, n_rows, d, seed=17):
    """Activation pools of the three data regimes of SURVEY.md 8d.  "mean": x = z + mu with a per-dimension mean (the throughput
    data: the headline); "isotropic": x = z; "lowrank": x = A s + 0.1 eps with A (d x 4d) unit columns and s 16-sparse with exp(1)
    magnitudes (the parity data: TopK is meaningful, some latents die)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    if kind == "mean":
        return torch.randn(n_rows, d, device=dev, generator=g) + torch.randn(d, device=dev, generator=torch.Generator(device=dev).manual_seed(17))
    if kind == "isotropic":
        return torch.randn(n_rows, d, device=dev, generator=g)
    assert kind == "lowrank"
    A = torch.randn(4 * d, d, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    A /= A.norm(dim=1, keepdim=True)
    out = torch.empty(n_rows, d, device=dev)
    for lo in range(0, n_rows, 8192):
        n = min(8192, n_rows - lo)
        idx = torch.randint(0, 4 * d, (n, 16), device=dev, generator=g)
        mag = -torch.log(torch.rand(n, 16, device=dev, generator=g).clamp_min(1e-12))
        out[lo:lo + n] = torch.einsum("nk,nkd->nd", mag, A[idx]) + 0.1 * torch.randn(n, d, device=dev, generator=g)
    return out


def data_regime_record(dev, kind, *, pretrain=1500, steps=40, pool_batches=16):
    """configs[1] on another data regime: the same loop as the headline (draw from a pool + train step), steady state."""
    from saev_amd.engine import EngineConfig, SaeEngine

    eng = SaeEngine(EngineConfig(d_model=D_MODEL, d_sae=D_SAE, top_k=TOP_K, max_batch=BATCH, aux_dead_cap=4096), dev)
    g = torch.Generator(device=dev).manual_seed(42)
    W = (torch.rand(D_SAE, D_MODEL, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D_MODEL)
    W /= W.norm(dim=1, keepdim=True)
    eng.view("W_dec").copy_(W)
    eng.view("W_enc").copy_(W.t())
    del W
    pool = synthetic_pool(dev, kind, pool_batches * BATCH, D_MODEL)
    perm = torch.randperm(pool.shape[0], device=dev, generator=g)
    x = torch.empty(BATCH, D_MODEL, device=dev)

    def one(i):
        eng.train_step_gather(pool, perm[(i % pool_batches) * BATCH:(i % pool_batches + 1) * BATCH], 4e-4 * min(1.0, i / 500), 1.0, out=x)

    for i in range(pretrain):
        one(i)
    eng.enable_kernel_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(pretrain, pretrain + steps):
        one(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.read_stats()
    rec = {"data": kind, "pretrain_steps": pretrain, "steps": steps, "ms_per_step": dt / steps * 1e3, "activations_per_sec": BATCH * steps / dt,
           "encoder_kernel_ms": eng.encoder_ms(), "mse_last": st.mse, "n_dead_last": st.n_dead, "aux_route_last": eng.aux_route(),
           "dense_route": st.dense_route, "cand_max": st.cand_max}
    eng.close()
    del eng, pool
    torch.cuda.empty_cache()
    return rec


def train_e2e_record(dev, *, steps=400, pool_batches=32):
    """What a USER of the package gets: framework.train.train() (loader, limiter, lr schedule, per-step Python) on a resident pool
    at configs[1]'s shape against the bare engine loop on the same pool from the same initial weights, both from random init."""
    import saev_amd.utils.scheduling as sched
    from saev_amd import data, nn
    from saev_amd.engine import EngineConfig, SaeEngine
    from saev_amd.framework import train as T
    from saev_amd.nn import modeling, objectives

    pool = synthetic_pool(dev, "mean", pool_batches * BATCH, D_MODEL)
    dcfg = data.ShuffledConfig(batch_size=BATCH, seed=17)
    cfg = T.Config(train_data=dcfg, val_data=dcfg, n_train=steps * BATCH, n_val=BATCH,
                   sae=nn.SparseAutoencoderConfig(d_model=D_MODEL, d_sae=D_SAE, reinit_blend=0.0, activation=modeling.TopK(top_k=TOP_K)),
                   objective=objectives.Matryoshka(n_prefixes=1), log_every=10**9, track=False, runs_root=pathlib.Path("/tmp/saev_bench_runs"))
    t_loop = {}
    orig_iter = sched.BatchLimiter.__iter__

    def timed_iter(self, _orig=orig_iter, _t=t_loop):
        torch.cuda.synchronize()
        _t["t0"] = time.perf_counter()
        yield from _orig(self)

    sched.BatchLimiter.__iter__ = timed_iter
    try:
        saes, objs, run, n_steps = T.train([cfg], train_pool=pool)
    finally:
        sched.BatchLimiter.__iter__ = orig_iter
    torch.cuda.synchronize()
    dt_train = time.perf_counter() - t_loop["t0"]
    del saes, objs
    torch.cuda.empty_cache()
    # the bare loop: same shapes, same lr ramp, same pool
    eng = SaeEngine(EngineConfig(d_model=D_MODEL, d_sae=D_SAE, top_k=TOP_K, max_batch=BATCH), dev)
    torch.manual_seed(cfg.seed)
    ref = nn.SparseAutoencoder(cfg.sae)
    eng.load_params({k: v.detach() for k, v in ref.state_dict().items()})
    g = torch.Generator(device=dev).manual_seed(1)
    perm = torch.randperm(pool.shape[0], device=dev, generator=g)
    x = torch.empty(BATCH, D_MODEL, device=dev)
    sched_lr = sched.WarmupCosine(0.0, cfg.n_lr_warmup, cfg.lr, n_steps, 0.0)
    lr = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_steps):
        eng.train_step_gather(pool, perm[(i % pool_batches) * BATCH:(i % pool_batches + 1) * BATCH], lr, cfg.grad_clip, out=x)
        lr = sched_lr.step()
    torch.cuda.synchronize()
    dt_loop = time.perf_counter() - t0
    eng.close()
    del eng, pool
    torch.cuda.empty_cache()
    return {"steps": n_steps, "train_ms_per_step": dt_train / n_steps * 1e3, "train_activations_per_sec": BATCH * n_steps / dt_train,
            "engine_loop_ms_per_step": dt_loop / n_steps * 1e3, "train_over_engine_loop": dt_loop / dt_train,
            "feed": "resident pool of 32 batches, seeded permutation per epoch (data.ShuffledDataLoader), batch drawn inside the step",
            "note": "framework.train.train() end to end (first steps from random init, no log step in the window) against the bare "
                    "SaeEngine.train_step_gather loop on the same pool; tools/bench_train_e2e.py has the streaming feed from a shard directory"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--encoder", choices=["f32", "f16x3", "bf16", "f16r"], default=None, help="encoder arithmetic (default: engine default)")
    ap.add_argument("--overlap", action="store_true",
                    help="bucketed gradient all-reduce overlapped with the backward instead of one flat all-reduce after it")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the data-parallel code path (RCCL init + per-step collectives) even with one rank")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank "
                                                      "code path with several ranks on ONE device, see --same-device)")
    ap.add_argument("--same-device", action="store_true", help="every rank uses cuda:0 (rehearsal on a one-GPU box; needs --backend gloo)")
    ap.add_argument("--tail", choices=["auto", "replicated", "sharded"], default="auto",
                    help="data-parallel optimizer tail: every rank all of it after an all-reduce, or reduce-scatter -> 1/N of "
                         "the tail per rank -> all-gather (framework/ddp.py).  auto: sharded when a start-up self-check on a "
                         "small problem reproduces the all-reduce path on every rank, else replicated")
    ap.add_argument("--exchange", choices=["dense", "sparse", "auto"], default="dense",
                    help="what the ranks exchange per step: the gradient (all-reduce, or reduce-scatter / all-gather with --tail "
                         "sharded), or the sparse step state -- x, dL/dx_hat, codes all-gathered, every rank runs the backward "
                         "over the global batch (framework/ddp.py; for small per-rank batches, i.e. strong scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: this many rows per step in total, split evenly over the ranks (overrides --batch; "
                         "configs[2] at BASELINE's global batch: 16384)")
    ap.add_argument("--pretrain-steps", type=int, default=1500,
                    help="train this many steps from random init before the warm-up and the timed steps, so that the timed region "
                         "is the steady-state step: past the 10 M-token dead-latent threshold (611 steps), the lr warm-up (500) and "
                         "the ~1 000 steps over which latent usage keeps spreading and the step keeps slowing down "
                         "(tools/experiments/long_run_probe.py: 2.8 ms at step 100, 3.07 at 700, 3.25-3.35 from 1 200 to 8 000); the "
                         "first 5 + 20 of them are timed as `from_random_init`.  0 = time from random init (rounds 1-3)")
    ap.add_argument("--sustained-steps", type=int, default=2000,
                    help="length of a long segment timed after the headline steps (0 = skip): the same loop, continued")
    ap.add_argument("--sustained-after", type=int, default=0, help="further untimed steps between the headline and that segment")
    ap.add_argument("--no-auxk-probe", action="store_true", help="skip the AuxK-active sub-records (forced dead sets)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` sub-records (configs[0], configs[3] in bf16, configs[1] with 10 Matryoshka prefixes)")
    ap.add_argument("--extract-e2e", action="store_true",
                    help="add the `extract_e2e` sub-record: configs[4] on one GPU -- ViT-L/14-shaped transformer forward (random init, "
                         "bf16 autocast) -> hooks -> device reservoir -> SAE train steps, no disk in between")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the `vendor_gemm`, `data_regimes` and `train_e2e` sub-records (profiling runs)")
    ap.add_argument("--no-busbw", action="store_true", help="skip the `collectives` sub-record (RCCL bus bandwidth of the step's collectives)")
    ap.add_argument("--n-saes", type=int, default=1,
                    help="train this many SAEs on every batch (the reference's parallel groups); the extra ones differ in "
                         "parameters only and share the first one's x statistics / operand images.  value still counts each batch once")
    ap.add_argument("--shape", default=None,
                    help="d_model,d_sae,top_k of a TOY run (rehearsals of the multi-rank code path on one device, e.g. "
                         "--shape 256,2048,16 --batch 256 --backend gloo --same-device): the line then says so in config.workload "
                         "and carries none of the sub-records that are sized for configs[1]")
    args = ap.parse_args()
    global D_MODEL, D_SAE, TOP_K
    toy = args.shape is not None
    if toy:
        D_MODEL, D_SAE, TOP_K = (int(v) for v in args.shape.split(","))
        args.no_other_configs = args.no_extras = args.no_auxk_probe = args.no_cpu_baseline = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        from saev_amd.framework.ddp import init_distributed

        # (timeouts on every collective, abort on a failed rank: framework/ddp.py)
        dist = init_distributed(args.backend, rank=rank, world_size=world, device=dev if args.backend == "nccl" else None)

    from saev_amd.engine import EngineConfig, SaeEngine
    from saev_amd.framework.ddp import DataParallelStepper, choose_exchange, collective_busbw

    B = args.batch
    strong = args.global_batch > 0
    if strong:
        assert args.global_batch % world == 0, "--global-batch must divide evenly over the ranks"
        B = args.global_batch // world
    # Which exchange runs (framework/ddp.py: choose_exchange): `--tail auto` takes the sharded tail when a start-up self-check
    # on a small SAE reproduces the all-reduce path on every rank, the replicated one otherwise; `--exchange auto` picks the
    # sparse step state for per-rank batches of at most 4 096 rows, verified the same way; explicit choices are honoured.
    tail_mode, exchange_mode, exchange_report = "replicated", "dense", {}
    if dist is not None and not args.overlap:
        tail_mode, exchange_mode, exchange_report = choose_exchange(
            dist, world, rank, dev, B, tail=args.tail if args.exchange == "dense" else "auto", exchange=args.exchange)

    sparse = dist is not None and exchange_mode == "sparse"
    # The reference's dead-latent threshold, 10 M tokens (objectives.py:24), is 611 of these steps: the sustained segment
    # below starts after it, so it runs in the regime a real run spends its life in (tracker consulted every step, AuxK on
    # whatever is dead).
    dead_thr = 10_000_000
    # (the sparse-state exchange runs the backward over every rank's rows: scratch for the global batch)
    ecfg = EngineConfig(d_model=D_MODEL, d_sae=D_SAE, top_k=TOP_K, max_batch=B, max_backward_rows=B * world if sparse else 0,
                        dead_threshold_tokens=dead_thr, shard_world=world if tail_mode == "sharded" else 1)
    if args.encoder:
        import dataclasses

        ecfg = dataclasses.replace(ecfg, encoder=args.encoder)
    eng = SaeEngine(ecfg, dev)
    # random-init weights of the reference architecture (modeling.py:306-329), model seed 42
    g = torch.Generator(device=dev).manual_seed(42)
    W = (torch.rand(D_SAE, D_MODEL, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D_MODEL)
    W /= W.norm(dim=1, keepdim=True)
    eng.view("W_dec").copy_(W)
    eng.view("W_enc").copy_(W.t())
    del W
    # synthetic activation pool: x = z + mu, generator seed 17 (+rank)
    g = torch.Generator(device=dev).manual_seed(17 + rank)
    mu = torch.randn(D_MODEL, device=dev, generator=torch.Generator(device=dev).manual_seed(17))
    pool = torch.randn(POOL_BATCHES * B, D_MODEL, device=dev, generator=g) + mu
    perm = torch.randperm(pool.shape[0], device=dev, generator=g)
    x = torch.empty(B, D_MODEL, device=dev)
    stepper = DataParallelStepper(eng, dist, world, force=args.force_dist, overlap=args.overlap, tail=tail_mode,
                                  exchange="sparse" if sparse else "dense")
    extra = []  # further SAEs of the group (--n-saes): same batches, their own parameters
    for j in range(1, args.n_saes):
        import dataclasses as _dc

        e2 = SaeEngine(ecfg, dev)
        e2.params.copy_(eng.params)
        e2.share_x(eng)
        extra.append(DataParallelStepper(e2, dist, world, force=args.force_dist, overlap=args.overlap, tail=tail_mode,
                                         exchange="sparse" if sparse else "dense"))
    lr_sched = lambda i: 4e-4 * min(1.0, i / 500)  # noqa: E731  warm-up region of the reference schedule

    def one_step(i):
        rows = perm[(i % POOL_BATCHES) * B : (i % POOL_BATCHES + 1) * B]
        if stepper.dist is None and not extra:
            # one GPU, one SAE: the draw rides in the step's first kernel (saev_train_step_gather; x receives the batch)
            eng.train_step_gather(pool, rows, lr_sched(i), 1.0, out=x)
            return
        eng.gather_rows(pool, rows, out=x)
        stepper.train_step(x, lr_sched(i), 1.0)
        for st2 in extra:
            st2.train_step(x, lr_sched(i), 1.0)

    route_hist = {}

    def timed(first, n, routes=False):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(first, first + n):
            one_step(i)
            if routes:  # (a host-side field of the context: no synchronisation)
                r = eng.aux_route()
                route_hist[r] = route_hist.get(r, 0) + 1
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t = time.perf_counter() - t
        if dist is not None:
            tt = torch.tensor([t], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = tt.item()
        return t

    # ---- state preparation: from random init into the regime a run lives in (the first 5 + 20 steps timed on the way) ----
    step_i = 0
    early = None
    if args.pretrain_steps > 0:
        n_w, n_t = min(5, args.pretrain_steps), min(20, max(0, args.pretrain_steps - 5))
        for i in range(n_w):
            one_step(i)
        step_i = n_w
        if n_t > 0:
            eng.enable_kernel_timing(True)
            t_e = timed(step_i, n_t)
            step_i += n_t
            early = {"ms_per_step": t_e / n_t * 1e3, "activations_per_sec": B * world * n_t / t_e, "warmup": n_w, "steps": n_t,
                     "encoder_kernel_ms": eng.encoder_ms(),
                     "note": "the first steps from random init (the timed region of rounds 1-3): ~19 000 of 32 768 latents still unused, "
                             "nothing near the dead-latent threshold, so no AuxK work"}
            eng.enable_kernel_timing(False)
        for i in range(step_i, args.pretrain_steps):
            one_step(i)
        step_i = max(step_i, args.pretrain_steps)
    # ---- the contract's region: W untimed warm-up steps, then exactly K timed steps ---------------------------------------
    first_warm = step_i
    for i in range(step_i, step_i + args.warmup):
        one_step(i)
    step_i += args.warmup
    eng.enable_kernel_timing(True)
    dt = timed(step_i, args.steps)
    step_i += args.steps
    enc_ms = eng.encoder_ms()
    stats = eng.read_stats()
    aux_route_timed = eng.aux_route()
    eng.enable_kernel_timing(False)

    # ---- steady state: a long segment well past the dead-latent threshold -----------------------------------------
    sustained = None
    if args.sustained_steps > 0:
        for i in range(step_i, step_i + args.sustained_after):
            one_step(i)
        step_i += args.sustained_after
        rb0 = eng.dead_readbacks()
        t_sus = timed(step_i, args.sustained_steps, routes=True)
        step_i += args.sustained_steps
        st_s = eng.read_stats()
        sustained = {
            "ms_per_step": t_sus / args.sustained_steps * 1e3, "activations_per_sec": B * world * args.sustained_steps / t_sus,
            "steps": args.sustained_steps, "first_step": step_i - args.sustained_steps,
            "dead_threshold_tokens": dead_thr, "tokens_seen_at_start": (step_i - args.sustained_steps) * B * world,
            "n_dead_last": st_s.n_dead, "aux_last": st_s.aux, "mse_last": st_s.mse, "aux_route_last": eng.aux_route(),
            "n_dead_readbacks_in_segment": eng.dead_readbacks() - rb0,
            "aux_route_steps": {{0: "no auxiliary work", 1: "few-dead-latents kernels", 2: "few-dead-latents kernels after a read-back",
                                 3: "dense algebra"}.get(r_, str(r_)): n_ for r_, n_ in sorted(route_hist.items())},
            "note": "the headline's loop, continued for a long window.  A TRAJECTORY figure: the dead count drifts between 0 and ~35 with "
                    "bursts of 100-240 latents every ~600 steps (dense-algebra steps, +0.3 ... +0.6 ms each), and one ulp on one parameter at "
                    "step 700 moves this mean by +-0.03 ms on one box (tools/experiments/r6_chaos_probe.py, "
                    "profiles/r06_sustained_chaos_probe_six.txt)",
        }

    # ---- AuxK active on a forced dead set (configs[2]'s single-GPU half) --------------------------------------------
    auxk_active = None
    if not args.no_auxk_probe and eng.cfg.k_aux > 0:
        auxk_active = []
        gsel = torch.Generator(device=dev).manual_seed(99)
        order = torch.randperm(D_SAE, device=dev, generator=gsel)
        for nd in (8, 24, 48, 1000):  # one-pass kernel / fp32-MFMA kernels with one and two latent blocks / dense algebra
            sel = order[:nd]
            eng.view("b_enc")[sel] = -100.0  # never selected by the main path: they stay dead
            toks = torch.zeros(D_SAE, dtype=torch.int64, device=dev)
            toks[sel] = dead_thr
            eng.set_tracker(toks)
            for i in range(step_i, step_i + 10):
                one_step(i)
            step_i += 10
            rb0 = eng.dead_readbacks()
            t_a = timed(step_i, 30)
            step_i += 30
            st_a = eng.read_stats()
            auxk_active.append({"n_dead_forced": nd, "n_dead": st_a.n_dead, "ms_per_step": t_a / 30 * 1e3, "steps": 30,
                                "aux": st_a.aux, "aux_route": eng.aux_route(), "n_dead_readbacks": eng.dead_readbacks() - rb0})

    # ---- the other BASELINE configurations at their own shapes (single GPU; extra keys, the headline stays configs[1]) ----
    other_configs = None
    if world == 1 and not args.no_other_configs and B == BATCH:
        other_configs = [
            other_config_record(dev, name="configs[0]: d_in=768, 8x, k=32, batch=4096", d=768, s=6144, k=32, b=4096, encoder=args.encoder),
            other_config_record(dev, name="configs[3] (one GPU's share): d_in=1280, 64x (81 920 latents), k=64, bf16, batch=16384",
                                d=1280, s=81920, k=64, b=16384, encoder="bf16"),
            other_config_record(dev, name="configs[1] with the reference's default objective: 10 Matryoshka prefixes sampled per step",
                                d=D_MODEL, s=D_SAE, k=TOP_K, b=BATCH, encoder=args.encoder, n_prefixes=10),
        ]
    # ---- the step's own collectives at their real sizes (N > 1): which side of the dense / sparse crossover is this node on? ----
    collectives = None
    if dist is not None and args.backend == "nccl" and not args.no_busbw:
        sparse_rows = min(B, 2048)  # configs[2]'s per-rank share of a 16 384-row global batch
        collectives = collective_busbw(dist, world, dev, eng.n_params, eng.chunk_a if eng.cfg.shard_world > 1 else (D_SAE + 1 + world - 1) // world * D_MODEL,
                                       eng.chunk_b if eng.cfg.shard_world > 1 else ((D_MODEL * D_SAE + D_SAE + world - 1) // world + 3) // 4 * 4,
                                       sparse_bytes_per_rank=sparse_rows * (8 * D_MODEL + 8 * TOP_K))
        collectives["note"] = ("RCCL on the live communicator, mean of 5 after 2 warm-up calls, max over ranks; busbw = algbw x 2(n-1)/n (all-reduce) "
                               "or x (n-1)/n (reduce-scatter / all-gather); the sparse step state is sized for 2 048 rows per rank")
    vendor_gemm = data_regimes = train_e2e = sweep_group = None
    if world == 1 and not args.no_extras and B == BATCH:
        vendor_gemm = vendor_gemm_record(dev)
        data_regimes = [data_regime_record(dev, kind) for kind in ("isotropic", "lowrank")]
        train_e2e = train_e2e_record(dev)
        sweep_group = sweep_group_record(dev)
    extract_e2e = None
    if world == 1 and args.extract_e2e:
        from tools.bench_extract_e2e import run as extract_run

        extract_e2e = extract_run(dev)

    if rank == 0:
        flops = 2.0 * B * D_MODEL * D_SAE
        achieved = flops / (enc_ms * 1e-3) / 1e12 if enc_ms > 0 else None
        f16x3 = eng.cfg.encoder == "f16x3"
        peak = F32_MFMA_PEAK_TFLOPS if eng.cfg.encoder == "f32" else F16_MFMA_PEAK_TFLOPS
        kernel_name = {"f16x3": "encode_f16x3_kernel<EPI_TOPK,32,3> (3 x v_mfma_f32_32x32x16_f16 per fp32 product)",
                       "bf16": "encode_m16_kernel<1> (v_mfma_f32_16x16x32_bf16)",
                       "f16r": "encode_m16_kernel<2> (v_mfma_f32_16x16x32_f16 first pass; exact fp32 refinement in select)",
                       "f32": "encode_gemm_kernel<EPI_TOPK> (v_mfma_f32_32x32x2_f32)"}[eng.cfg.encoder]
        dtype_name = {"f16x3": "f32 (encoder products as 3 x f16 MFMA on fp16 hi/lo splits, fp32 accumulate; all else fp32)",
                      "bf16": "bf16 encoder operands, fp32 accumulate; all else fp32 (NOT the headline precision)",
                      "f16r": "f32 (encoder: fp16 MFMA first pass with a rigorous error margin, survivors recomputed exactly in fp32; all else fp32)",
                      "f32": "f32"}[eng.cfg.encoder]
        roof = {"bound": "mfma",
                "kernel": kernel_name,
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": (achieved / peak) if achieved else None,
                "kernel_ms": enc_ms, "traffic": None}
        # HBM/fabric bytes per launch of that kernel come from rocprofv3 PMC passes (they cannot be collected inside the
        # timed run); the committed summaries are the source, and they only apply to the shape/encoder they were taken on
        tfile = {"f16x3": ROOT / "profiles" / "r01_d_encoder_traffic.json"}.get(eng.cfg.encoder)
        if eng.cfg.encoder == "f16r":  # (the newest committed PMC summary of this kernel)
            tfile = next((ROOT / "profiles" / f"r0{r}_encoder_traffic.json" for r in (6, 5, 4, 3, 2)
                          if (ROOT / "profiles" / f"r0{r}_encoder_traffic.json").exists()), None)
        # what a register-resident loop of the same MFMA sustains on random fp16 operands (tools/ubench/mfma_issue.hip,
        # profiles/r02_mfma_issue.txt): the matrix pipes are clock-limited by power on real data
        roof["power_limited_mfma_ceiling_tflops"] = 1930.0 if eng.cfg.encoder in ("f16r", "bf16") else 1690.0  # 16x16x32 / 32x32x16
        if tfile is not None and B == BATCH and tfile.exists():
            tj = json.loads(tfile.read_text())
            roof["traffic"] = tj["traffic_bytes_per_launch"]
            roof["traffic_unit"] = "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"
            roof["traffic_source"] = tj["source"]
            op_bytes = 4.0 if f16x3 else 2.0  # bytes per operand element in the staged images (hi+lo fp16 / single fp16)
            roof["algorithmic_bytes"] = op_bytes * (B * D_MODEL + D_MODEL * D_SAE) + 8.0 * B * 1000  # operands once + ~1k candidates/row
        if f16x3 and achieved:
            roof["executed_tflops"] = 3 * achieved
            roof["executed_frac"] = 3 * achieved / peak
        if eng.cfg.encoder == "f16r":
            roof["note"] = ("first pass only (one fp16 MFMA per product, flops = 2*B*D*S); the exact fp32 refinement of the "
                            "~45 survivors per row (refine_slices_kernel + refine_sum_kernel, 0.29 ms) and the selects are separate kernels inside ms_per_step")
        out = {
            "metric": "activations/sec (train step), d_in=1024 x32 k=32",
            "value": B * world * args.steps / dt,
            "unit": "activations/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": dtype_name,
            "data": "synthetic",
            "config": {"workload": ("TOY SHAPE (rehearsal, not BASELINE's workload): " if toy else "configs[1]: ") +
                                   f"d_in={D_MODEL}, d_sae={D_SAE} ({D_SAE // D_MODEL}x), k={TOP_K}, batch={B}/GPU, "
                                   "TopK SAE train step incl. AuxK bookkeeping + Adam, pool of 64 batches",
                       "global_batch": B * world, "parallelism": f"dp{world}", "encoder": eng.cfg.encoder, "n_saes": args.n_saes,
                       "weight_gradients": ("whole-row gathers (dw_rows)" if os.environ.get("SAEV_AMD_DW") == "rows" or D_MODEL % 32 != 0
                                            or stepper.overlap
                                            else "32-column slices out of the XCD L2s (dw_slices)"),
                       "grad_exchange": ("none" if stepper.dist is None else
                                         ("no gradient crosses ranks: all-gather of x, dL/dx_hat and the codes ((8 D + 8 k) bytes per row), "
                                          "backward over the global batch on every rank, the auxiliary term's compact rows all-reduced, "
                                          "replicated tail" if stepper.exchange == "sparse" else
                                          "bucketed all-reduce overlapped with the backward" if stepper.overlap else
                                          ("reduce-scatter of the two gradient halves, tail on 1/N of the elements per rank, all-gather of "
                                           "the parameter halves (decoder half on a side stream); verified at start-up against the "
                                           "all-reduce path" if stepper.tail == "sharded" else "one flat all-reduce, replicated tail")))},
            "mse_last": stats.mse, "n_overflow_rows": stats.n_overflow_rows, "cand_max": stats.cand_max,
            "topk_bounds": dict(eng.bound_state(), mode=eng.cfg.bounds,
                                note="predicted row bounds verified by the select stage; `repeats` launches (of `launches`, whole "
                                     "run) had a failed prediction and were re-run with guaranteed bounds on the device"),
            "roofline": roof,
            "timed_region": {"first_step": first_warm + args.warmup, "pretrain_steps": args.pretrain_steps,
                             "tokens_seen_at_start": (first_warm + args.warmup) * B * world, "dead_threshold_tokens": dead_thr,
                             "n_dead_last": stats.n_dead, "aux_route_last": aux_route_timed,
                             "note": (f"value = {args.steps} steps after {args.warmup} warm-up steps, both AFTER {args.pretrain_steps} training steps "
                                      "from random init (state preparation: past the dead-latent threshold and the lr warm-up) -- the "
                                      "steady-state train step of SURVEY.md 8d; `from_random_init` has the early-training figure"
                                      if args.pretrain_steps > 0 else
                                      f"value = {args.steps} steps after {args.warmup} warm-up steps from random init (--pretrain-steps 0)")},
        }
        if early is not None:
            out["from_random_init"] = early
        if sustained is not None:
            out["sustained_ms_per_step"] = sustained["ms_per_step"]
            out["sustained"] = sustained
        if auxk_active is not None:
            out["auxk_active"] = auxk_active
        if other_configs is not None:
            out["other_configs"] = other_configs
        if vendor_gemm is not None:
            roof["vendor_gemm_tflops"] = vendor_gemm["tflops"]
            roof["vendor_gemm"] = vendor_gemm
        if data_regimes is not None:
            out["data_regimes"] = data_regimes
        if train_e2e is not None:
            out["train_e2e"] = train_e2e
        if sweep_group is not None:
            out["sweep_group"] = sweep_group
        if extract_e2e is not None:
            out["extract_e2e"] = extract_e2e
        if collectives is not None:
            out["collectives"] = collectives
        if exchange_report:
            out["exchange_selection"] = exchange_report
        if world == 1 and not args.no_cpu_baseline:
            out.update(mse_vs_oracle(eng, x))
            out["cpu_baseline"] = cpu_baseline()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    sys.stderr.flush()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        # RCCL prints a version banner on stdout when the process exits normally; leave without running exit handlers so
        # that the result line stays the last thing on stdout (everything is flushed, the process group is gone)
        os._exit(0)


if __name__ == "__main__":
    main()
