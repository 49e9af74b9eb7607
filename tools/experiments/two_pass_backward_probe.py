"""One-pass backward against the two-pass form a sharded data-parallel step uses (decoder gradient first, so that its
reduce-scatter can travel behind the encoder pass): time of the backward at configs[1], one GPU.
   PYTHONPATH=$PWD python tools/experiments/two_pass_backward_probe.py"""
import math
import sys
import time

import torch

sys.path.insert(0, ".")
from saev_amd.engine import EngineConfig, SaeEngine  # noqa: E402

D, S, B, K = 1024, 32768, 16384, 32
eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=B), torch.device("cuda:0"))
g = torch.Generator(device="cuda").manual_seed(0)
W = (torch.rand(S, D, device="cuda", generator=g) * 2 - 1) * math.sqrt(6.0 / D)
W /= W.norm(dim=1, keepdim=True)
eng.view("W_dec").copy_(W)
eng.view("W_enc").copy_(W.t())
x = torch.randn(B, D, device="cuda", generator=g) + torch.randn(D, device="cuda", generator=g)
eng.step_forward(x, training=True, n_rows_global=B)
eng.step_dead(B)


def one():
    eng.step_backward()


def two():
    eng.backward_begin()
    eng.backward_rows(0, S, 1)
    eng.backward_rows(0, S, 2)
    eng.backward_end()


for name, fn in (("one pass", one), ("two passes", two), ("one pass", one), ("two passes", two)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per backward", flush=True)
