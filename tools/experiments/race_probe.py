"""Repeat the fused encoder on one input and compare the codes bit for bit (a missing wait shows up as a rare mismatch).
   PYTHONPATH=$PWD python tools/experiments/race_probe.py [mode] [repeats]"""
import math
import sys

import torch

sys.path.insert(0, ".")
from saev_amd.engine import EngineConfig, SaeEngine  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f16r"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for (D, S, B, K) in ((1024, 32768, 16384, 32), (768, 6144, 4096, 32), (1280, 8192, 5000, 64), (256, 1280, 700, 16)):
    eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, k_aux=0, max_batch=B, encoder=mode), torch.device("cuda:0"))
    g = torch.Generator(device="cuda").manual_seed(D)
    W = (torch.rand(S, D, device="cuda", generator=g) * 2 - 1) * math.sqrt(6.0 / D)
    eng.view("W_dec").copy_(W)
    eng.view("W_enc").copy_(W.t() + 0.01 * torch.randn(D, S, device="cuda", generator=g))
    x = torch.randn(B, D, device="cuda", generator=g) + torch.randn(D, device="cuda", generator=g)
    i0, v0 = eng.encode_topk(x)
    i0, v0 = i0.clone(), v0.clone()
    bad = 0
    for r in range(reps if B > 5000 else 3 * reps):
        i1, v1 = eng.encode_topk(x)
        if not (torch.equal(i0, i1) and torch.equal(v0, v1)):
            bad += 1
    print(f"{mode} D={D} S={S} B={B} k={K}: {bad} mismatching repeats", flush=True)
    del eng
