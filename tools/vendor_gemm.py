"""Anchor for the encoder's first pass: what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on
this box for the encoder's own contraction, 16 384 x 1 024 x 32 768 in fp16 / bf16 with fp32 accumulation, 200 back-to-back
launches timed with HIP events (so that the power controller has settled, as in tools/ubench/enc_loop2.hip).  Measurement only:
nothing in the product path calls a library GEMM.  Random operands at the scale of the encoder's images, and all-zero operands
(the clock then stays high: the gap between the two columns is the power limit, not the kernel).

    python tools/vendor_gemm.py [B D S]  -> one JSON line
"""
import json
import sys

import torch

B, D, S = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (16384, 1024, 32768)
dev = torch.device("cuda:0")
out = {"shape": [B, D, S], "launches": 200}
for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
    for fill in ("random", "zeros"):
        g = torch.Generator(device=dev).manual_seed(0)
        if fill == "random":
            x = (torch.randn(B, D, device=dev, generator=g) * 4096).to(dt)
            w = (torch.randn(D, S, device=dev, generator=g) * 256).to(dt)
        else:
            x = torch.zeros(B, D, device=dev, dtype=dt)
            w = torch.zeros(D, S, device=dev, dtype=dt)
        # both operand layouts a caller could hand over: W as (D, S) row-major, and as (S, D) row-major (the "NT" form)
        for layout, wop in (("nn", w), ("nt", w.t().contiguous().t())):
            y = torch.empty(B, S, device=dev, dtype=dt)
            for _ in range(20):
                torch.matmul(x, wop, out=y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                torch.matmul(x, wop, out=y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 200
            out[f"{name}_{fill}_{layout}"] = {"ms": round(ms, 4), "tflops": round(2.0 * B * D * S / ms / 1e9, 1)}
best = max(v["tflops"] for k, v in out.items() if isinstance(v, dict) and k.startswith("fp16_random"))
out["vendor_gemm_tflops"] = best
out["note"] = "output written as fp16 / bf16 (the library's fastest form; the encoder's epilogue never writes h at all)"
print(json.dumps(out))
