"""Does a memory-bound stream (Adam-like: 7 fp32 streams) hide behind the power-limited encoder?  One box:
   PYTHONPATH=$PWD python tools/experiments/overlap_probe.py
Times (a) the forward (encoder + selects + decode) alone, (b) an elementwise update of half the parameters alone,
(c) both enqueued one after the other on one stream, (d) on two streams."""
import sys
import time

import torch

sys.path.insert(0, ".")
from saev_amd.engine import EngineConfig, SaeEngine  # noqa: E402

D, S, B = 1024, 32768, 16384
eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=32, k_aux=0, max_batch=B, encoder="f16r"), torch.device("cuda:0"))
g = torch.Generator(device="cuda").manual_seed(0)
W = (torch.rand(S, D, device="cuda", generator=g) * 2 - 1) * (6.0 / D) ** 0.5
W /= W.norm(dim=1, keepdim=True)
eng.view("W_dec").copy_(W)
eng.view("W_enc").copy_(W.t())
x = torch.randn(B, D, device="cuda", generator=g) + torch.randn(D, device="cuda", generator=g)
n = S * D  # half the parameters
p, gr, m, v = (torch.randn(n, device="cuda") for _ in range(4))
side = torch.cuda.Stream()


def adam_like():
    # reads p, g, m, v; writes p, m, v (three torch kernels: ~10 streams instead of 7 -- an upper bound)
    m.mul_(0.9).add_(gr, alpha=0.1)
    v.mul_(0.999).addcmul_(gr, gr, value=0.001)
    p.addcdiv_(m, v.sqrt().add_(1e-8), value=-1e-4)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def fwd():
    eng.step_forward(x, training=False)


def both_serial():
    fwd()
    adam_like()


def both_overlap():
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        adam_like()
        e2 = torch.cuda.Event()
        e2.record()
    fwd()
    torch.cuda.current_stream().wait_event(e2)


a, b = timed(fwd), timed(adam_like)
c, d = timed(both_serial), timed(both_overlap)
print(f"forward alone {a:.3f} ms; update alone {b:.3f} ms; one stream {c:.3f} ms; two streams {d:.3f} ms")
