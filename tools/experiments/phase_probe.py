"""Per-phase step times (forward / dead / backward / tail) for several engine instances in one process: shows which
phases depend on where the allocator happened to put the buffers."""
import gc
import sys

import torch

sys.path.insert(0, ".")
from saev_amd.engine import EngineConfig, SaeEngine  # noqa: E402

D, S, B = 1024, 32768, 16384
g = torch.Generator(device="cuda").manual_seed(0)
W = (torch.rand(S, D, device="cuda", generator=g) * 2 - 1) * (6.0 / D) ** 0.5
W /= W.norm(dim=1, keepdim=True)
x = torch.randn(B, D, device="cuda", generator=g) + torch.randn(D, device="cuda", generator=g)
junk = []


def timed(fn, reps=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=32, k_aux=512, max_batch=B, encoder="f16r"), torch.device("cuda:0"))
    eng.view("W_dec").copy_(W)
    eng.view("W_enc").copy_(W.t())
    for _ in range(5):
        eng.train_step(x, 1e-4, 1.0)
    res = {}
    fw, dd, bw, tl = [], [], [], []
    for _ in range(8):
        fw.append(timed(lambda: eng.step_forward(x, training=True), 1))
        dd.append(timed(lambda: eng.step_dead(B), 1))
        bw.append(timed(lambda: eng.step_backward(), 1))
        tl.append(timed(lambda: eng.step_tail(1e-4, 1.0), 1))
    med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
    print(f"engine {rep}: forward {med(fw):.3f} dead {med(dd):.3f} backward {med(bw):.3f} tail {med(tl):.3f} ms", flush=True)
    del eng
    gc.collect()
    torch.cuda.synchronize()
    junk.append(torch.empty((rep + 1) * 37 * 1024 * 1024 // 4, device="cuda"))
