"""Tail kernels of the two paths (saev_train_step's fused Adam vs the phases' rpg + transpose + flat Adam) at a given shape,
for a rocprofv3 --kernel-trace run:  python tools/experiments/r3_tail_shapes.py D S K B [encoder]"""
import sys, pathlib, math, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from saev_amd.engine import EngineConfig, SaeEngine
d, s, k, b = (int(v) for v in sys.argv[1:5])
enc = sys.argv[5] if len(sys.argv) > 5 else "f16r"
eng = SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=k, max_batch=b, aux_dead_cap=4096, encoder=enc))
g = torch.Generator(device="cuda").manual_seed(0)
W = (torch.rand(s, d, device="cuda", generator=g) * 2 - 1) * math.sqrt(6.0 / d)
W /= W.norm(dim=1, keepdim=True)
eng.view("W_dec").copy_(W); eng.view("W_enc").copy_(W.t()); del W
x = torch.randn(b, d, device="cuda", generator=g) + torch.randn(d, device="cuda", generator=g)
for i in range(8):
    eng.train_step(x, 1e-4, 1.0)
for i in range(8):
    eng.step_forward(x, training=True); eng.step_dead(b); eng.step_backward(); eng.step_tail(1e-4, 1.0)
torch.cuda.synchronize()
print("done")
