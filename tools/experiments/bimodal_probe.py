import sys, gc
import torch
sys.path.insert(0, ".")
from saev_amd.engine import EngineConfig, SaeEngine
D, S, B = 1024, 32768, 16384
g = torch.Generator(device="cuda").manual_seed(0)
W = (torch.rand(S, D, device="cuda", generator=g) * 2 - 1) * (6.0 / D) ** 0.5
W /= W.norm(dim=1, keepdim=True)
x = torch.randn(B, D, device="cuda", generator=g) + torch.randn(D, device="cuda", generator=g)
junk = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=32, k_aux=0, max_batch=B, encoder="f16r"), torch.device("cuda:0"))
    eng.view("W_dec").copy_(W)
    eng.view("W_enc").copy_(W.t())
    eng.enable_kernel_timing(True)
    ts = []
    for i in range(30):
        eng.step_forward(x, training=False)
        torch.cuda.synchronize()
        ts.append(eng.encoder_ms())
    ts = ts[5:]
    print(f"engine {rep}: median {sorted(ts)[len(ts)//2]:.3f} min {min(ts):.3f}", flush=True)
    del eng
    gc.collect()
    torch.cuda.synchronize()
    junk.append(torch.empty((rep + 1) * 37 * 1024 * 1024 // 4, device="cuda"))
