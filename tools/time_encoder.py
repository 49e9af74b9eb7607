"""Time the fused encoder + TopK kernel alone on the configs[1] shape (forward in eval mode, kernel timing via HIP events)."""
import sys

import torch

sys.path.insert(0, ".")
from saev_amd.engine import EngineConfig, SaeEngine  # noqa: E402

D, S, B = 1024, 32768, 16384
eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=32, k_aux=0, max_batch=B, encoder=(sys.argv[1] if len(sys.argv) > 1 else "f16x3")),
                torch.device("cuda:0"))
g = torch.Generator(device="cuda").manual_seed(0)
W = (torch.rand(S, D, device="cuda", generator=g) * 2 - 1) * (6.0 / D) ** 0.5
W /= W.norm(dim=1, keepdim=True)
eng.view("W_dec").copy_(W)
eng.view("W_enc").copy_(W.t())
x = torch.randn(B, D, device="cuda", generator=g) + torch.randn(D, device="cuda", generator=g)
eng.enable_kernel_timing(True)
ts = []
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    eng.step_forward(x, training=False)
    torch.cuda.synchronize()
    ts.append(eng.encoder_ms())
ts = ts[4:]
st = eng.read_stats()
print("cand_max", st.cand_max, "overflow rows", st.n_overflow_rows, "dense route", st.dense_route)
print("encoder ms:", " ".join(f"{t:.3f}" for t in ts[:8]), " median %.3f min %.3f" % (sorted(ts)[len(ts) // 2], min(ts)))
