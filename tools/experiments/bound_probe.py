import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from saev_amd.engine import EngineConfig, SaeEngine
import sae_ref as R
def rand_params(d, s, seed=0):
    g = torch.Generator().manual_seed(seed)
    p = R.init_params(R.RefConfig(d_model=d, d_sae=s), g)
    p["b_enc"] = 0.05 * torch.randn(s, generator=g); p["b_dec"] = 0.1 * torch.randn(d, generator=g)
    p["W_enc"] = p["W_enc"] + 0.02 * torch.randn(d, s, generator=g)
    return p
for (d, s, k, n) in ((256, 8192, 32, 700), (1024, 32768, 32, 16384), (128, 4096, 16, 300)):
    p = rand_params(d, s, seed=73)
    x = (torch.randn(n, d, generator=torch.Generator().manual_seed(74)) + 0.5).cuda()
    eng = SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=k, k_aux=0, max_batch=n))
    eng.load_params(p)
    for i in range(4):
        eng.step_forward(x, training=False)
        st = eng.read_stats()
        print((d, s, k, n), i, eng.bound_state(), "dense", st.dense_route, "ovf", st.n_overflow_rows, "cmax", st.cand_max, flush=True)
    flags = None
