"""Debug: the (0, 600) case of test_dense_auxk_sized_by_a_bound_needs_no_readback, phase by phase, gradients per tensor."""
import sys, pathlib, math, torch
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
import sae_ref as R
from saev_amd.engine import EngineConfig, SaeEngine

def rand_params(d, s, seed=0):
    g = torch.Generator().manual_seed(seed)
    p = R.init_params(R.RefConfig(d_model=d, d_sae=s), g)
    p["b_enc"] = 0.05 * torch.randn(s, generator=g)
    p["b_dec"] = 0.1 * torch.randn(d, generator=g)
    p["W_enc"] = p["W_enc"] + 0.02 * torch.randn(d, s, generator=g)
    return p

n_dead, n_near = 0, 600
d, s, k, n, k_aux, thr = 128, 1024, 8, 200, 64, 100_000
p = rand_params(d, s, seed=180 + n_dead)
gen = torch.Generator().manual_seed(181 + n_dead)
cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k, k_aux=k_aux, dead_threshold_tokens=thr)
toks = torch.zeros(s, dtype=torch.int64)
order = torch.randperm(s, generator=torch.Generator().manual_seed(182))
near = order[:n_near]
toks[near] = thr - 2 * n
eng = SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=k, k_aux=k_aux, alpha=1/32, dead_threshold_tokens=thr, max_batch=n))
eng.load_params(p); eng.set_tracker(toks)
for i in range(9):
    x = torch.randn(n, d, generator=gen)
    state = R.TrainState(
        params={k_: v.cpu().clone() for k_, v in eng.param_views().items()},
        m={k_: eng.view(k_, eng.adam_m).cpu().clone() for k_ in R.PARAM_ORDER},
        v={k_: eng.view(k_, eng.adam_v).cpu().clone() for k_ in R.PARAM_ORDER},
        toks_since_active=eng.toks_since_active.cpu().clone(), adam_steps=eng.adam_steps, lr=1e-3)
    ref = R.train_step(state, x, cfg)
    xc = x.cuda()
    if i == 5 and len(sys.argv) > 1 and sys.argv[1] == "dirty":
        eng.set_tracker(eng.toks_since_active.clone())  # records invalid -> read-back -> route 2 (same kernels, count known to the host)
    eng.step_forward(xc, training=True); eng.step_dead(n)
    if i in (4, 5):
        torch.cuda.synchronize()
        dl = (eng.toks_since_active >= thr).nonzero().flatten()
        idx, val, x_hat = eng.last_codes(n)
        Wd, We, be, bd = eng.view("W_dec").double(), eng.view("W_enc").double(), eng.view("b_enc").double(), eng.view("b_dec").double()
        H = xc.double() @ We[:, dl] + be[dl]
        E = H @ Wd[dl] + bd
        diff = E - (xc.double() - x_hat.double())
        print(f"    by hand on the device state: dead {dl.tolist()} aux {(diff * diff).mean().item() / 32:.6e}; dead latent among the codes: {(idx == dl[0]).any().item()}; "
              f"oracle W_dec row norm {state.params['W_dec'][dl[0]].norm().item():.7f} ours {eng.view('W_dec')[dl[0]].norm().item():.7f}; x_hat vs oracle-normalised...")
    eng.step_backward()
    torch.cuda.synchronize()
    raw = {k_: v.cpu().clone() for k_, v in eng.grad_views().items()}
    route = eng.aux_route()
    eng.step_tail(1e-3, 1.0)
    st = eng.read_stats()
    # oracle's raw (unprojected) grads: recompute
    P = {k_: state.params[k_] for k_ in R.PARAM_ORDER}
    print(f"step {i}: route {route} n_dead {st.n_dead}/{ref['n_dead']} aux {st.aux:.6e}/{ref['aux']:.6e} gn {st.grad_norm:.8f}/{ref['grad_norm']:.8f} rel {abs(st.grad_norm-ref['grad_norm'])/ref['grad_norm']:.2e}")
    dead_now = (eng.toks_since_active.cpu() >= thr).nonzero().flatten().tolist()
    for key in R.PARAM_ORDER:
        g_ref = ref["grads"][key]
        g_our = raw[key]
        if key == "W_dec":
            g_our = R.remove_parallel_grads(g_our, R.normalize_w_dec(p["W_dec"]) if False else eng.view("W_dec").cpu()) if False else g_our
        diff = (g_our - g_ref).abs()
        if key != "W_dec":
            print(f"    {key}: max abs diff {diff.max().item():.3e} (ref max {g_ref.abs().max().item():.3e}), norm ours {g_our.norm().item():.6e} ref {g_ref.norm().item():.6e}")
        else:
            print(f"    {key} (ours raw, ref projected): norm ours {g_our.norm().item():.6e} ref {g_ref.norm().item():.6e}")
    if dead_now:
        print("    dead latents:", dead_now[:8], "W_enc col grad ours", raw["W_enc"][:, dead_now[0]].norm().item(), "ref", ref["grads"]["W_enc"][:, dead_now[0]].norm().item(),
              "b_enc ours", raw["b_enc"][dead_now[0]].item(), "ref", ref["grads"]["b_enc"][dead_now[0]].item())
