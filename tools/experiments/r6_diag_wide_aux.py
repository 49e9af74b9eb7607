import sys, torch, math
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import sae_ref as R
from test_gpu_parity import rand_params, make_engine
d, s, k, k_aux, n = 256, 2048, 8, 128, 256
thr = 30 * n + n // 2
ND = int(sys.argv[1]) if len(sys.argv) > 1 else 20
p = rand_params(d, s, seed=850)
g = torch.Generator().manual_seed(851)
perm = torch.randperm(s, generator=g)
dead, sleepy = perm[:ND], perm[ND:110]
p["b_enc"][dead] = -100.0; p["b_enc"][sleepy] = -100.0; p["W_enc"][0, sleepy] = 300.0
xs = []
for i in range(64):
    x = torch.randn(n, d, generator=g); x[:, 0] = 1.0 if i % 30 == 29 else 0.0
    xs.append(x.cuda())
res = []
for wide_off in (0, 1):
    eng = make_engine(d, s, k, k_aux=k_aux, thr=thr, max_batch=n, aux_wide_route=wide_off)
    eng.load_params(p)
    out = {}
    for i, x in enumerate(xs):
        eng.step_forward(x, training=True); eng.step_dead(n); eng.step_backward()
        if i in (45, 60, 61):
            out[i] = ({k_: v.clone() for k_, v in eng.grad_views().items()}, eng.aux_route(), eng.read_stats().n_dead)
        eng.step_tail(0.0, 1.0)
    res.append(out); eng.close()
for i in (45, 60, 61):
    (ga, ra, na), (gb, rb, nb) = res[0][i], res[1][i]
    print("step", i, "routes", ra, rb, "n_dead", na, nb)
    for key in R.PARAM_ORDER:
        diff = (ga[key] - gb[key]).abs().max().item(); ref = gb[key].abs().max().item()
        print("   ", key, "max abs diff", diff, "max ref", ref)
    dd = dead.cuda()
    print("    dead rows W_dec diff", (ga["W_dec"][dd] - gb["W_dec"][dd]).abs().max().item(), "ref", gb["W_dec"][dd].abs().max().item())
    print("    dead cols W_enc diff", (ga["W_enc"][:, dd] - gb["W_enc"][:, dd]).abs().max().item(), "ref", gb["W_enc"][:, dd].abs().max().item())
