"""One FUSED train step (lr > 0) under a loose bound of the dead count: the parameters the matrix-core route (bound 100, 20-64 dead) leaves
against those of the dense route (aux_wide_route=1) from the same state."""
import sys, torch
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
    for i, x in enumerate(xs[:60]):
        eng.train_step(x, 0.0, 1.0)
    before = eng.params.clone()
    eng.train_step(xs[60], 1e-2, 1.0)
    r60 = eng.aux_route()
    eng.train_step(xs[61], 1e-2, 1.0)
    res.append((before, {k_: v.clone() for k_, v in eng.param_views().items()}, r60, eng.aux_route(), eng.read_stats()))
    eng.close()
(b0, pa, ra0, ra1, sa), (b1, pb, rb0, rb1, sb) = res
print("same start", torch.equal(b0, b1), "routes", (ra0, ra1), (rb0, rb1), "n_dead", sa.n_dead, sb.n_dead, "aux", sa.aux, sb.aux)
dd = dead.cuda()
for key in R.PARAM_ORDER:
    print(key, "max abs diff", (pa[key] - pb[key]).abs().max().item(), "max", pb[key].abs().max().item())
print("dead rows W_dec: diff", (pa["W_dec"][dd] - pb["W_dec"][dd]).abs().max().item(), " moved (dense)", (pb["W_dec"][dd] - rand_params(d, s, seed=850)["W_dec"].cuda()[dd]).abs().max().item())
print("dead cols W_enc: diff", (pa["W_enc"][:, dd] - pb["W_enc"][:, dd]).abs().max().item())
print("dead b_enc: diff", (pa["b_enc"][dd] - pb["b_enc"][dd]).abs().max().item(), "values", pa["b_enc"][dd][:4].tolist(), pb["b_enc"][dd][:4].tolist())
