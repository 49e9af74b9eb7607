"""One case of r4_fuzz_register_layout.py in detail: which parameters differ from the oracle's, on which rows, and how close the
row's k-th and (k+1)-th pre-activations are (a near-tie flips legitimately; a lost survivor would not be one).
   python tools/experiments/r4_diag_fuzz_case.py <seed> <trial>"""
import math, random, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import sae_ref as R
from test_gpu_parity import make_engine, rand_params
seed, want = int(sys.argv[1]), int(sys.argv[2])
random.seed(seed)
match = tuple(int(v) for v in sys.argv[3:9]) if len(sys.argv) >= 9 else None  # (d s k n P nd): the first trial of that shape
for trial in range(want + 1 if match is None else 10_000):
    d = random.choice([256, 512, 768, 1024]); k = random.choice([1, 2, 5, 16, 31, 32]); n = random.choice([1, 3, 17, 64, 129, 333])
    s = random.choice([2, 4]) * d; P = random.choice([1, 1, 2, 7, 16]); nd = random.choice([0, 0, 1, 3, 8])
    if match is not None and (d, s, k, n, P, nd) == match:
        want = trial
        break
print(f"trial {want}: d={d} s={s} k={k} n={n} P={P} nd={nd}")
trial = want
thr = 1000
p = rand_params(d, s, seed=trial)
toks = torch.zeros(s, dtype=torch.int64)
if nd:
    dead = torch.randperm(s, generator=torch.Generator().manual_seed(trial))[:nd]
    p["b_enc"][dead] = -5.0; toks[dead] = thr
cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k, k_aux=64 if nd else 0, n_prefixes=P, dead_threshold_tokens=thr, grad_clip=1.0)
eng = make_engine(d, s, k, k_aux=64 if nd else 0, thr=thr, max_batch=n)
eng.load_params(p); eng.set_tracker(toks)
x = torch.randn(n, d, generator=torch.Generator().manual_seed(1000 + trial)) + 0.2
# exact pre-activations of the normalised parameters (fp64) and the gaps at the cut
Wd = p["W_dec"].double(); Wd = Wd / Wd.norm(dim=1, keepdim=True)
h = x.double() @ p["W_enc"].double() + p["b_enc"].double()
top = h.topk(min(k + 1, s), dim=1).values
gap = (top[:, k - 1] - top[:, k]) if k < s else None
print("smallest gaps between the k-th and (k+1)-th pre-activation:", gap.sort().values[:5].tolist())
idx, val = eng.encode_topk(x.cuda()) if hasattr(eng, "encode_topk") else (None, None)
if idx is not None:
    ref_idx = h.topk(k, dim=1).indices.sort(dim=1).values
    got = idx.cpu().long().sort(dim=1).values
    rows = (ref_idx != got).any(dim=1).nonzero().flatten().tolist()
    print("rows whose codes differ from the fp64 top-k:", rows, [float(gap[r]) for r in rows])
state = R.TrainState.create({kk: v.clone() for kk, v in p.items()}); state.toks_since_active = toks.clone(); state.lr = 1e-3
torch.manual_seed(5000 + trial)
prefixes = R.sample_prefixes(s, P) if P > 1 else None
torch.manual_seed(5000 + trial)
ref = R.train_step(state, x, cfg)
eng.set_prefixes(prefixes)
eng.train_step(x.cuda(), 1e-3, 1.0)
for key in R.PARAM_ORDER:
    a, b = eng.view(key).cpu(), state.params[key]
    badm = ~torch.isclose(a, b, rtol=1e-4, atol=2e-6)
    if badm.any():
        if a.dim() == 2:
            r = badm.any(dim=1).nonzero().flatten().tolist(); c = badm.any(dim=0).nonzero().flatten().tolist()
            print(key, "mismatch fraction", badm.float().mean().item(), "rows", r[:8], "n_cols", len(c), "max abs diff", (a - b).abs().max().item())
        else:
            print(key, "mismatch at", badm.nonzero().flatten().tolist()[:8], (a - b).abs().max().item())
print("dead latents:", dead.tolist() if nd else [])
