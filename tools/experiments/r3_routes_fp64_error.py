"""Relative Frobenius error of the weight gradients of both routes against an fp64 recomputation from the step's own codes."""
import math, os, pathlib, sys
import torch
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_dw_slices as T

for (d, s, k, b, n, kind) in [(1024, 8192, 32, 4096, 4096, "dense_latent"), (1024, 32768, 32, 16384, 16384, "plain")]:
    x = T._data(d, b, n, kind)
    for route in ("rows", "slices"):
        eng = T._engine(d, s, k, b, route)
        eng.step_forward(x); eng.step_dead(n); eng.step_backward(); torch.cuda.synchronize()
        idx, val = eng.last_codes(n)[:2]
        W_dec = eng.view("W_dec").double()
        f = torch.zeros(n, s, dtype=torch.float64, device="cuda"); f.scatter_(1, idx.long(), val.double())
        g = 2.0 * (f @ W_dec + eng.view("b_dec").double() - x.double()) / (n * d)
        mask = torch.zeros(n, s, dtype=torch.float64, device="cuda"); mask.scatter_(1, idx.long(), 1.0)
        dval = (g @ W_dec.t()) * mask
        ref = {"W_dec": f.t() @ g, "W_enc": x.double().t() @ dval, "b_enc": dval.sum(0)}
        gv = eng.grad_views()
        print(d, s, n, kind, route, {k_: f"{((gv[k_].double() - r).norm() / r.norm()).item():.2e}" for k_, r in ref.items()})
        del eng, f, mask, dval, g
        torch.cuda.empty_cache()
