"""The bf16 encoder mode (BASELINE.json configs[3]) against the oracle's bf16 model and the fp32 oracle.

Tolerances: against the oracle evaluated with the same bf16-rounded encoder operands the
HIP path is held to the fp32 bands of test_gpu_parity.py; against the fp32 oracle the reconstruction MSE must
agree to 1e-2 relative."""

import math

import pytest
import torch

import sae_ref as R
from conftest import load_golden
from test_gpu_parity import codes_to_dense, make_engine, rand_params

pytestmark = [pytest.mark.gpu, pytest.mark.encoder_modes("f32")]  # bf16 tests pick their own encoder mode: collected once


@pytest.mark.parametrize("n,d,s", [(96, 48, 320), (300, 128, 1024), (5, 16, 24), (257, 256, 768), (64, 100, 260)])
def test_bf16_encode_dense_matches_bf16_oracle(n, d, s):
    p = rand_params(d, s, seed=n)
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(n + 1))
    eng = make_engine(d, s, 8, max_batch=max(n, 8), encoder="bf16")
    eng.load_params(p)
    h = eng.encode_dense(x.cuda()).cpu()
    ref = R.encode_pre_bf16(x, p["W_enc"], p["b_enc"])
    torch.testing.assert_close(h, ref, rtol=1e-5, atol=1e-5)
    # and it really is bf16 arithmetic: visibly different from the fp32 product, but within bf16 rounding of it
    full = R.encode_pre(x, p["W_enc"], p["b_enc"])
    err = (h - full).abs().max().item()
    assert 1e-5 < err < 2.0 ** -7 * (x.abs().max() * p["W_enc"].abs().max() * d).item()


@pytest.mark.parametrize("n,d,s,k", [(128, 64, 512, 8), (300, 128, 1024, 16), (200, 64, 2048, 32), (130, 32, 4096, 64)])
def test_bf16_fused_encode_topk_matches_bf16_oracle(n, d, s, k):
    p = rand_params(d, s, seed=n + k)
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(n))
    eng = make_engine(d, s, k, max_batch=n, encoder="bf16")
    eng.load_params(p)
    idx, val = eng.encode_topk(x.cuda())
    h = R.encode_pre_bf16(x, p["W_enc"], p["b_enc"])
    want = torch.topk(h, min(k, s), dim=-1).values.sort(dim=-1).values
    torch.testing.assert_close(val.cpu().sort(dim=-1).values, want, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(h.gather(1, idx.cpu().long()), val.cpu(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_bf16_teacher_forced_steps(tag):
    """Each step from identical state: HIP bf16 step vs the oracle's bf16 model (tight) and vs the fp32 oracle
    (MSE within 1e-2 relative)."""
    g = load_golden(f"g9_train_{tag}")
    d, s, k, bsz = int(g["d"]), int(g["s"]), int(g["k"]), int(g["bsz"])
    kw = dict(d_model=d, d_sae=s, top_k=k, k_aux=int(g["k_aux"]), dead_threshold_tokens=int(g["thr"]))
    cfg_bf, cfg_32 = R.RefConfig(encoder_bf16=True, **kw), R.RefConfig(**kw)
    eng = make_engine(d, s, k, k_aux=int(g["k_aux"]), thr=int(g["thr"]), max_batch=bsz, encoder="bf16")
    eng.load_params({key: g["init_" + key] for key in R.PARAM_ORDER})
    sched = R.WarmupCosine(0.0, int(g["n_warm"]), float(g["lr"]), math.ceil(int(g["n_train"]) / bsz), 0.0)
    batches = list(g["acts"].split(bsz))
    lr = 0.0
    n_flip_steps = 0
    for i, x in enumerate(R.limited_batches(batches, int(g["n_train"]), bsz, drop_last=False)):
        def snapshot():
            return R.TrainState(
                params={k_: v.cpu().clone() for k_, v in eng.param_views().items()},
                m={k_: eng.view(k_, eng.adam_m).cpu().clone() for k_ in R.PARAM_ORDER},
                v={k_: eng.view(k_, eng.adam_v).cpu().clone() for k_ in R.PARAM_ORDER},
                toks_since_active=eng.toks_since_active.cpu().clone(), adam_steps=eng.adam_steps, lr=lr)

        state, state32 = snapshot(), snapshot()
        ref = R.train_step(state, x, cfg_bf)
        ref32 = R.train_step(state32, x, cfg_32)
        eng.train_step(x.cuda(), lr, 1.0)
        st = eng.read_stats()
        assert math.isclose(st.mse, ref32["mse"], rel_tol=1e-2), (i, st.mse, ref32["mse"])
        flipped = not math.isclose(st.mse, ref["mse"], rel_tol=2e-6)
        n_flip_steps += flipped
        assert math.isclose(st.mse, ref["mse"], rel_tol=4.0 / (bsz * k)), (i, st.mse, ref["mse"])
        assert math.isclose(st.l0, ref["l0"], rel_tol=1e-6) and abs(st.n_dead - ref["n_dead"]) <= flipped
        if not flipped:
            assert torch.equal(eng.toks_since_active.cpu(), state.toks_since_active)
            assert math.isclose(st.aux, ref["aux"], rel_tol=1e-4, abs_tol=1e-9), (i, st.aux, ref["aux"])
            assert math.isclose(st.grad_norm, ref["grad_norm"], rel_tol=1e-4), (i, st.grad_norm, ref["grad_norm"])
            for key in R.PARAM_ORDER:
                torch.testing.assert_close(eng.view(key).cpu(), state.params[key], rtol=1e-4, atol=2e-6,
                                           msg=lambda m: f"step {i} {key}: {m}")
        lr = sched.step()
    assert n_flip_steps <= 3


def test_bf16_reconstruction_close_to_fp32_at_width():
    """Wider shape (D=1280 like configs[3], S=8192, k=64): the bf16 encoder selects nearly the same codes and its
    reconstruction error is within 1e-2 relative of the fp32 path's."""
    d, s, k, n = 1280, 8192, 64, 512
    p = rand_params(d, s, seed=3)
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(4)).cuda()
    out = {}
    for mode in ("f16x3", "bf16"):
        eng = make_engine(d, s, k, k_aux=0, max_batch=n, encoder=mode)
        eng.load_params(p)
        eng.step_forward(x, training=False)
        idx, val, _ = eng.last_codes(n)
        out[mode] = (eng.read_stats().mse, codes_to_dense(idx.cpu(), val.cpu(), s))
    assert math.isclose(out["bf16"][0], out["f16x3"][0], rel_tol=1e-2)
    same = ((out["bf16"][1] != 0) & (out["f16x3"][1] != 0)).sum().item() / (n * k)
    assert same > 0.9, same
