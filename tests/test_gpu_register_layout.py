"""The kernels that keep a row's decoder rows in registers (sparse.hip: decode_q_kernel, decode_matry_q_kernel; auxk.hip:
aux_small_fused_kernel) at every width they are instantiated for -- d_model 256 / 512 / 768 / 1024 = 1 / 2 / 3 / 4 waves per
activation row -- against the CPU oracle (modeling.py:351-409 decode, objectives.py:125-138 prefixes, modeling.py:75-103 AuxK,
train.py:347-362 backward and clip) and against the form they replace (SAEV_AMD_DW=slices_a: dval from the backward's first pass;
SAEV_AMD_AUX_SMALL_MAX=64: the five-pass few-dead-latents kernels)."""
import math
import os

import pytest
import torch

import sae_ref as R
from test_gpu_parity import make_engine, rand_params

pytestmark = pytest.mark.gpu


def _env(name, value):
    class _Ctx:
        def __enter__(self):
            self.old = os.environ.get(name)
            if value is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = value

        def __exit__(self, *a):
            if self.old is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = self.old

    return _Ctx()


def _teacher_forced_step(eng, cfg, x, lr, prefixes=None, seed=None):
    state = R.TrainState(
        params={k_: v.cpu().clone() for k_, v in eng.param_views().items()},
        m={k_: eng.view(k_, eng.adam_m).cpu().clone() for k_ in R.PARAM_ORDER},
        v={k_: eng.view(k_, eng.adam_v).cpu().clone() for k_ in R.PARAM_ORDER},
        toks_since_active=eng.toks_since_active.cpu().clone(), adam_steps=eng.adam_steps, lr=lr)
    if seed is not None:
        torch.manual_seed(seed)  # the oracle draws the same prefixes from torch's global generator
    ref = R.train_step(state, x, cfg)
    eng.set_prefixes(prefixes)
    eng.train_step(x.cuda(), lr, cfg.grad_clip)
    return state, ref, eng.read_stats()


@pytest.mark.parametrize("d,k", [(1280, 64), (1280, 32), (512, 64), (1024, 48), (256, 33), (768, 64)])
def test_steps_match_the_oracle_at_the_wide_shapes(d, k):
    """decode_q_kernel<NW, 2>: more than 32 codes per row are decoded in two halves of 32 decoder rows (the first half gathered once
    more for dval), and d_model 1280 (configs[3]: five waves per activation row) -- the same three teacher-forced steps.  The clip norm
    at SURVEY 8c's 1e-3: the ORACLE's fp32 norm of these larger gradients is itself 1-2e-4 off its own fp64 value (every route of the
    HIP path agrees with the fp64 norm of its gradient to 1e-8: tools/experiments/r5_diag_wide.py)."""
    test_steps_match_the_oracle_at_every_width(d, k, 1, norm_tol=1e-3)


@pytest.mark.parametrize("d", [256, 512, 768, 1024])
@pytest.mark.parametrize("k,n_pre", [(8, 1), (32, 1), (32, 4), (17, 10)])
def test_steps_match_the_oracle_at_every_width(d, k, n_pre, norm_tol=1e-4):
    """Three teacher-forced train steps (lr 0 first, as the reference's scheduler gives) with 1 / 4 / 10 Matryoshka prefixes: losses,
    gradient norm and every parameter against the oracle; n = 300 rows is not a multiple of anything the kernels tile by."""
    s, n = 4 * d, 300
    p = rand_params(d, s, seed=300 + d + k)
    gen = torch.Generator().manual_seed(301 + d + k)
    cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k, k_aux=0, n_prefixes=n_pre, grad_clip=1.0)
    eng = make_engine(d, s, k, k_aux=0, max_batch=n)
    eng.load_params(p)
    flips = 0
    for i, lr in enumerate((0.0, 1e-3, 1e-3)):
        x = torch.randn(n, d, generator=gen) + 0.3
        prefixes = None
        if n_pre > 1:
            torch.manual_seed(2000 + i)
            prefixes = R.sample_prefixes(s, n_pre)
        state, ref, st = _teacher_forced_step(eng, cfg, x, lr, prefixes, seed=2000 + i if n_pre > 1 else None)
        flipped = not math.isclose(st.mse, ref["mse"], rel_tol=2e-6)
        flips += flipped
        assert math.isclose(st.mse, ref["mse"], rel_tol=4.0 / (n * k)), (i, st.mse, ref["mse"])
        if not flipped:
            assert math.isclose(st.grad_norm, ref["grad_norm"], rel_tol=norm_tol), (i, st.grad_norm, ref["grad_norm"])
            for key in R.PARAM_ORDER:
                bad = ~torch.isclose(eng.view(key).cpu(), state.params[key], rtol=1e-4, atol=2e-6)
                assert bad.float().mean() <= 1e-4, f"step {i} {key}: {bad.sum().item()} of {bad.numel()} elements off"
    assert flips <= 1


@pytest.mark.parametrize("d,k", [(1280, 64), (512, 64), (1280, 17)])
def test_gradients_agree_with_dval_formed_in_the_backward_wide(d, k, encoder_mode):
    test_gradients_agree_with_dval_formed_in_the_backward(d, 1, encoder_mode, k=k)


@pytest.mark.encoder_modes("f16r")  # the decode does not depend on the encoder arithmetic: run once
@pytest.mark.parametrize("d", [256, 512, 768, 1024])
@pytest.mark.parametrize("n_pre", [1, 5])
def test_gradients_agree_with_dval_formed_in_the_backward(d, n_pre, encoder_mode, k=32):
    """The same forward + backward with dval = <dL/dx_hat row (or the prefix block's suffix sum), decoder row> taken from the decode
    and formed by the first pass of the column slices: the four gradients agree to rounding, codes and loss bit for bit."""
    s, n = 8 * d, 1000
    p = rand_params(d, s, seed=400 + d)
    x = (torch.randn(n, d, generator=torch.Generator().manual_seed(401 + d)) + 0.2).cuda()
    torch.manual_seed(77)
    prefixes = R.sample_prefixes(s, n_pre) if n_pre > 1 else None
    out = {}
    for route in ("slices", "slices_a"):
        with _env("SAEV_AMD_DW", route):
            eng = make_engine(d, s, k, k_aux=0, max_batch=n)
        eng.load_params(p)
        eng.set_prefixes(prefixes)
        eng.step_forward(x)
        eng.step_dead(n)
        eng.step_backward()
        torch.cuda.synchronize()
        out[route] = ({name: v.clone() for name, v in eng.grad_views().items()}, eng.read_stats().mse, eng.last_codes(n))
    assert out["slices"][1] == out["slices_a"][1]
    assert torch.equal(out["slices"][2][0], out["slices_a"][2][0])
    for name in ("W_dec", "W_enc", "b_enc", "b_dec"):
        a, c = out["slices_a"][0][name], out["slices"][0][name]
        scale = a.abs().max().item() + 1e-30
        assert (a - c).abs().max().item() <= 2e-6 * scale + 1e-12, (name, (a - c).abs().max().item(), scale)
    assert out["slices"][0]["W_enc"].abs().sum() > 0


@pytest.mark.encoder_modes("f32", "f16r")  # run in the exact-fp32 and the default mode
@pytest.mark.parametrize("d", [256, 512, 768, 1024])
@pytest.mark.parametrize("n_dead", [1, 5, 8])
def test_one_pass_auxk_matches_the_oracle_and_the_five_pass_kernels(d, n_dead, encoder_mode):
    """At most eight dead latents: aux_small_fused_kernel.  Teacher-forced steps against the oracle (aux loss, dead count, gradient
    norm, parameters incl. the dead latents' rows), and the same steps on the five-pass kernels (aux_small_max = 64 keeps them)."""
    s, k, n, k_aux, thr = 4 * d, 8, 210, 64, 100_000
    p = rand_params(d, s, seed=500 + d + n_dead)
    dead = torch.randperm(s, generator=torch.Generator().manual_seed(502 + d))[:n_dead]
    # never selected (the k-th largest pre-activation is ~ +2.5), so they stay dead.  Not -100 as in tests/test_gpu_parity.py: the dead
    # latents' gradient rows would then hold all but 3e-4 of the squared gradient norm, and the ORACLE's fp32 norm of the 4 M-element
    # W_dec gradient drops the small rows behind the five big ones (2.73232 against 2.73268 in fp64 from its own gradient; the HIP
    # path reports 2.73268 -- tools/experiments/r4_diag_fused_aux3.py)
    p["b_enc"][dead] = -5.0
    toks = torch.zeros(s, dtype=torch.int64)
    toks[dead] = thr
    cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k, k_aux=k_aux, dead_threshold_tokens=thr)
    engs = {}
    for name, asm in (("one_pass", "0"), ("five_pass", "64")):
        with _env("SAEV_AMD_AUX_SMALL_MAX", asm):
            engs[name] = make_engine(d, s, k, k_aux=k_aux, thr=thr, max_batch=n)
        engs[name].load_params(p)
        engs[name].set_tracker(toks)
    gen = torch.Generator().manual_seed(503 + d + n_dead)
    flips = 0
    for i in range(7):
        x = torch.randn(n, d, generator=gen)
        state, ref, st = _teacher_forced_step(engs["one_pass"], cfg, x, 1e-3)
        # (the five-pass engine runs the same steps from the same initial state on its own)
        e5 = engs["five_pass"]
        assert st.n_dead == ref["n_dead"] == n_dead, (i, st.n_dead, ref["n_dead"])
        # (a row whose k-th and (k+1)-th pre-activation are a rounding apart may take the other one: one row of 210 -- the main
        # residual of that row, and with it the auxiliary target, then differs; such a step is only held to the coarse band)
        flipped = not math.isclose(st.mse, ref["mse"], rel_tol=2e-6)
        flips += flipped
        assert math.isclose(st.mse, ref["mse"], rel_tol=4.0 / (n * k)), (i, st.mse, ref["mse"])
        assert math.isclose(st.aux, ref["aux"], rel_tol=1e-2 if flipped else 1e-4, abs_tol=1e-12), (i, st.aux, ref["aux"])
        if not flipped:
            assert math.isclose(st.grad_norm, ref["grad_norm"], rel_tol=1e-4), (i, st.grad_norm, ref["grad_norm"])
            for key in R.PARAM_ORDER:
                bad = ~torch.isclose(engs["one_pass"].view(key).cpu(), state.params[key], rtol=1e-4, atol=2e-6)
                assert bad.float().mean() <= 1e-4, f"step {i} {key}: {bad.sum().item()} of {bad.numel()} elements off"
        e5.train_step(x.cuda(), 1e-3, 1.0)
        s5 = e5.read_stats()
        assert s5.n_dead == st.n_dead and math.isclose(s5.aux, st.aux, rel_tol=1e-5, abs_tol=1e-12), (i, s5.aux, st.aux)
        assert math.isclose(s5.grad_norm, st.grad_norm, rel_tol=1e-5)
    # the two engines ran the same seven steps from the same state on different kernels
    assert flips <= 1, flips
    diff = (engs["one_pass"].params - engs["five_pass"].params).abs().max().item()
    assert diff < 5e-5, diff
    assert engs["one_pass"].aux_route() in (1, 2) and engs["five_pass"].aux_route() in (1, 2)


@pytest.mark.encoder_modes("f16r")  # the decode does not depend on the encoder arithmetic: run once
@pytest.mark.parametrize("d,n", [(1024, 1000), (768, 300), (256, 515)])
def test_slice_decode_agrees_with_the_row_decode(d, n, encoder_mode):
    """SAEV_AMD_DW=slices_s: the decode out of 32-column slices of W_dec (sparse.hip: decode_s_kernel; opt-in) against the default
    register decode: the same codes, x_hat and loss bit for bit (both sum x_hat in code order), gradients to rounding (the dval
    shares are added in another order)."""
    s, k = 8 * d, 32
    p = rand_params(d, s, seed=600 + d)
    x = (torch.randn(n, d, generator=torch.Generator().manual_seed(601 + d)) + 0.2).cuda()
    out = {}
    for route in ("slices", "slices_s"):
        with _env("SAEV_AMD_DW", route):
            eng = make_engine(d, s, k, k_aux=0, max_batch=n)
        eng.load_params(p)
        eng.step_forward(x)
        eng.step_dead(n)
        eng.step_backward()
        torch.cuda.synchronize()
        idx, val, x_hat = eng.last_codes(n)
        st = eng.read_stats()
        out[route] = ({name: v.clone() for name, v in eng.grad_views().items()}, st.mse, idx.clone(), x_hat.clone(), st.l0, eng.fired.clone())
    assert torch.equal(out["slices"][2], out["slices_s"][2]) and torch.equal(out["slices"][3], out["slices_s"][3])
    assert math.isclose(out["slices"][1], out["slices_s"][1], rel_tol=1e-6) and out["slices"][4] == out["slices_s"][4]
    assert torch.equal(out["slices"][5], out["slices_s"][5])
    for name in ("W_dec", "W_enc", "b_enc", "b_dec"):
        a, c = out["slices"][0][name], out["slices_s"][0][name]
        scale = a.abs().max().item() + 1e-30
        assert (a - c).abs().max().item() <= 2e-6 * scale + 1e-12, (name, (a - c).abs().max().item(), scale)
