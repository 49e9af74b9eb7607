"""Size-independent properties at BASELINE.json's full sizes (configs[1]: d=1024, S=32768, k=32,
B=16384; and configs[0]: d=768, S=6144, B=4096), where the CPU oracle is too slow to run whole."""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def build(d, s, k, b, seed=0, **kw):
    from saev_amd.engine import EngineConfig, SaeEngine

    eng = SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=k, max_batch=b, **kw))
    g = torch.Generator(device="cuda").manual_seed(seed)
    W = (torch.rand(s, d, device="cuda", generator=g) * 2 - 1) * math.sqrt(6.0 / d)
    W /= W.norm(dim=1, keepdim=True)
    eng.view("W_dec").copy_(W)
    eng.view("W_enc").copy_(W.t() + 0.01 * torch.randn(d, s, device="cuda", generator=g))
    eng.view("b_enc").copy_(0.05 * torch.randn(s, device="cuda", generator=g))
    eng.view("b_dec").copy_(0.05 * torch.randn(d, device="cuda", generator=g))
    x = torch.randn(b, d, device="cuda", generator=g) + torch.randn(d, device="cuda", generator=g)
    return eng, x


@pytest.mark.parametrize("d,s,k,b", [(1024, 32768, 32, 16384), (768, 6144, 32, 4096), (1280, 81920, 64, 2048)])
def test_fused_topk_is_exact_at_full_size(d, s, k, b):
    """EVERY row against an fp64 product (slabs of 512 rows): the emitted values are the pre-activations at the emitted
    latents, and nothing left out exceeds the smallest kept value -- both to the rounding of a d-term fp32 dot product,
    tol_b = 8 * 2^-24 * ||x_b|| * max_s ||W_enc[:, s]|| (2e-5 on this data; the f16r first pass works to ~3e-2 and relies on
    its margin + exact refinement to get here)."""
    eng, x = build(d, s, k, b)
    idx, val = eng.encode_topk(x)
    st_idx = idx.long()
    assert idx.shape == (b, k) and (idx[:, 1:] > idx[:, :-1]).all() and idx.min() >= 0 and idx.max() < s
    W, be = eng.view("W_enc").double(), eng.view("b_enc").double()
    wmax = W.norm(dim=0).max().item()
    worst_val = worst_cut = 0.0
    for lo in range(0, b, 512):
        rows = slice(lo, min(b, lo + 512))
        h = x[rows].double() @ W + be
        tol = 8.0 * 2.0 ** -24 * x[rows].double().norm(dim=1) * wmax
        err = (h.gather(1, st_idx[rows]) - val[rows].double()).abs().amax(dim=1)
        worst_val = max(worst_val, (err / tol).max().item())
        assert (err <= tol).all(), f"rows {lo}..: value error {err.max().item():.3e} > tol {tol.min().item():.3e}"
        kth = val[rows].min(dim=1).values.double()
        over = h.scatter(1, st_idx[rows], float("-inf")).amax(dim=1) - kth
        worst_cut = max(worst_cut, (over / tol).max().item())
        assert (over <= tol).all(), f"rows {lo}..: a left-out pre-activation exceeds the smallest kept one by {over.max().item():.3e}"
        del h
    print(f"full-size TopK ({d}, {s}, {k}, {b}): worst value error {worst_val:.2f} tol, worst cut excess {worst_cut:.2f} tol")
    # the dense route agrees exactly on the selected values
    rows = torch.randperm(b, device="cuda")[:64]
    hd = eng.encode_dense(x[rows])
    i2, v2 = eng.topk_dense(hd, k)
    torch.testing.assert_close(v2.sort(dim=1).values, val[rows].sort(dim=1).values, rtol=1e-5, atol=1e-5)


def test_decode_is_affine_in_the_codes():
    eng, x = build(1024, 32768, 32, 4096)
    idx, val = eng.encode_topk(x)
    b_dec = eng.view("b_dec")
    y1 = eng.decode_sparse(idx, val)[:, 0] - b_dec
    y2 = eng.decode_sparse(idx, 2.5 * val)[:, 0] - b_dec
    torch.testing.assert_close(y2, 2.5 * y1, rtol=1e-5, atol=1e-5)
    y0 = eng.decode_sparse(idx, torch.zeros_like(val))[:, 0]
    torch.testing.assert_close(y0, b_dec.expand_as(y0), rtol=0, atol=0)


def test_full_size_step_invariants():
    d, s, k, b = 1024, 32768, 32, 16384
    eng, x = build(d, s, k, b)
    eng2, _ = build(d, s, k, b)
    losses = []
    for i in range(6):
        lr = 0.0 if i == 0 else 4e-4
        eng.train_step(x, lr, 1.0)
        eng2.train_step(x, lr, 1.0)
        st = eng.read_stats()
        losses.append(st.mse)
        assert st.l0 == k and st.n_dead == 0 and st.aux == 0.0 and st.n_overflow_rows == 0
        assert 0 < st.cand_max <= 4096
        # sse identities: fp64 SSE / (B*D) equals the scaled fp32 MSE
        assert math.isclose(st.sse / (b * d), st.mse, rel_tol=1e-4)
    assert losses[-1] < losses[1], losses
    # decoder rows were unit-norm when the step used them; one Adam step later they are close to it
    norms = eng.view("W_dec").norm(dim=1)
    assert (norms - 1).abs().max() < 0.05
    # replicas fed the same data stay bit-identical (what data-parallel training relies on)
    assert torch.equal(eng.params, eng2.params)
    # gradient identities on the last step: db_dec = column sums of dL/dx_hat; rows of latents that
    # never fired have zero gradient in W_dec and b_enc
    eng.step_forward(x, training=True); eng.step_dead(b); eng.step_backward()
    idx, val, x_hat = eng.last_codes(b)
    g = eng.grad_views()
    dx = 2.0 / (b * d) * (x_hat - x)
    torch.testing.assert_close(g["b_dec"], dx.sum(dim=0), rtol=1e-3, atol=1e-7)
    fired = torch.zeros(s, dtype=torch.bool, device="cuda")
    fired[idx.reshape(-1).long()] = True
    assert (g["W_dec"][~fired] == 0).all() and (g["b_enc"][~fired] == 0).all() and (g["W_enc"][:, ~fired] == 0).all()
    # dW_dec for a sampled latent equals the explicit sum
    lat = idx[0, 0].item()
    rows, cols = (idx == lat).nonzero(as_tuple=True)
    want = (val[rows, cols, None].double() * dx[rows].double()).sum(dim=0).float()
    torch.testing.assert_close(g["W_dec"][lat], want, rtol=1e-3, atol=1e-8)
