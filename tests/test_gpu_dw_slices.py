"""The column-sliced weight-gradient passes (launch_dw_slices, the default route of a one-pass backward) against the row kernels
(dw_rows / dw_combine, SAEV_AMD_DW=rows) and against an fp64 recomputation of the reference's autograd gradients
(src/saev/framework/train.py:347-348) from the step's own codes."""

import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(d, s, k, b, route, seed=0, **kw):
    from saev_amd.engine import EngineConfig, SaeEngine

    old = os.environ.get("SAEV_AMD_DW")
    os.environ["SAEV_AMD_DW"] = route
    try:
        eng = SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=k, max_batch=b, **kw))
    finally:
        if old is None:
            del os.environ["SAEV_AMD_DW"]
        else:
            os.environ["SAEV_AMD_DW"] = old
    g = torch.Generator(device="cuda").manual_seed(seed)
    W = (torch.rand(s, d, device="cuda", generator=g) * 2 - 1) * math.sqrt(6.0 / d)
    W /= W.norm(dim=1, keepdim=True)
    eng.view("W_dec").copy_(W)
    eng.view("W_enc").copy_(W.t() + 0.01 * torch.randn(d, s, device="cuda", generator=g))
    eng.view("b_enc").copy_(0.05 * torch.randn(s, device="cuda", generator=g))
    eng.view("b_dec").copy_(0.05 * torch.randn(d, device="cuda", generator=g))
    return eng


def _data(d, b, n, kind, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(n, d, device="cuda", generator=g)
    if kind == "dense_latent":
        # a strong common direction: one latent fires on (nearly) every row, its pairs span hundreds of runs
        x = x + 6.0 * torch.randn(d, device="cuda", generator=g)
    elif kind == "few_latents":
        # rows drawn from a handful of prototypes: a few latents carry most pairs, most latents are unused
        protos = torch.randn(5, d, device="cuda", generator=g) * 3
        x = 0.2 * x + protos[torch.randint(0, 5, (n,), device="cuda", generator=g)]
    return x


SHAPES = [
    # d, s, k, max_batch, n rows, data
    (1024, 32768, 32, 16384, 16384, "plain"),
    (1024, 32768, 32, 16384, 16384, "dense_latent"),
    (1024, 8192, 32, 4096, 4096, "few_latents"),
    (768, 6144, 32, 4096, 4096, "plain"),        # 24 slices: not a multiple of the eight XCDs
    (1280, 20480, 64, 2048, 2048, "plain"),      # 40 slices, k = 64
    (128, 1024, 8, 512, 300, "plain"),           # fewer rows than max_batch, pairs not a multiple of the run length
    (64, 256, 4, 64, 3, "plain"),                # 12 pairs: a single run
    (32, 64, 2, 8, 1, "plain"),                  # one row, one slice
]


@pytest.mark.parametrize("d,s,k,b,n,kind", SHAPES)
def test_slices_match_rows_and_fp64(d, s, k, b, n, kind):
    x = _data(d, b, n, kind)
    grads = {}
    stats = {}
    for route in ("rows", "slices"):
        eng = _engine(d, s, k, b, route)
        eng.step_forward(x)
        eng.step_dead(n)
        eng.step_backward()
        torch.cuda.synchronize()
        grads[route] = {name: v.clone() for name, v in eng.grad_views().items()}
        stats[route] = eng
    # (the codes are the same: the forward does not depend on the route)
    e_r, e_s = stats["rows"], stats["slices"]
    idx_r, val_r = e_r.last_codes(n)[:2]
    idx_s, val_s = e_s.last_codes(n)[:2]
    assert torch.equal(idx_r, idx_s) and torch.equal(val_r, val_s)
    for name in ("W_dec", "W_enc", "b_enc", "b_dec"):
        a, c = grads["rows"][name], grads["slices"][name]
        scale = a.abs().max().item() + 1e-30
        assert (a - c).abs().max().item() <= 2e-6 * scale + 1e-12, (name, (a - c).abs().max().item(), scale)
    # dW_dec sums val * g in the same (row-ascending) order per element on both routes, with fmas
    # fp64 recomputation of the autograd gradients from the codes: dW_dec = f^T g, dW_enc = x^T (dval), db_enc = colsum(dval)
    W_dec = e_s.view("W_dec").double()
    f = torch.zeros(n, s, dtype=torch.float64, device="cuda")
    f.scatter_(1, idx_s.long(), val_s.double())
    x_hat = f @ W_dec + e_s.view("b_dec").double()
    g = 2.0 * (x_hat - x.double()) / (n * d)
    mask = torch.zeros(n, s, dtype=torch.float64, device="cuda")
    mask.scatter_(1, idx_s.long(), 1.0)
    dval = (g @ W_dec.t()) * mask
    ref = {"W_dec": f.t() @ g, "W_enc": x.double().t() @ dval, "b_enc": dval.sum(0), "b_dec": g.sum(0)}
    for name, r in ref.items():
        c = grads["slices"][name].double()
        scale = r.abs().max().item() + 1e-30
        assert (c - r).abs().max().item() <= 2e-5 * scale, (name, (c - r).abs().max().item(), scale)


def test_train_steps_track_the_row_route():
    """A few whole steps (clip norm from the row statistics the finalize pass leaves, projection inside Adam): parameters of
    the two routes stay within rounding of each other, and replicas on the slices route stay bit-identical."""
    d, s, k, b = 1024, 16384, 32, 8192
    x = _data(d, b, b, "dense_latent")
    engs = [_engine(d, s, k, b, "rows"), _engine(d, s, k, b, "slices"), _engine(d, s, k, b, "slices")]
    for i in range(4):
        for e in engs:
            e.train_step(x, 4e-4, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(engs[1].params, engs[2].params)
    diff = (engs[0].params - engs[1].params).abs().max().item()
    assert diff < 2e-5, diff
    sr, ss = engs[0].read_stats(), engs[1].read_stats()
    assert math.isclose(sr.mse, ss.mse, rel_tol=1e-5) and math.isclose(sr.grad_norm, ss.grad_norm, rel_tol=1e-4)


@pytest.mark.parametrize("d,s,k,b,n,kind", [SHAPES[1], SHAPES[3], SHAPES[5]])
def test_two_pass_slices_are_bit_identical_to_one_pass(d, s, k, b, n, kind):
    """The decoder half / encoder half form of the backward (what the sharded data-parallel step runs, framework/ddp.py) goes
    through the same passes in the same order."""
    x = _data(d, b, n, kind)
    grads = []
    for two_pass in (False, True):
        eng = _engine(d, s, k, b, "slices")
        eng.step_forward(x)
        eng.step_dead(n)
        if two_pass:
            eng.grad_w_enc_t()
            eng.backward_begin()
            eng.backward_rows(0, s, 1)
            eng.backward_rows(0, s, 2)
            eng.backward_end()
        else:
            eng.step_backward()
        torch.cuda.synchronize()
        grads.append(eng.grads.clone())
    assert torch.equal(grads[0], grads[1]) and grads[0].abs().sum() > 0


@pytest.mark.parametrize("d,s,k,b,n,prefixes", [
    (1024, 32768, 32, 16384, 16384, [37, 300, 1200, 5000, 9000, 20000, 32768]),
    (768, 6144, 32, 4096, 4096, [1, 2, 64, 6144]),
    (128, 1024, 8, 512, 300, [100, 300, 1024]),
])
def test_matryoshka_slices_match_rows(d, s, k, b, n, prefixes):
    """Matryoshka prefixes (the reference's default objective, objectives.py:125-138): a pair of latent i gathers the suffix sum
    C_p(i) of its row; the slices read it from the [slice][p][row] copy the decode leaves."""
    x = _data(d, b, n, "dense_latent")
    grads = {}
    for route in ("rows", "slices"):
        eng = _engine(d, s, k, b, route)
        eng.set_prefixes(prefixes)
        eng.step_forward(x)
        eng.step_dead(n)
        eng.step_backward()
        torch.cuda.synchronize()
        grads[route] = {name: v.clone() for name, v in eng.grad_views().items()}
    for name in ("W_dec", "W_enc", "b_enc", "b_dec"):
        a, c = grads["rows"][name], grads["slices"][name]
        scale = a.abs().max().item() + 1e-30
        assert (a - c).abs().max().item() <= 2e-6 * scale + 1e-12, (name, (a - c).abs().max().item(), scale)
    assert grads["slices"]["W_enc"].abs().sum() > 0


@pytest.mark.parametrize("d,s,k,n", [(1024, 16384, 32, 4096), (768, 6144, 32, 1000), (128, 1024, 8, 150)])
def test_gathered_backward_on_slices(d, s, k, n):
    """The backward of a sparse-state exchange (framework/ddp.py: x, dL/dx_hat and the codes of ALL ranks gathered row-major,
    saev_backward_override): the gathered rows get their slice-major copies in saev_backward_begin.  Two 'ranks' worth of rows
    through one context: (i) gathered over a single rank's own rows = the plain backward (bit for bit on the row kernels); (ii) over both ranks'
    rows = the row kernels to rounding."""
    xs = [_data(d, 2 * n, n, "dense_latent", seed=5), _data(d, 2 * n, n, "plain", seed=6)]
    out = {}
    for route in ("rows", "slices"):
        eng = _engine(d, s, k, 2 * n, route)
        state = []
        for x in xs:
            eng.step_forward(x)
            eng.step_dead(n)
            g = torch.empty(n, d, device="cuda"); idx = torch.empty(n, k, device="cuda", dtype=torch.int32); val = torch.empty(n, k, device="cuda")
            eng.copy_step_state(n, g, idx, val)
            state.append((x, g, idx, val))
        # (the forward in flight is the second batch)
        eng.step_backward()
        torch.cuda.synchronize()
        plain = eng.grads.clone()
        eng.grad_w_enc_t()
        eng.backward_begin_gathered(*state[1])
        eng.backward_rows(0, s)
        eng.backward_end()
        torch.cuda.synchronize()
        own = eng.grads.clone()
        eng.backward_begin_gathered(*[torch.cat([a, b]).contiguous() for a, b in zip(*state)])
        eng.backward_rows(0, s)
        eng.backward_end()
        torch.cuda.synchronize()
        out[route] = (plain, own, eng.grads.clone())
    # (rows: bit for bit.  slices: the plain backward takes dval = <dL/dx_hat row, decoder row> from the decode, the gathered one
    # forms it in its first pass -- another summation order, so equal to rounding)
    assert torch.equal(out["rows"][0], out["rows"][1])
    p0, p1 = out["slices"][0], out["slices"][1]
    assert (p0 - p1).abs().max().item() <= 2e-6 * p0.abs().max().item() + 1e-12
    a, c = out["rows"][2], out["slices"][2]
    assert (a - c).abs().max().item() <= 2e-6 * a.abs().max().item() + 1e-12
    assert not torch.equal(out["slices"][1], out["slices"][2])


@pytest.mark.parametrize("d,s,k,b,prefixes", [(1024, 8192, 32, 2048, 1), (768, 6144, 32, 1000, 1), (512, 4096, 32, 1024, 4),
                                               (1280, 5120, 64, 512, 1), (128, 1024, 8, 300, 1)])
def test_decode_filled_bit_map_is_bit_identical_to_the_builds_own_fill(d, s, k, b, prefixes):
    """saev_debug_cfg.csc_route: the training decode sets the (latent, row) bits of the backward's pair-list build (default) or the
    build fills its bit map itself (1).  The pair list is the same list either way, so parameters and moments agree bit for bit --
    also across the cases that decide whether the bit map can be trusted: a smaller batch, a forward without a backward (an
    evaluation between two steps leaves the bits of ITS codes behind), and the batch growing back."""
    from saev_amd.nn.objectives import sample_prefixes

    engs = [_engine(d, s, k, b, "slices", seed=3, csc_route=r) for r in (0, 1)]
    xs = [_data(d, b, n, "plain", seed=10 + i) for i, n in enumerate((b, b, max(1, b // 3), b, b))]
    torch.manual_seed(0)
    cuts = [sample_prefixes(s, prefixes) if prefixes > 1 else None for _ in xs]
    for eng in engs:
        for i, x in enumerate(xs):
            if prefixes > 1:
                eng.set_prefixes(cuts[i])
            if i == 3:  # a training forward that no backward follows
                eng.step_forward(x, training=True)
            eng.train_step(x, 1e-3, 1.0)
    torch.cuda.synchronize()
    for name in ("params", "adam_m", "adam_v"):
        assert torch.equal(getattr(engs[0], name), getattr(engs[1], name)), name


@pytest.mark.parametrize("d,s,k,b,n,kind,prefixes", [(1024, 32768, 32, 16384, 16384, "dense_latent", 1), (1024, 8192, 32, 4096, 4096, "few_latents", 1),
                                                      (768, 6144, 32, 4096, 4096, "plain", 1), (512, 4096, 32, 1024, 1000, "plain", 4),
                                                      (128, 1024, 8, 512, 300, "plain", 1), (64, 256, 4, 64, 3, "plain", 1)])
def test_light_finalize_agrees_with_the_row_reading_one(d, s, k, b, n, kind, prefixes):
    """saev_debug_cfg.fin_route: the one-launch finalize forms the projection coefficient <g_i, w_i> / ||w_i||^2, the squares of the
    projected row and of the W_enc^T gradient row and db_enc from per-slice squares, the pair lists (<dW_dec[i], w_i> = sum of
    val * dval over the latent's pairs) and ||w_i||^2 left by normalize_rows; route 1 re-reads every gradient and decoder row, as
    round 4 did.  Same quantities, other summation orders: clip norms agree to fp32 rounding, and so do the parameters after steps
    with the clip active -- including latents cut by run boundaries, unused latents, AuxK rows added behind the finalize."""
    from saev_amd.nn.objectives import sample_prefixes

    thr = 3 * n
    engs = [_engine(d, s, k, b, "slices", seed=5, fin_route=r, k_aux=16, dead_threshold_tokens=thr) for r in (0, 1)]
    xs = [_data(d, b, n, kind, seed=20 + i) for i in range(5)]
    torch.manual_seed(0)
    cuts = [sample_prefixes(s, prefixes) if prefixes > 1 else None for _ in xs]
    norms = [[], []]
    for e, eng in enumerate(engs):
        for i, x in enumerate(xs):
            if prefixes > 1:
                eng.set_prefixes(cuts[i])
            eng.train_step(x, 1e-3, 0.05 if i % 2 else 1.0)
            norms[e].append(eng.read_stats().grad_norm)
    torch.cuda.synchronize()
    # (the routes round differently, so a near-tie of a later step's TopK may fall the other way: the first steps to fp32
    # rounding, the later ones to the size of a flipped code)
    for i, (a, c) in enumerate(zip(*norms)):
        assert math.isclose(a, c, rel_tol=2e-6 if i < 3 else 2e-4), (norms[0], norms[1])
    assert engs[0].read_stats().n_dead == engs[1].read_stats().n_dead
    # (the parameters after five free-running steps: a code that flipped in a later step moves its latent's rows by a whole
    # Adam step of either sign, so the bulk must agree closely and the rest stay within a few learning rates)
    for name in ("params", "adam_m", "adam_v"):
        p0, p1 = getattr(engs[0], name), getattr(engs[1], name)
        bad = ~torch.isclose(p0, p1, rtol=1e-4, atol=1e-7 * p1.abs().max().item())
        assert bad.float().mean().item() <= 2e-2, f"{name}: {bad.sum().item()} of {bad.numel()} elements apart"
    assert (engs[0].params - engs[1].params).abs().max().item() <= 5 * 1e-3
