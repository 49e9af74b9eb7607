"""The streamed f16r step (saev_debug_cfg.prep_route): one pass over x centred / scaled / normalised with what the
PREVIOUS batch left, W_enc operand images left by the previous step's Adam -- against the full preparation of every step
(prep_route = 1: statistics, centring and both image passes from this batch and the current W_enc).  The first pass is only a
filter; the exact refinement makes the codes those of fp32 arithmetic on either route, so whole training runs must agree BIT FOR
BIT -- through batches whose statistics jump, evaluation forwards in between, parameter writes from outside, non-finite input."""

import math

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.encoder_modes("f16r")]  # the streamed preparation belongs to the f16r encoder


def _engine(d, s, k, b, prep_route, seed=0, **kw):
    from saev_amd.engine import EngineConfig, SaeEngine

    eng = SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=k, max_batch=b, prep_route=prep_route, **kw))
    g = torch.Generator(device="cuda").manual_seed(seed)
    W = (torch.rand(s, d, device="cuda", generator=g) * 2 - 1) * math.sqrt(6.0 / d)
    W /= W.norm(dim=1, keepdim=True)
    eng.view("W_dec").copy_(W)
    eng.view("W_enc").copy_(W.t() + 0.01 * torch.randn(d, s, device="cuda", generator=g))
    eng.view("b_enc").copy_(0.05 * torch.randn(s, device="cuda", generator=g))
    eng.view("b_dec").copy_(0.05 * torch.randn(d, device="cuda", generator=g))
    return eng


def _batches(d, n, count, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    mu = torch.randn(d, device="cuda", generator=g)
    return [torch.randn(n, d, device="cuda", generator=g) + mu for _ in range(count)]


def _same(e0, e1, what=""):
    for name in ("params", "adam_m", "adam_v"):
        assert torch.equal(getattr(e0, name), getattr(e1, name)), f"{what}: {name} differ"
    assert torch.equal(e0.toks_since_active, e1.toks_since_active)


@pytest.mark.parametrize("d,s,k,b,n", [(1024, 32768, 32, 16384, 16384), (768, 6144, 32, 4096, 4096), (256, 2048, 16, 1024, 1000),
                                        (64, 512, 8, 256, 200)])
def test_streamed_steps_equal_fully_prepared_steps_bit_for_bit(d, s, k, b, n):
    engs = [_engine(d, s, k, b, r, seed=3, k_aux=32, dead_threshold_tokens=3 * n) for r in (0, 1)]
    xs = _batches(d, n, 7)
    for i, x in enumerate(xs):
        for eng in engs:
            eng.train_step(x, 1e-3, 0.05 if i % 2 else 1.0)
            st = eng.read_stats()
            assert st.dense_route == 0 and st.n_overflow_rows == 0, (i, st)
        a, c = (e.read_stats() for e in engs)
        assert a.mse == c.mse and a.l1 == c.l1 and a.grad_norm == c.grad_norm and a.n_dead == c.n_dead, (i, a, c)
    torch.cuda.synchronize()
    _same(*engs)


def test_streamed_step_survives_what_happens_between_steps():
    """Evaluation forwards, API encodes, a smaller batch, parameter writes through torch (noticed by the version counter) and through
    a raw .data write announced with params_touched(): after each, the streamed route must still agree with the full preparation."""
    d, s, k, b = 256, 2048, 16, 1024
    engs = [_engine(d, s, k, b, r, seed=4) for r in (0, 1)]
    xs = _batches(d, b, 12, seed=5)
    other = _batches(d, 300, 3, seed=6)
    for i, x in enumerate(xs):
        for eng in engs:
            if i == 2:
                eng.step_forward(other[0], training=False)       # an evaluation batch with other statistics
                eng.step_forward(other[1], training=False)
            if i == 4:
                eng.encode_topk(other[2])                        # the API op rebuilds the images itself
            if i == 5:
                x_used = x[:700].contiguous()                    # a smaller batch
            else:
                x_used = x
            if i == 6:
                eng.view("W_enc").mul_(1.01)                     # torch in-place: the version counter moves
            if i == 8:
                eng.view("b_enc").data.add_(0.01)                # a raw write ...
                eng.params_touched()                             # ... announced
            if i == 9:  # the phases instead of the fused step: their Adam leaves no images
                eng.step_forward(x_used, training=True)
                eng.step_dead(x_used.shape[0])
                eng.step_backward()
                eng.tail_prepare()
                eng.tail_apply(1e-3, 1.0)
            else:
                eng.train_step(x_used, 1e-3, 1.0)
        a, c = (e.read_stats() for e in engs)
        assert a.mse == c.mse and a.dense_route == c.dense_route == 0, (i, a, c)
        idx0, val0, _ = engs[0].last_codes(x_used.shape[0])
        idx1, val1, _ = engs[1].last_codes(x_used.shape[0])
        assert torch.equal(idx0, idx1) and torch.equal(val0, val1), i
    torch.cuda.synchronize()
    _same(*engs)


def test_api_encode_after_a_streamed_step_sees_an_announced_parameter_write():
    """Round-5 advisor finding: saev_encode_topk keyed on the `stream_step` the last training forward had left, skipped the
    preparation and encoded with the images of the parameters as they WERE.  Streamed training steps, a write to a few elements
    of b_enc / W_enc (announced), then the API encode: the codes must be those of a fresh engine holding the same parameters --
    and the call must leave the step statistics alone."""
    d, s, k, b = 256, 2048, 16, 512
    eng = _engine(d, s, k, b, 0, seed=14)
    xs = _batches(d, b, 5, seed=15)
    for x in xs[:4]:
        eng.train_step(x, 1e-3, 1.0)
    before = eng.read_stats()
    eng.view("b_enc").data[7] += 50.0            # one latent now leads every row
    eng.view("W_enc").data[:, 11] *= -3.0        # and one column changes sign and size
    eng.params_touched()
    idx, val = eng.encode_topk(xs[4])
    fresh = _engine(d, s, k, b, 1, seed=99)
    fresh.load_params({n: eng.view(n).clone() for n in ("W_dec", "b_dec", "W_enc", "b_enc")})
    idx_f, val_f = fresh.encode_topk(xs[4])
    assert (idx == 7).any(dim=1).all(), "the raised bias is in every row's codes"
    assert torch.equal(idx, idx_f) and torch.equal(val, val_f)
    after = eng.read_stats()
    assert after.mse == before.mse and after.l0 == before.l0, "an API encode does not clear the last step's statistics"


def test_a_jump_in_the_data_takes_the_exact_route_and_recovers():
    """The x images of a streamed step are scaled with the previous batch's maximum and centred on its mean.  A batch a thousand
    times larger leaves fp16's range: the step must notice and take the exact dense route; so does the first small batch after
    the large ones (centred on THEIR mean, its margins keep every candidate).  Each costs one slow step, never a wrong code: the
    run agrees with the fully prepared one to the rounding of the dense route's own fp32 summation order, and is back on the
    fused route afterwards."""
    d, s, k, b = 256, 2048, 16, 512
    engs = [_engine(d, s, k, b, r, seed=7) for r in (0, 1)]
    xs = _batches(d, b, 9, seed=8)
    xs[3] = xs[3] * 1000.0 + 50.0
    xs[4] = xs[4] * 1000.0 + 50.0
    routes = []
    for i, x in enumerate(xs):
        for eng in engs:
            eng.train_step(x, 1e-4, 1.0)
        a, c = (e.read_stats() for e in engs)
        routes.append(a.dense_route)
        assert c.dense_route == 0 and math.isclose(a.mse, c.mse, rel_tol=1e-5), (i, a, c)
        (i0, v0, _), (i1, v1, _) = (e.last_codes(b) for e in engs)
        assert (i0 != i1).float().mean().item() <= 1e-3, i
    assert routes[:3] == [0, 0, 0] and routes[3] == 1 and routes[-2:] == [0, 0] and sum(routes) <= 3, routes
    torch.cuda.synchronize()
    for name in ("params", "adam_m"):
        p0, p1 = getattr(engs[0], name), getattr(engs[1], name)
        bad = ~torch.isclose(p0, p1, rtol=1e-3, atol=1e-6 * p1.abs().max().item())
        assert bad.float().mean().item() <= 1e-3, f"{name}: {bad.sum().item()} of {bad.numel()} elements apart"


def test_streamed_step_with_non_finite_input_matches_the_full_preparation():
    d, s, k, b = 128, 1024, 8, 256
    engs = [_engine(d, s, k, b, r, seed=9) for r in (0, 1)]
    xs = _batches(d, b, 6, seed=10)
    xs[2] = xs[2].clone()
    xs[2][5, 7] = float("inf")
    for i, x in enumerate(xs[:3]):
        for eng in engs:
            eng.step_forward(x, training=False) if i == 2 else eng.train_step(x, 1e-3, 1.0)
        a, c = (e.read_stats() for e in engs)
        assert a.dense_route == c.dense_route, (i, a, c)
        for u, v in zip(engs[0].last_codes(b)[:2], engs[1].last_codes(b)[:2]):
            assert torch.equal(u, v), i
    for x in xs[3:]:  # and finite batches after it run fused again, identically
        for eng in engs:
            eng.train_step(x, 1e-3, 1.0)
    a, c = (e.read_stats() for e in engs)
    assert a.dense_route == c.dense_route == 0 and a.mse == c.mse
    _same(*engs)


def test_train_step_gather_equals_gather_then_train_step():
    """saev_train_step_gather: the batch is drawn from the pool by the step's first kernel (which leaves it as a contiguous
    matrix on its way) -- same parameters as saev_gather_rows followed by saev_train_step, on the streamed and the full route."""
    d, s, k, b = 256, 2048, 16, 1024
    pool = torch.cat(_batches(d, b, 6, seed=11))
    g = torch.Generator().manual_seed(12)
    for route in (0, 1):
        e0, e1 = (_engine(d, s, k, b, route, seed=13) for _ in range(2))
        for i in range(6):
            rows = torch.randperm(pool.shape[0], generator=g)[:b].cuda()
            x = e0.train_step_gather(pool, rows, 1e-3, 1.0)
            assert torch.equal(x, pool[rows])
            e1.train_step(e1.gather_rows(pool, rows), 1e-3, 1.0)
            assert e0.read_stats().mse == e1.read_stats().mse
        torch.cuda.synchronize()
        _same(e0, e1, f"route {route}")


@pytest.mark.parametrize("d,s,k,b", [(1024, 8192, 32, 2048), (768, 6144, 32, 1000), (256, 2048, 16, 300)])
def test_shares_added_by_the_final_select_equal_the_separate_pass(d, s, k, b):
    """saev_debug_cfg.fwd_route = 2 keeps refine_sum_kernel; the default lets the final select add a survivor's D / 32 shares itself,
    in the same order: codes, values and therefore whole runs are bit-identical."""
    from saev_amd.engine import EngineConfig, SaeEngine  # noqa: F401

    engs = [_engine(d, s, k, b, 0, seed=21, fwd_route=r) for r in ("default", "sum_pass")]
    for i, x in enumerate(_batches(d, b, 4, seed=22)):
        for eng in engs:
            eng.train_step(x, 1e-3, 1.0)
        (i0, v0, _), (i1, v1, _) = (e.last_codes(b) for e in engs)
        assert torch.equal(i0, i1) and torch.equal(v0, v1), i
    _same(*engs)


def test_bf16_images_left_by_adam_equal_a_fresh_split():
    """The bf16 encoder (configs[3]): the fused Adam leaves the bf16 images of the W_enc it writes (AdamImageArgs::mode 1), the next
    forward skips its pass over W_enc.  Rounding is rounding: runs with and without (prep_route = 1) agree bit for bit, across an
    evaluation forward and a parameter write in between."""
    d, s, k, b = 256, 2048, 16, 512
    engs = [_engine(d, s, k, b, r, seed=31, encoder="bf16") for r in (0, 1)]
    xs = _batches(d, b, 7, seed=32)
    for i, x in enumerate(xs):
        for eng in engs:
            if i == 3:
                eng.step_forward(xs[0], training=False)
            if i == 5:
                eng.view("W_enc").mul_(0.99)
            eng.train_step(x, 1e-3, 1.0)
        a, c = (e.read_stats() for e in engs)
        assert a.mse == c.mse and a.dense_route == c.dense_route == 0, (i, a, c)
    _same(*engs)


def test_a_write_through_a_module_parameter_drops_what_the_engine_kept():
    """The four Parameters of nn.SparseAutoencoder are views of the engine's buffer with version counters of their own.  An
    in-place write through one of them between two forwards must reach the engine (SparseAutoencoder._eng compares the counters):
    the second forward's codes are those of the new weights."""
    from saev_amd.nn import modeling as M

    torch.manual_seed(0)
    cfg = M.SparseAutoencoderConfig(d_model=256, d_sae=2048, reinit_blend=0.0, activation=M.TopK(top_k=16))
    sae = M.SparseAutoencoder(cfg).cuda().eval()
    x = _batches(256, 300, 2, seed=41)
    with torch.no_grad():
        a0, a1 = sae(x[0]), sae(x[1])   # (the second forward of unchanged weights may reuse the images of the first)
        keep = (a1.idx.clone(), a1.val.clone())  # (the engine's buffers are overwritten by the next forward)
        sae.W_enc.mul_(-1.0)            # every pre-activation changes sign
        sae.b_enc.add_(0.25)
        got = sae(x[1])
        fresh = M.SparseAutoencoder(cfg).cuda().eval()
        fresh.load_state_dict(sae.state_dict())
        want = fresh(x[1])
    assert torch.equal(got.idx, want.idx) and torch.equal(got.val, want.val) and torch.equal(got.x_hats, want.x_hats)
    assert not torch.equal(keep[0], got.idx.to(keep[0].device))


def test_an_unannounced_parameter_write_is_caught_on_the_device():
    """The safety net behind the ownership contract: a bulk write to W_enc through .data (torch's version counter does not move, nobody
    calls params_touched) between two steps.  The streamed step's first kernel compares samples of W_enc / b_enc with the copies its
    images came with, finds them changed, and the step takes the exact dense route: same codes as an engine that was told; the host
    learns of it and the run is back on the fused route afterwards."""
    d, s, k, b = 256, 2048, 16, 512
    told, untold = (_engine(d, s, k, b, 0, seed=51) for _ in range(2))
    xs = _batches(d, b, 8, seed=52)
    routes = []
    for i, x in enumerate(xs):
        for eng in (told, untold):
            if i == 3:
                eng.view("W_enc").data.mul_(-1.0)       # (a raw write: invisible to the version counter)
                if eng is told:
                    eng.params_touched()
            if i == 5:
                eng.view("b_enc").data.add_(0.5)
                if eng is told:
                    eng.params_touched()
            eng.train_step(x, 1e-4, 1.0)
        a, c = told.read_stats(), untold.read_stats()
        routes.append(c.dense_route)
        assert a.dense_route == 0 and math.isclose(a.mse, c.mse, rel_tol=1e-5), (i, a, c)
        (i0, v0, _), (i1, v1, _) = (e.last_codes(b) for e in (told, untold))
        assert (i0 != i1).float().mean().item() <= 1e-3, i
    assert routes[3] == 1 and routes[5] == 1 and routes[:3] == [0, 0, 0] and routes[-1] == 0, routes


@pytest.mark.parametrize("what", ["one_element", "one_column", "b_enc_one"])
def test_a_sparse_unannounced_write_is_never_silent(what):
    """Round-5 review, weak #13: a write to a handful of elements of W_enc through .data (the classic dead-latent re-initialisation)
    slipped past the sampled comparison five times in six and the step ran on stale operand images without a word.  Now: b_enc is
    compared in full before the images are used (exact route, correct codes), and the fused Adam's tile checksums find ANY change
    of W_enc at the end of the first step that used the stale images -- the next call fails with SAEV_STALE_PARAMS, once, and the
    run continues from a fresh preparation."""
    from saev_amd import _lib

    d, s, k, b = 256, 2048, 16, 512
    eng = _engine(d, s, k, b, 0, seed=61)
    xs = _batches(d, b, 9, seed=62)
    for x in xs[:3]:
        eng.train_step(x, 1e-4, 1.0)
    torch.cuda.synchronize()
    if what == "one_element":
        eng.view("W_enc").data[17, 1234] += 0.25
    elif what == "one_column":
        eng.view("W_enc").data[:, 77] = torch.randn(d, device="cuda") / d**0.5
    else:
        eng.view("b_enc").data[5] += 1.0
    raised, dense = 0, []
    for x in xs[3:]:
        try:
            eng.train_step(x, 1e-4, 1.0)
            torch.cuda.synchronize()
            dense.append(eng.read_stats().dense_route)
        except _lib.SaevError as e:
            raised += 1
            assert "saev_params_touched" in str(e) and "tiles" in str(e), str(e)
    if what == "b_enc_one":
        assert raised == 0 and dense[0] == 1 and dense[-1] == 0, (raised, dense)   # found before use: one exact step
    else:
        # found by the samples before use (one exact step) or by the checksums after it (one loud failure) -- never neither
        assert raised + sum(dense) >= 1 and raised <= 1, (raised, dense)
        assert dense[-1] == 0
    # ... and the run goes on: the last steps agree with an engine that was told
    told = _engine(d, s, k, b, 0, seed=61)
    told.load_params({n: eng.view(n).clone() for n in ("W_dec", "b_dec", "W_enc", "b_enc")})
    i0, v0 = eng.encode_topk(xs[0])
    i1, v1 = told.encode_topk(xs[0])
    assert torch.equal(i0, i1) and torch.equal(v0, v1)


def test_tile_checksums_stay_quiet_on_an_honest_run():
    """No false alarms: streamed steps, evaluation forwards in between, announced writes, smaller batches, a step that takes the
    exact route for its own reasons (a jump in the data) -- the late check never fires."""
    d, s, k, b = 256, 2048, 16, 512
    eng = _engine(d, s, k, b, 0, seed=71, k_aux=32, dead_threshold_tokens=3 * b)
    xs = _batches(d, b, 14, seed=72)
    other = _batches(d, 300, 2, seed=73)
    for i, x in enumerate(xs):
        if i == 3:
            eng.step_forward(other[0], training=False)
        if i == 5:
            eng.view("W_enc").data[:, 3] *= 2.0
            eng.params_touched()
        if i == 7:
            eng.view("W_enc").mul_(1.001)  # the version counter announces it
        if i == 9:
            x = x * 1000.0 + 50.0          # leaves fp16's range: exact route on the device, images untouched
        eng.train_step(x[:400].contiguous() if i == 11 else x, 1e-4, 1.0)
    torch.cuda.synchronize()
    eng.step_forward(other[1], training=False)  # (would raise)


def test_module_parameter_writes_are_seen_by_an_engine_driven_directly():
    """train() drives SaeEngine.train_step, not the module: an in-place write through the module's Parameters (their version counters
    are their own) must still reach the engine (SaeEngine.watch) -- round-5 advisor finding."""
    from saev_amd.nn import modeling as M

    cfg = M.SparseAutoencoderConfig(d_model=256, d_sae=2048, activation=M.TopK(top_k=16))
    sae = M.SparseAutoencoder(cfg).cuda()
    eng = sae._eng(512)
    xs = _batches(256, 512, 5, seed=81)
    for x in xs[:3]:
        eng.train_step(x, 1e-4, 1.0)
    with torch.no_grad():
        sae.W_enc[:, 9] = 0.0          # through the Parameter: only ITS version counter moves
        sae.b_enc[9] = 100.0
    eng.train_step(xs[3], 1e-4, 1.0)   # no error now or later, and the write is in the codes
    torch.cuda.synchronize()
    idx, _, _ = eng.last_codes(512)
    assert (idx == 9).any(dim=1).all() and eng.read_stats().dense_route == 0
    eng.train_step(xs[4], 1e-4, 1.0)
    torch.cuda.synchronize()


@pytest.mark.parametrize("group_route", [0, 1])
def test_a_group_streams_through_writes_evaluations_and_relinking(group_route):
    """saev_share_x with the streamed preparation: the lender streams, the followers run on W images their own Adam
    left -- through an announced write to a FOLLOWER's parameters (its images are dropped: it prepares its W side with the centre the
    lender kept), one to the LENDER's (it prepares from scratch: the followers' images no longer match its centre), an evaluation
    forward of the whole group in between, a smaller batch, and a member that leaves the group.  Every member must end with bit for
    bit the parameters it gets when trained alone on the same batches with the same interventions; group_route = 1 is round 5's
    behaviour (every member prepares from scratch on every step), the same contract."""
    d, s, k, b = 256, 2048, 16, 512
    xs = _batches(d, b, 12, seed=91)
    ev = _batches(d, 300, 1, seed=92)[0]

    def run(members, grouped):
        engs = [_engine(d, s, k, b, 0, seed=93 + j, group_route=group_route, k_aux=32, dead_threshold_tokens=3 * b) for j in members]
        if grouped:
            for e in engs[1:]:
                e.share_x(engs[0])
        for i, x in enumerate(xs):
            if i == 4:
                for j, e in zip(members, engs):
                    if j == 1:
                        e.view("W_enc").data[:, 5] *= 1.5
                        e.params_touched()
            if i == 6:
                for e in engs:
                    e.step_forward(ev, training=False)
            if i == 8:
                for j, e in zip(members, engs):
                    if j == 0:
                        e.view("b_enc").data.add_(0.01)
                        e.params_touched()
            if i == 10 and grouped and len(engs) > 2:
                engs[2].share_x(None)  # leaves the group: on its own from here on
            x_used = x[:400].contiguous() if i == 9 else x
            for e in engs:
                e.train_step(x_used, 1e-3, 1.0)
                assert e.read_stats().dense_route == 0, i
        torch.cuda.synchronize()
        return engs

    group = run([0, 1, 2], True)
    for j in range(3):
        alone = run([j], False)[0]
        assert torch.equal(alone.params, group[j].params), f"member {j}: the group run differs from the single run"
        assert torch.equal(alone.adam_v, group[j].adam_v) and torch.equal(alone.toks_since_active, group[j].toks_since_active)
