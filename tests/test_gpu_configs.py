"""BASELINE.json's configurations at their own shapes (needs an MI355X: -m gpu).

configs[0]  d=768 x8, k=32, B=4096: small enough for the CPU oracle -- full train steps, teacher-forced.
configs[2]  d=1024 x32, k=32, B=16384 with the auxiliary loss ACTIVE (single-GPU half; the multi-GPU half is the
            exchange protocol, tests/test_ddp_gloo.py): ~100 and ~2 000 dead latents forced, loss and the dead latents'
            gradients against an fp64 recomputation of reference nn/modeling.py:75-103 over the whole batch.
configs[3]  d=1280 x64 (81 920 latents), k=64, B=16384, bf16 encoder: codes against an fp64 product of the bf16-rounded
            operands on sampled rows, step invariants, replica bit-identity.
Each case runs once, in the encoder mode its config names (the fp32-accurate default, or bf16)."""

import math

import pytest
import torch

import sae_ref as R

pytestmark = [pytest.mark.gpu, pytest.mark.encoder_modes("f16r")]  # full-shape config tests pick their own encoder mode: collected once


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


def test_config0_full_steps_against_the_oracle():
    """configs[0] whole: three teacher-forced train steps (lr = 0, then > 0) at d=768, S=6144, k=32, B=4096 against the CPU
    oracle -- losses, gradient norm, tracker, every parameter."""
    d, s, k, b = 768, 6144, 32, 4096
    eng, _ = build(d, s, k, b, seed=5)
    cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k)
    gen = torch.Generator().manual_seed(6)
    mu = torch.randn(d, generator=gen)
    flips = 0
    for i, lr in enumerate((0.0, 4e-4, 4e-4)):
        x = torch.randn(b, d, generator=gen) + mu
        state = R.TrainState(
            params={k_: v.cpu().clone() for k_, v in eng.param_views().items()},
            m={k_: eng.view(k_, eng.adam_m).cpu().clone() for k_ in R.PARAM_ORDER},
            v={k_: eng.view(k_, eng.adam_v).cpu().clone() for k_ in R.PARAM_ORDER},
            toks_since_active=eng.toks_since_active.cpu().clone(), adam_steps=eng.adam_steps, lr=lr)
        ref = R.train_step(state, x, cfg)
        eng.train_step(x.cuda(), lr, cfg.grad_clip)
        st = eng.read_stats()
        assert st.dense_route == 0 and st.n_overflow_rows == 0
        assert math.isclose(st.mse, ref["mse"], rel_tol=1e-4), (i, st.mse, ref["mse"])  # north_star: 1e-4 rel
        flipped = not math.isclose(st.mse, ref["mse"], rel_tol=2e-6)  # a near-tie resolved the other way (1 / (B k) = 8e-6)
        flips += flipped
        assert st.n_dead == ref["n_dead"] == 0 and math.isclose(st.l0, ref["l0"], rel_tol=1e-6)
        assert math.isclose(st.l1, ref["l1"], rel_tol=1e-4)
        assert math.isclose(st.grad_norm, ref["grad_norm"], rel_tol=1e-3)
        if not flipped:
            assert torch.equal(eng.toks_since_active.cpu(), state.toks_since_active)
        for key in R.PARAM_ORDER:
            bad = ~torch.isclose(eng.view(key).cpu(), state.params[key], rtol=1e-4, atol=2e-6)
            assert bad.float().mean() <= (2e-3 if flipped else 1e-5), f"step {i} {key}: {bad.sum().item()} of {bad.numel()} off"
    assert flips <= 2


def test_config1_full_steps_against_the_oracle():
    """configs[1] -- the benchmark's own shape -- whole: two teacher-forced train steps (lr = 0, then > 0) at d=1024, S=32768,
    k=32, B=16384 against the CPU oracle's restatement of the loop body (reference framework/train.py:332-460): losses,
    gradient norm, tracker, every parameter.  ~12 s of oracle time per step on the GPU box's host cores."""
    d, s, k, b = 1024, 32768, 32, 16384
    eng, _ = build(d, s, k, b, seed=15)
    cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k)
    gen = torch.Generator().manual_seed(16)
    mu = torch.randn(d, generator=gen)
    flips = 0
    for i, lr in enumerate((0.0, 4e-4)):
        x = torch.randn(b, d, generator=gen) + mu
        state = R.TrainState(
            params={k_: v.cpu().clone() for k_, v in eng.param_views().items()},
            m={k_: eng.view(k_, eng.adam_m).cpu().clone() for k_ in R.PARAM_ORDER},
            v={k_: eng.view(k_, eng.adam_v).cpu().clone() for k_ in R.PARAM_ORDER},
            toks_since_active=eng.toks_since_active.cpu().clone(), adam_steps=eng.adam_steps, lr=lr)
        ref = R.train_step(state, x, cfg)
        eng.train_step(x.cuda(), lr, cfg.grad_clip)
        st = eng.read_stats()
        assert st.dense_route == 0 and st.n_overflow_rows == 0
        assert math.isclose(st.mse, ref["mse"], rel_tol=1e-4), (i, st.mse, ref["mse"])  # north_star: 1e-4 rel
        flipped = not math.isclose(st.mse, ref["mse"], rel_tol=1e-6)  # a near-tie resolved the other way (1 / (B k) = 2e-6)
        flips += flipped
        assert st.n_dead == ref["n_dead"] == 0 and math.isclose(st.l0, ref["l0"], rel_tol=1e-6)
        assert math.isclose(st.l1, ref["l1"], rel_tol=1e-4)
        assert math.isclose(st.grad_norm, ref["grad_norm"], rel_tol=1e-3)
        if not flipped:
            assert torch.equal(eng.toks_since_active.cpu(), state.toks_since_active)
        for key in R.PARAM_ORDER:
            bad = ~torch.isclose(eng.view(key).cpu(), state.params[key], rtol=1e-4, atol=2e-6)
            assert bad.float().mean() <= (2e-3 if flipped else 1e-5), f"step {i} {key}: {bad.sum().item()} of {bad.numel()} off"
    assert flips <= 1


def _tie_restatement_to_the_oracle(eng, x, x_hat, dead, k_aux, alpha, diff, A, n_sub=1024):
    """The fp64 restatement of AuxK.loss these full-shape tests check the HIP kernels against, checked itself against the
    PINNED oracle (R.auxk_loss, fixtures G4 / G5) on the first ``n_sub`` rows of the same batch with the same forced dead
    set: the loss over those rows, and -- through autograd, as the reference takes them -- the gradients that reach the dead
    latents' decoder rows, encoder columns and biases.  (The auxiliary loss is a mean over rows: a row subset is a complete
    instance of it, scaled by B / n_sub.)"""
    s, d = eng.cfg.d_sae, eng.cfg.d_model
    xs, xh = x[:n_sub].cpu(), x_hat[:n_sub].cpu()
    W_enc = eng.view("W_enc").cpu().clone().requires_grad_(True)
    b_enc = eng.view("b_enc").cpu().clone().requires_grad_(True)
    W_dec = eng.view("W_dec").cpu().clone().requires_grad_(True)  # (as the step used it: rows normalised at its top)
    b_dec = eng.view("b_dec").cpu().clone().requires_grad_(True)
    mask = torch.zeros(s, dtype=torch.bool)
    mask[dead.cpu()] = True
    h = R.encode_pre(xs, W_enc, b_enc)
    h.retain_grad()
    loss = R.auxk_loss(x=xs, h=h, x_hat_last=xh, dead_mask=mask, W_dec=W_dec, b_dec=b_dec, k_aux=k_aux, alpha=alpha)
    loss.backward()
    sub = diff[:n_sub]
    mine = alpha * (sub * sub).mean().item()
    assert math.isclose(loss.item(), mine, rel_tol=2e-5), (loss.item(), mine)
    dc = dead.cpu()
    # The restatement selects on fp64 pre-activations, the oracle (like the reference) on fp32 ones: with more dead latents than
    # k_aux a near-tie at the cut can fall either way, which moves one (row, latent) term of the gradients.  The selections must
    # agree almost everywhere; the FORMULAS are then tied on the oracle's own selection.
    # (the oracle's selection, exact ties included, read off autograd: dL/dh is non-zero exactly at the kept entries)
    As = A[:n_sub]
    hd = h.detach()[:, dc].double()
    kept = h.grad[:, dc] != 0
    assert (kept.sum(dim=1) == min(k_aux, dc.numel())).all()
    A_or = torch.where(kept, hd, torch.zeros_like(hd)).to(As.device)
    assert ((A_or != 0) != (As != 0)).float().mean().item() <= 1e-4, "restatement and oracle select different dead latents"
    As = A_or
    W_dd, b_dd = eng.view("W_dec")[dead].double(), eng.view("b_dec").double()
    sub = As @ W_dd + b_dd - (x[:n_sub].double() - x_hat[:n_sub].double())
    # the restatement's gradient formulas on the subset (its factor is 2 alpha / (B d); the subset's own is 2 alpha / (n_sub d))
    g_sub = (2.0 * alpha / (n_sub * d)) * sub
    mine_Wdec = (As.t() @ g_sub).cpu()
    dA = (g_sub @ eng.view("W_dec")[dead].double().t()) * (As != 0)
    mine_WencT = (dA.t() @ x[:n_sub].double()).cpu()
    for got, want in ((W_dec.grad[dc].double(), mine_Wdec), (W_enc.grad[:, dc].double().t(), mine_WencT),
                      (b_enc.grad[dc].double(), dA.sum(dim=0).cpu()), (b_dec.grad.double(), g_sub.sum(dim=0).cpu())):
        torch.testing.assert_close(got, want, rtol=2e-3, atol=1e-4 * want.abs().max().item())
    live = torch.ones(s, dtype=torch.bool)
    live[dc] = False
    assert (W_dec.grad[live] == 0).all() and (b_enc.grad[live] == 0).all(), "the auxiliary term reaches dead latents only"


@pytest.mark.parametrize("n_dead", [100, 2000])
def test_config2_auxk_active_at_full_size(n_dead):
    """configs[2], single-GPU half: the auxiliary loss with a forced dead set at d=1024, S=32768, k=32, k_aux=512, B=16384.
    Loss and the dead latents' gradients against an fp64 recomputation of AuxK.loss (modeling.py:75-103) over the WHOLE
    batch: H = x W_enc[:, dead] + b_enc[dead]; keep the min(k_aux, n_dead) largest per row; reconstruct through W_dec[dead]
    + b_dec; alpha * mean((recon - (x - x_hat))^2); gradients reach W_dec[dead], b_dec, W_enc[:, dead], b_enc[dead] only."""
    d, s, k, b, k_aux, alpha, thr = 1024, 32768, 32, 16384, 512, 1 / 32, 1_000_000
    eng, x = build(d, s, k, b, seed=7, k_aux=k_aux, alpha=alpha, dead_threshold_tokens=thr)
    dead = torch.randperm(s, generator=torch.Generator().manual_seed(8))[:n_dead].sort().values.cuda()
    toks = torch.zeros(s, dtype=torch.int64)
    toks[dead.cpu()] = thr
    eng.view("b_enc")[dead] = -100.0  # never among the top-k of the main path: they stay dead
    # dead latents still need distinguishable pre-activations among themselves: vary their bias a little
    eng.view("b_enc")[dead] += 0.5 * torch.randn(n_dead, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
    eng.set_tracker(toks)
    eng.step_forward(x, training=True)
    eng.step_dead(b)
    eng.step_backward()
    st = eng.read_stats()
    assert st.n_dead == n_dead and st.dense_route == 0
    # (100 dead latents: the matrix-core kernels, every dead latent selected; 2 000: dense algebra over the compacted dead set)
    assert eng.aux_route() == (2 if n_dead <= 128 else 3)
    idx, val, x_hat = eng.last_codes(b)
    assert not torch.isin(idx.long(), dead).any(), "a dead latent fired in the main path"
    # W_dec as the step used it (rows normalised at the top of the step)
    W_dec, b_dec = eng.view("W_dec").double(), eng.view("b_dec").double()
    W_enc_d, b_enc_d = eng.view("W_enc")[:, dead].double(), eng.view("b_enc")[dead].double()
    x64, resid = x.double(), x.double() - x_hat.double()
    H = x64 @ W_enc_d + b_enc_d  # (B, n_dead)
    k_use = min(k_aux, n_dead)
    top = torch.topk(H, k_use, dim=1)
    A = torch.zeros_like(H).scatter_(1, top.indices, top.values)
    recon = A @ W_dec[dead] + b_dec
    diff = recon - resid
    aux = alpha * (diff * diff).mean().item()
    assert math.isclose(st.aux, aux, rel_tol=1e-4), (st.aux, aux)
    _tie_restatement_to_the_oracle(eng, x, x_hat, dead, k_aux, alpha, diff, A)
    # gradients of the auxiliary term (the main path does not touch dead latents' rows: they never fire)
    g_aux = (2.0 * alpha / (b * d)) * diff                       # d aux / d recon
    g = eng.grad_views()
    want_Wdec = A.t() @ g_aux                                    # (n_dead, D)
    dA = (g_aux @ W_dec[dead].t()) * (A != 0)                    # straight through the kept entries only
    want_WencT = dA.t() @ x64                                    # (n_dead, D)
    want_benc = dA.sum(dim=0)
    scale = want_Wdec.abs().max().item()
    torch.testing.assert_close(g["W_dec"][dead].double(), want_Wdec, rtol=2e-3, atol=1e-4 * scale)
    scale = want_WencT.abs().max().item()
    torch.testing.assert_close(g["W_enc"][:, dead].double().t(), want_WencT, rtol=2e-3, atol=1e-4 * scale)
    torch.testing.assert_close(g["b_enc"][dead].double(), want_benc, rtol=2e-3, atol=1e-4 * want_benc.abs().max().item())
    dx = (2.0 / (b * d)) * (x_hat.double() - x64)
    want_bdec = dx.sum(dim=0) + g_aux.sum(dim=0)
    torch.testing.assert_close(g["b_dec"].double(), want_bdec, rtol=2e-3, atol=1e-4 * want_bdec.abs().max().item())
    # the tail still runs on top of it and the next step still sees them dead
    eng.step_tail(4e-4, 1.0)
    st2 = eng.read_stats()
    assert math.isfinite(st2.grad_norm) and st2.grad_norm > 0


def test_config3_bf16_full_shape():
    """configs[3] on one GPU: d=1280, 81 920 latents, k=64, B=16384, bf16 encoder.  (1) the fused codes are the top-64 of
    the pre-activations formed from bf16-ROUNDED operands (fp64 product on sampled rows); (2) they are within bf16
    rounding of the fp32 pre-activations; (3) train steps: l0 = k, finite and falling loss, no fallback route, SSE
    identity; (4) two replicas fed the same batches stay bit-identical."""
    d, s, k, b = 1280, 81920, 64, 16384
    eng, x = build(d, s, k, b, seed=11, encoder="bf16", k_aux=0)
    idx, val = eng.encode_topk(x)
    assert idx.shape == (b, k) and (idx[:, 1:] > idx[:, :-1]).all() and idx.min() >= 0 and idx.max() < s
    rows = torch.randperm(b, device="cuda", generator=torch.Generator(device="cuda").manual_seed(12))[:48]
    W, be = eng.view("W_enc"), eng.view("b_enc")
    h_bf = x[rows].to(torch.bfloat16).double() @ W.to(torch.bfloat16).double() + be.double()
    sel = h_bf.gather(1, idx[rows].long())
    torch.testing.assert_close(sel.float(), val[rows], rtol=1e-4, atol=1e-4)
    kth = val[rows].min(dim=1).values.double()
    assert (h_bf.scatter(1, idx[rows].long(), float("-inf")).max(dim=1).values <= kth + 1e-4).all()
    h32 = x[rows].double() @ W.double() + be.double()
    bound = 2.0 ** -7 * (x[rows].double().norm(dim=1, keepdim=True) * W.double().norm(dim=0).max())
    assert ((h_bf - h32).abs() <= bound).all()
    del h_bf, h32, sel
    eng2, _ = build(d, s, k, b, seed=11, encoder="bf16", k_aux=0)
    losses = []
    for i in range(4):
        lr = 0.0 if i == 0 else 4e-4
        eng.train_step(x, lr, 1.0)
        eng2.train_step(x, lr, 1.0)
        st = eng.read_stats()
        losses.append(st.mse)
        assert st.l0 == k and st.n_dead == 0 and st.aux == 0.0 and st.dense_route == 0 and st.n_overflow_rows == 0
        assert math.isfinite(st.grad_norm) and math.isclose(st.sse / (b * d), st.mse, rel_tol=1e-4)
    assert losses[-1] < losses[1], losses
    assert torch.equal(eng.params, eng2.params)


def test_config3_bf16_with_auxk_active_at_full_shape():
    """configs[3] with the SAE default k_aux = 512 and ~500 dead latents forced (the bf16 encoder only rounds the operands
    of the TopK contraction; the auxiliary branch works on exact fp32 pre-activations, DESIGN.md 3.1 (a')): loss and the
    dead latents' gradients against an fp64 recomputation of AuxK.loss (modeling.py:75-103) over the whole batch, at
    d=1280, 81 920 latents, k=64, B=16384."""
    d, s, k, b, k_aux, alpha, thr, n_dead = 1280, 81920, 64, 16384, 512, 1 / 32, 1_000_000, 500
    eng, x = build(d, s, k, b, seed=21, encoder="bf16", k_aux=k_aux, alpha=alpha, dead_threshold_tokens=thr, aux_dead_cap=2048)
    dead = torch.randperm(s, generator=torch.Generator().manual_seed(22))[:n_dead].sort().values.cuda()
    toks = torch.zeros(s, dtype=torch.int64)
    toks[dead.cpu()] = thr
    eng.view("b_enc")[dead] = -100.0
    eng.view("b_enc")[dead] += 0.5 * torch.randn(n_dead, device="cuda", generator=torch.Generator(device="cuda").manual_seed(23))
    eng.set_tracker(toks)
    eng.step_forward(x, training=True)
    eng.step_dead(b)
    eng.step_backward()
    st = eng.read_stats()
    assert st.n_dead == n_dead and st.dense_route == 0 and st.l0 == k and eng.aux_route() == 3
    idx, val, x_hat = eng.last_codes(b)
    assert not torch.isin(idx.long(), dead).any(), "a dead latent fired in the main path"
    W_dec, b_dec = eng.view("W_dec").double(), eng.view("b_dec").double()
    x64, resid = x.double(), x.double() - x_hat.double()
    H = x64 @ eng.view("W_enc")[:, dead].double() + eng.view("b_enc")[dead].double()
    top = torch.topk(H, min(k_aux, n_dead), dim=1)  # n_dead <= k_aux: every dead latent is selected
    A = torch.zeros_like(H).scatter_(1, top.indices, top.values)
    diff = A @ W_dec[dead] + b_dec - resid
    aux = alpha * (diff * diff).mean().item()
    assert math.isclose(st.aux, aux, rel_tol=1e-4), (st.aux, aux)
    _tie_restatement_to_the_oracle(eng, x, x_hat, dead, k_aux, alpha, diff, A, n_sub=512)
    g_aux = (2.0 * alpha / (b * d)) * diff
    g = eng.grad_views()
    want_Wdec = A.t() @ g_aux
    dA = (g_aux @ W_dec[dead].t()) * (A != 0)
    want_WencT = dA.t() @ x64
    for got, want in ((g["W_dec"][dead].double(), want_Wdec), (g["W_enc"][:, dead].double().t(), want_WencT),
                      (g["b_enc"][dead].double(), dA.sum(dim=0))):
        torch.testing.assert_close(got, want, rtol=2e-3, atol=1e-4 * want.abs().max().item())
    # and the tail takes the step: finite norm, every dead latent's decoder row moves
    before = eng.view("W_dec")[dead].clone()
    eng.step_tail(4e-4, 1.0)
    st2 = eng.read_stats()
    assert math.isfinite(st2.grad_norm) and st2.grad_norm > 0
    assert ((eng.view("W_dec")[dead] - before).abs().amax(dim=1) > 0).all()


def test_config4_vit_l14_extraction_feeds_the_sae_at_full_shape():
    """configs[4] at its real shape on one GPU (reference route: data/shards.py:239-273 hooks, :697-850 worker_fn ->
    data/shuffled.py:506-552): a ViT-L/14-shaped transformer (24 blocks, d_model 1024, 257 tokens; random init, bf16
    autocast forward), hooks on block 23, device reservoir, SAE of 32 768 latents trained on 16 384-row batches.
    (1) one epoch delivers every (example, content token) exactly once; (2) each delivered row equals, bit for bit, what the
    hook saw for that (example, token) -- recomputed here with the same image batches outside the feed; (3) three train
    steps drawn from the feed keep the step invariants of the full-size tests (l0 = k, no fallback route, SSE identity,
    falling loss)."""
    from saev_amd import data
    from saev_amd.data.vit import VisionTransformer
    from saev_amd.engine import EngineConfig, SaeEngine

    torch.manual_seed(0)
    n_img, img_batch, layer, B = 192, 32, 23, 16384  # 192 x 256 content tokens = 49 152 rows = three full batches
    vit = VisionTransformer.vit_l14().cuda().eval()
    imgs = torch.randn(n_img, 3, 224, 224, generator=torch.Generator().manual_seed(1))

    class Autocast(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, x):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return self.inner(x)

    rec = data.ActivationRecorder(vit, vit.blocks, layers=(layer,), content_tokens_per_example=256, cls_token=True)
    rec.model = Autocast(vit)

    def images():
        for lo in range(0, n_img, img_batch):
            yield imgs[lo : lo + img_batch], torch.arange(lo, lo + img_batch)

    # what the hook sees, image batch by image batch (the same batches the feed forwards): (n_img, 257, 1024)
    seen = []
    with torch.no_grad():
        for b_, _ in images():
            seen.append(rec(b_.cuda())[1][:, 0].float().clone())
    seen = torch.cat(seen)
    assert seen.shape == (n_img, 257, 1024) and torch.isfinite(seen).all()

    eng = SaeEngine(EngineConfig(d_model=1024, d_sae=32768, top_k=32, max_batch=B, aux_dead_cap=4096))
    g = torch.Generator(device="cuda").manual_seed(2)
    W = (torch.rand(32768, 1024, device="cuda", generator=g) * 2 - 1) * math.sqrt(6.0 / 1024)
    W /= W.norm(dim=1, keepdim=True)
    eng.view("W_dec").copy_(W)
    eng.view("W_enc").copy_(W.t())
    del W
    feed = data.ExtractionFeed(data.ExtractConfig(layer=layer, batch_size=B, buffer_size=2, seed=3), rec, images,
                               n_examples=n_img, d_model=1024, device="cuda", engine=eng)
    assert feed.n_samples == n_img * 256 and len(feed) == 3
    hits = torch.zeros(n_img, 256, dtype=torch.int32, device="cuda")
    losses = []
    for i, batch in enumerate(feed):
        act, ex, tk = batch["act"], batch["example_idx"].long().cuda(), batch["token_idx"].long().cuda()
        assert act.shape == (B, 1024) and act.dtype == torch.float32 and act.is_cuda
        hits.index_put_((ex, tk), torch.ones_like(ex, dtype=torch.int32), accumulate=True)
        assert torch.equal(act, seen[ex, tk + 1]), "a delivered row differs from what the hook recorded for its (example, token)"
        eng.train_step(act, 0.0 if i == 0 else 4e-4, 1.0)
        st = eng.read_stats()
        losses.append(st.mse)
        assert st.l0 == 32 and st.dense_route == 0 and st.n_overflow_rows == 0 and st.n_dead == 0
        assert math.isfinite(st.grad_norm) and st.grad_norm > 0 and math.isclose(st.sse / (B * 1024), st.mse, rel_tol=1e-4)
    assert (hits == 1).all(), "every (example, content token) exactly once per epoch"
    assert len(losses) == 3 and losses[-1] < losses[1], losses
