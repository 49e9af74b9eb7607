"""Pieces of the data-parallel step on one MI355X: the ranged backward must reproduce the monolithic one bit for bit,
and the overlapped gradient exchange (RCCL, one rank) must leave the same parameters as the plain step."""

import socket

import pytest
import torch

import sae_ref as R
from conftest import load_golden
from test_gpu_parity import make_engine

pytestmark = pytest.mark.gpu


def _setup(tag="b", **kw):
    g = load_golden(f"g9_train_{tag}")
    d, s, k, bsz = int(g["d"]), int(g["s"]), int(g["k"]), int(g["bsz"])
    eng = make_engine(d, s, k, k_aux=int(g["k_aux"]), thr=int(g["thr"]), max_batch=bsz, **kw)
    eng.load_params({key: g["init_" + key] for key in R.PARAM_ORDER})
    toks = torch.zeros(s, dtype=torch.int64)
    toks[::9] = int(g["thr"])  # dead latents: the AuxK branch contributes gradient rows too
    eng.set_tracker(toks)
    return eng, g["acts"][:bsz].cuda(), s


@pytest.mark.parametrize("prefixes", [None, [100, 300, 1024]])
def test_ranged_backward_is_bit_identical_to_the_monolithic_one(prefixes):
    grads = []
    for ranges in (None, [(0, 1), (1, 130), (130, 131), (131, 1000), (1000, 1024)]):
        eng, x, s = _setup()
        eng.set_prefixes(prefixes)
        eng.step_forward(x, training=True)
        eng.step_dead(x.shape[0])
        if ranges is None:
            eng.step_backward()
        else:
            wt = eng.grad_w_enc_t()
            eng.backward_begin()
            for lo, hi in ranges:
                eng.backward_rows(lo, hi)
            torch.testing.assert_close(wt.T, wt.T)  # (touch: the host-owned scratch is what the context writes)
            eng.backward_end()
        assert eng.read_stats().n_dead > 0
        grads.append(eng.grads.clone())
    assert torch.equal(grads[0], grads[1])
    assert grads[0].abs().sum() > 0


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_overlapped_exchange_matches_plain_step_on_one_rank(encoder_mode):
    """world_size 1 through RCCL: every async all-reduce is the identity, so the overlapped path must end in exactly
    the parameters of eng.train_step -- this checks bucket bounds, the host-owned transposed-gradient scratch, stream
    ordering of the async works and the final transpose."""
    if encoder_mode != "f16r":
        pytest.skip("one encoder mode is enough here")
    import torch.distributed as dist

    from saev_amd.framework.ddp import DataParallelStepper

    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        out = []
        for mode in ("plain", "flat", "overlap"):
            eng, x, s = _setup()
            stepper = DataParallelStepper(eng, dist if mode != "plain" else None, 1, force=mode != "plain",
                                          overlap=mode == "overlap", n_buckets=5)
            for i in range(3):
                stepper.train_step(x, 1e-3 * i, 1.0)
            torch.cuda.synchronize()
            out.append((eng.params.clone(), eng.adam_m.clone(), eng.toks_since_active.clone()))
        for other in out[1:]:
            for a, b in zip(out[0], other):
                assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()
