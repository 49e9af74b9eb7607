"""CPU-only tests of the host side: C-ABI surface, schedules, checkpoints, config grouping, the shard
cache reader and the device-resident loader (run on CPU tensors here)."""

import dataclasses
import json
import math
import pathlib
import re
import subprocess

import numpy as np
import pytest
import torch

import sae_ref as R
from conftest import ROOT, load_golden


# ---- C ABI ----------------------------------------------------------------------------------


def header_functions():
    text = (ROOT / "include" / "saev_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(saev_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    subprocess.run(["make", "-C", str(ROOT)], check=True, capture_output=True)
    from saev_amd import _lib

    lib = _lib.load()
    assert lib.saev_abi_version() == _lib.ABI_VERSION
    declared = header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/saev_amd.h but not exported"
    assert set(_lib.EXPORTED_SYMBOLS) == set(declared), set(_lib.EXPORTED_SYMBOLS) ^ set(declared)


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors of saev_cfg / saev_step_stats against what a C compiler makes of include/saev_amd.h."""
    import ctypes

    from saev_amd import _lib

    fields = {"saev_cfg": [f for f, _ in _lib.SaevCfg._fields_], "saev_step_stats": [f for f, _ in _lib.SaevStepStats._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "saev_amd.h"', "int main(void) {"]
    for st, fs in fields.items():
        src.append(f'printf("{st} %zu\\n", sizeof({st}));')
        src += [f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));' for f in fs]
    src.append("return 0; }")
    (tmp_path / "layout.c").write_text("\n".join(src))
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(tmp_path / "layout.c"), "-o", str(tmp_path / "layout")], check=True)
    want = dict(line.split() for line in subprocess.run([str(tmp_path / "layout")], check=True, capture_output=True, text=True).stdout.splitlines())
    for st, cls in (("saev_cfg", _lib.SaevCfg), ("saev_step_stats", _lib.SaevStepStats)):
        assert ctypes.sizeof(cls) == int(want[st]), st
        for f in fields[st]:
            assert getattr(cls, f).offset == int(want[f"{st}.{f}"]), f"{st}.{f}"


def test_no_gpu_means_loud_failure_not_fallback():
    from saev_amd import _lib
    from saev_amd.engine import EngineConfig, SaeEngine
    from saev_amd.nn import modeling as M

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.SaevError, match="no CPU path"):
        SaeEngine(EngineConfig(d_model=16, d_sae=32, top_k=4))
    sae = M.SparseAutoencoder(M.SparseAutoencoderConfig(d_model=16, d_sae=32, activation=M.TopK(top_k=4)))
    with pytest.raises(RuntimeError, match="no CPU path"):
        sae.encode(torch.zeros(2, 16))


def test_product_package_never_imports_the_oracle():
    for py in (ROOT / "saev_amd").rglob("*.py"):
        src = py.read_text()
        assert "sae_ref" not in src and "import oracle" not in src and "from oracle" not in src, py


# ---- schedules -------------------------------------------------------------------------------


def test_schedules_match_golden_and_oracle():
    from saev_amd.utils import scheduling as S

    g = load_golden("g11_schedule")
    for tag in "abc":
        a = g[f"args_{tag}"].tolist()
        sc = S.WarmupCosine(a[0], int(a[1]), a[2], int(a[3]), a[4])
        got = [sc.step() for _ in range(len(g[f"lr_{tag}"]))]
        np.testing.assert_array_equal(got, g[f"lr_{tag}"].numpy())

    class DL:
        def __init__(self, n, b):
            self.n, self.batch_size, self.drop_last = n, b, False

        def __iter__(self):
            for i in range(0, self.n, self.batch_size):
                yield {"act": torch.zeros(min(self.batch_size, self.n - i), 1)}

    for n_rows, bsz, n_train, ln, n_steps, n_seen in g["limiter"].tolist():
        lim = S.BatchLimiter(DL(n_rows, bsz), n_train)
        sizes = [len(b["act"]) for b in lim]
        assert (len(lim), len(sizes), sum(sizes)) == (ln, n_steps, n_seen)
    assert lim.batch_size == 128 and lim.n == 4096  # attribute pass-through
    # data parallel: each rank sees batch_size / world rows per step but the limiter counts global rows, so the step
    # count equals the single-process one (the limiter used to run world x too many steps)
    class Half(DL):
        def __iter__(self):
            for b in DL.__iter__(self):
                yield {"act": b["act"][: len(b["act"]) // 2]}

    for n_rows, bsz, n_train, ln, n_steps, _ in g["limiter"].tolist():
        if n_rows % bsz == 0:
            lim2 = S.BatchLimiter(Half(n_rows, bsz), n_train, rows_scale=2)
            assert (len(lim2), sum(1 for _ in lim2)) == (ln, n_steps)


def test_sample_prefixes_equals_the_reference_draws():
    """Fixture G15: the reference's own Matryoshka prefix draws (objectives.py:159-201) under fixed seeds; the product
    and the oracle must consume torch's global RNG identically (same draws, same order)."""
    from saev_amd.nn import objectives as O

    g = load_golden("g15_sample_prefixes")
    for impl in (O.sample_prefixes, R.sample_prefixes):
        for seed in g["seeds"].tolist():
            torch.manual_seed(seed)
            for d_sae, n in g["cases"].tolist():
                got = torch.stack([impl(d_sae, n) for _ in range(3)])
                assert got.dtype == torch.int64 and torch.equal(got, g[f"s{seed}_{d_sae}_{n}"]), (seed, d_sae, n)
        torch.manual_seed(7)
        got = torch.stack([impl(1000, 8, pareto_power=1.5) for _ in range(3)])
        assert torch.equal(got, g["s7_alt_1000_8"])


# ---- checkpoints -----------------------------------------------------------------------------


def test_dump_matches_reference_header_and_roundtrips(tmp_path):
    from saev_amd.nn import modeling as M

    g = load_golden("g12_checkpoint")
    want = json.loads(bytes(g["header_json"].numpy().tolist()).decode())
    cfg = M.SparseAutoencoderConfig(d_model=16, d_sae=48, reinit_blend=0.0,
                                    activation=M.TopK(top_k=4, aux=M.AuxK(k_aux=7, alpha=0.125)))
    sae = M.SparseAutoencoder(cfg)
    with torch.no_grad():
        for k in R.PARAM_ORDER:
            getattr(sae, k).copy_(g["sd_" + k])
    M.dump(tmp_path / "sae.pt", sae)
    raw = (tmp_path / "sae.pt").read_bytes()
    header = json.loads(raw.split(b"\n", 1)[0])
    assert header["schema"] == want["schema"] == 5
    assert header["cfg"] == want["cfg"], "cfg block identical to what the reference's nn.dump writes"
    assert set(header) == set(want)
    back = M.load(tmp_path / "sae.pt")
    assert back.cfg == cfg and list(back.state_dict()) == ["W_dec", "b_dec", "W_enc", "b_enc"]
    for k in R.PARAM_ORDER:
        assert torch.equal(getattr(back, k), g["sd_" + k])
    assert back.W_enc.data_ptr() != back.W_dec.data_ptr()


def test_load_reads_legacy_schemas(tmp_path):
    """Every older header layout the reference's nn.load reads (modeling.py:586-645) gives the config the REFERENCE's
    loader makes of it (fixture G16: the reference read these very headers) and the stored parameters."""
    import dataclasses
    import io

    from saev_amd.nn import modeling as M

    g = load_golden("g16_legacy_checkpoints")
    sd = {k: g["sd_" + k] for k in R.PARAM_ORDER}
    blob = io.BytesIO()
    torch.save({k: sd[k] for k in ("W_dec", "b_dec", "W_enc", "b_enc")}, blob)
    names = [str(n) for n in g["names"]]
    assert len(names) == 8
    for name in names:
        hdr = bytes(g["hdr_" + name].numpy().tolist())
        want = json.loads(bytes(g["cfg_" + name].numpy().tolist()).decode())
        (tmp_path / "old.pt").write_bytes(hdr + b"\n" + blob.getvalue())
        sae = M.load(tmp_path / "old.pt")
        got = dataclasses.asdict(sae.cfg)
        got["activation"] = M._ser(sae.cfg.activation)
        assert got == want, name
        for k in R.PARAM_ORDER:
            assert torch.equal(getattr(sae, k), sd[k]), (name, k)


def test_load_legacy_edge_cases(tmp_path):
    import io

    from saev_amd.nn import modeling as M

    sae = M.SparseAutoencoder(M.SparseAutoencoderConfig(d_model=8, d_sae=16, activation=M.TopK(top_k=2)))
    buf = io.BytesIO()
    torch.save(sae.state_dict(), buf)

    def load(hdr):
        (tmp_path / "old.pt").write_bytes(json.dumps(hdr).encode() + b"\n" + buf.getvalue())
        return M.load(tmp_path / "old.pt")

    # schema 1 with the activation named beside a flat config that carries its own top_k (the reference's loader trips over
    # the extra key; the value is what the trainer of that time used, so it is honoured here)
    assert load({"schema": 1, "cls": "TopK", "cfg": {"d_model": 8, "d_sae": 16, "top_k": 2}}).cfg.activation == M.TopK(top_k=2)
    with pytest.raises(ValueError, match="schema 99 is not supported"):
        load({"schema": 99, "cfg": {}})
    with pytest.raises(ValueError, match="exp_factor but no d_model"):
        load({"schema": 2, "cfg": {"exp_factor": 2, "activation": {"cls": "TopK", "params": {"top_k": 2}}}})
    with pytest.raises(ValueError, match="both 'kind' and 'key'"):
        load({"schema": 3, "cfg": {"d_model": 8, "d_sae": 16, "activation": {"cls": "TopK", "params": {"kind": "top-k", "key": "top-k"}}}})
    with pytest.raises(ValueError, match="does not have"):
        load({"schema": 4, "cfg": {"d_model": 8, "d_sae": 16, "activation": {"cls": "Gelu", "params": {}}}})
    # schema 5 is strict: the older spellings are not accepted there
    with pytest.raises(TypeError):
        load({"schema": 5, "cfg": {"d_model": 8, "d_sae": 16, "activation": {"cls": "TopK", "params": {"kind": "top-k"}}}})


def test_reference_loader_reads_our_checkpoint(tmp_path):
    import _refshim

    if not _refshim.available():
        pytest.skip("reference checkout not present (GPU box)")
    from saev_amd.nn import modeling as M

    ref = _refshim.install()
    sae = M.SparseAutoencoder(M.SparseAutoencoderConfig(d_model=16, d_sae=48, activation=M.TopK(top_k=4)))
    M.dump(tmp_path / "ours.pt", sae)
    theirs = ref.modeling.load(tmp_path / "ours.pt")
    assert torch.equal(theirs.W_enc, sae.W_enc) and theirs.cfg.activation.top_k == 4
    ref.modeling.dump(tmp_path / "theirs.pt", theirs)
    again = M.load(tmp_path / "theirs.pt")
    assert again.cfg == sae.cfg and torch.equal(again.b_dec, sae.b_dec)


def test_module_surface_matches_reference_shapes():
    from saev_amd.nn import modeling as M

    cfg = M.SparseAutoencoderConfig()
    assert (cfg.d_model, cfg.d_sae, cfg.reinit_blend, cfg.remove_parallel_grads, cfg.normalize_w_dec) == (1024, 16384, 0.8, True, True)
    assert cfg.activation == M.TopK(top_k=32, sparsity=M.NoSparsity(), aux=M.AuxK(k_aux=512, alpha=1 / 32))
    sae = M.SparseAutoencoder(M.SparseAutoencoderConfig(d_model=12, d_sae=40))
    assert sae.W_dec.shape == (40, 12) and sae.W_enc.shape == (12, 40) and sae.b_dec.shape == (12,) and sae.b_enc.shape == (40,)
    torch.testing.assert_close(sae.W_dec.norm(dim=1), torch.ones(40))
    assert torch.equal(sae.W_enc, sae.W_dec.T) and (sae.b_enc == 0).all()
    with pytest.raises(AssertionError):
        M.TopK(top_k=0)


# ---- config grouping (reference tests/test_framework_train.py:14-61) ------------------------------


def test_split_cfgs_groups_by_data_not_by_hyperparams():
    from saev_amd import data
    from saev_amd.framework import train as T

    def mk(path):
        return T.Config(train_data=data.ShuffledConfig(shards=pathlib.Path(path)), val_data=data.ShuffledConfig(shards=pathlib.Path(path)))

    a1, a2 = dataclasses.replace(mk("/p/a"), lr=1e-3), dataclasses.replace(mk("/p/a"), lr=2e-3, seed=7)
    b = mk("/p/b")
    groups = T.split_cfgs([a1, a2, b])
    assert sorted(len(g) for g in groups) == [1, 2]
    for grp in groups:
        assert len({c.train_data.shards for c in grp}) == 1
        for c in grp:
            assert c.train_data.seed == c.seed == c.val_data.seed
    assert len(T.split_cfgs([a1, dataclasses.replace(a1, lr=3e-3), dataclasses.replace(a1, lr=4e-3)])) == 1
    c = T.Config()
    assert (c.n_train, c.lr, c.n_lr_warmup, c.grad_clip, c.log_every, c.seed, c.optim) == (100_000_000, 4e-4, 500, 1.0, 25, 42, "adam")


# ---- shard cache + loader ---------------------------------------------------------------------


def test_shard_cache_roundtrip_and_loader_semantics(tmp_path):
    from saev_amd import data
    from saev_amd.data import shards as SH

    rng = np.random.default_rng(0)
    acts = rng.standard_normal((10, 2, 5, 8)).astype(np.float32)  # 10 examples, 2 layers, CLS + 4 patches, d=8
    d = data.write_shards(tmp_path, acts, layers=(6, 11), cls_token=True, max_tokens_per_shard=30)
    md = data.Metadata.load(d)
    assert d.name == md.hash and d.parent.name == "shards" and d.parent.parent.name == "saev"
    assert (md.tokens_per_example, md.examples_per_shard, md.n_shards) == (5, 3, 4)
    info = data.ShardInfo.load(d)
    assert [n for _, n in info] == [3, 3, 3, 1] and info.shards[2][0] == "acts000002.bin"
    mm = SH.open_shard(d, md, *info.shards[1])
    np.testing.assert_array_equal(np.asarray(mm), acts[3:6])

    cfg = data.ShuffledConfig(shards=d, layer=11, tokens="content", batch_size=16, seed=3)
    dl = data.ShuffledDataLoader(cfg, device="cpu")
    assert dl.n_samples == 40 and len(dl) == 3 and dl.batch_size == 16
    seen = {}
    for batch in dl:
        assert batch["act"].dtype == torch.float32 and batch["example_idx"].dtype == torch.int32
        for a, e, t in zip(batch["act"], batch["example_idx"].tolist(), batch["token_idx"].tolist()):
            assert (e, t) not in seen
            seen[(e, t)] = a
            np.testing.assert_array_equal(a.numpy(), acts[e, 1, t + 1])  # content token t sits after the CLS slot
    assert len(seen) == 40, "every row exactly once per epoch"
    first_epoch_order = list(seen)
    assert [k for b in dl for k in zip(b["example_idx"].tolist(), b["token_idx"].tolist())] != first_epoch_order
    # drop_last and data-parallel sharding
    dl2 = data.ShuffledDataLoader(dataclasses.replace(cfg, drop_last=True), device="cpu")
    assert [len(b["act"]) for b in dl2] == [16, 16]
    parts = [data.ShuffledDataLoader(cfg, device="cpu", rank=r, world_size=2) for r in range(2)]
    ex = [set(p.example_idx.tolist()) for p in parts]
    assert ex[0].isdisjoint(ex[1]) and len(ex[0] | ex[1]) == 10 and parts[0].local_batch == 8
    cls = data.ShuffledDataLoader(dataclasses.replace(cfg, tokens="special", layer=6), device="cpu")
    b = next(iter(cls))
    np.testing.assert_array_equal(b["act"][0].numpy(), acts[b["example_idx"][0].item(), 0, 0])
    with pytest.raises(ValueError, match="not in recorded layers"):
        data.ShuffledDataLoader(dataclasses.replace(cfg, layer=3), device="cpu")


@pytest.mark.parametrize("world", [1, 2])
def test_streaming_reservoir_feed_delivers_every_row_once(tmp_path, world):
    """Caches over the device budget stream through the reservoir (resident=False forces that mode): the reference's
    contract is every row exactly once per epoch, random order, ragged last batch unless drop_last."""
    from saev_amd import data

    rng = np.random.default_rng(1)
    acts = rng.standard_normal((37, 2, 5, 8)).astype(np.float32)
    d = data.write_shards(tmp_path, acts, layers=(6, 11), cls_token=True, max_tokens_per_shard=5 * 5 * 2)
    cfg = data.ShuffledConfig(shards=d, layer=11, tokens="content", batch_size=16, seed=3, buffer_size=3, min_buffer_fill=0.5)
    seen_all, steps = set(), []
    for rank in range(world):
        dl = data.ShuffledDataLoader(cfg, device="cpu", rank=rank, world_size=world, resident=False)
        assert dl.reservoir is not None and dl.pool is None and dl.n_samples == 37 * 4
        for epoch in range(2):
            seen, sizes = {}, []
            for batch in dl:
                sizes.append(len(batch["act"]))
                assert batch["act"].dtype == torch.float32 and batch["example_idx"].dtype == torch.int32
                for a, e, t in zip(batch["act"], batch["example_idx"].tolist(), batch["token_idx"].tolist()):
                    assert (e, t) not in seen
                    seen[(e, t)] = True
                    np.testing.assert_array_equal(a.numpy(), acts[e, 1, t + 1])
            # an epoch is the rank's share cut to the smallest share over all ranks (equal step counts and batch sizes on
            # every rank: the per-step collectives of a data-parallel run must stay matched)
            assert len(seen) == dl.n_epoch <= dl.n_local and len(sizes) == len(dl)
            assert all(n == dl.local_batch for n in sizes[:-1]) and sizes[-1] <= dl.local_batch
            steps.append(sizes)
            if epoch == 0:
                order0 = list(seen)
            else:
                assert list(seen) != order0
        seen_all |= set(seen)
    assert all(s == steps[0] for s in steps), "every rank, every epoch: the same batch sizes in the same order"
    if world == 1:
        assert len(seen_all) == 37 * 4
    # same rows, same once-per-epoch contract as the resident mode
    res = data.ShuffledDataLoader(cfg, device="cpu")
    assert res.reservoir is None and res.n_local == 37 * 4
    # drop_last
    dl = data.ShuffledDataLoader(dataclasses.replace(cfg, drop_last=True), device="cpu", resident=False)
    assert [len(b["act"]) for b in dl] == [16] * (37 * 4 // 16)


def test_streaming_reader_errors_reach_the_consumer():
    from saev_amd.data.reservoir import StreamingReservoir

    def blocks():
        yield np.zeros((4, 8), np.float32), np.zeros(4, np.int32), np.zeros(4, np.int32)
        raise OSError("disk went away")

    r = StreamingReservoir(blocks, d_model=8, capacity=64, chunk_rows=8, device="cpu", seed=0)
    r.start_epoch()
    with pytest.raises(RuntimeError, match="reader failed"):
        for _ in range(10):
            r.get(4)
    r.stop()


@pytest.mark.parametrize("resident", [True, False])
def test_ignore_labels_filters_patches(tmp_path, resident):
    from saev_amd import data

    rng = np.random.default_rng(2)
    acts = rng.standard_normal((9, 1, 5, 8)).astype(np.float32)
    labels = rng.integers(0, 3, (9, 4)).astype(np.uint8)
    d = data.write_shards(tmp_path, acts, layers=(11,), cls_token=True, max_tokens_per_shard=4 * 5, labels=labels)
    cfg = data.ShuffledConfig(shards=d, layer=11, batch_size=8, ignore_labels=[0, 2], buffer_size=2)
    dl = data.ShuffledDataLoader(cfg, device="cpu", resident=resident)
    want = {(e, t) for e in range(9) for t in range(4) if labels[e, t] == 1}
    assert dl.n_samples == len(want) and len(dl) == -(-len(want) // 8)
    got = [(e, t) for b in dl for e, t in zip(b["example_idx"].tolist(), b["token_idx"].tolist())]
    assert len(got) == len(want) and set(got) == want
    with pytest.raises(NotImplementedError):
        data.ShuffledDataLoader(dataclasses.replace(cfg, tokens="all"), device="cpu")
    d2 = data.write_shards(tmp_path / "nolabels", acts, layers=(11,), cls_token=True)
    with pytest.raises(FileNotFoundError):
        data.ShuffledDataLoader(dataclasses.replace(cfg, shards=d2), device="cpu")


def test_sample_prefixes_contract():
    from saev_amd.nn import objectives as O

    assert O.sample_prefixes(64, 1).tolist() == [64]
    torch.manual_seed(0)
    p = O.sample_prefixes(512, 10)
    assert p.dtype == torch.int64 and len(p) == 10 and p[-1] == 512 and (p[1:] > p[:-1]).all() and p[0] >= 1
    torch.manual_seed(0)
    assert torch.equal(p, R.sample_prefixes(512, 10)), "same draw as the oracle restatement under the same seed"
    assert O.Matryoshka() == O.Matryoshka(n_prefixes=10, dead_threshold_tokens=10_000_000)


def test_make_saes_datapoint_init_matches_reference(golden):
    """The reference's default (reinit_blend = 0.8) initialises encoder columns from mean-centred activations
    (train.py:108-189).  Same seed + same batches -> same weights as the reference produced (fixture G10)."""
    from saev_amd import data
    from saev_amd.framework import train as T
    from saev_amd.nn import modeling as M, objectives as O

    g = golden("g10_make_saes")
    acts, bsz = g["acts"], int(g["bsz"])

    class Loader:  # the reference's loader contract, in memory and in order
        n_samples = acts.shape[0]

        def __iter__(self):
            for lo in range(0, acts.shape[0], bsz):
                yield {"act": acts[lo : lo + bsz]}

    d, s = acts.shape[1], g["W_dec_0"].shape[0]
    cfgs = [(M.SparseAutoencoderConfig(d_model=d, d_sae=s, reinit_blend=float(b), activation=M.TopK(top_k=4)),
             O.Matryoshka(n_prefixes=1)) for b in g["blends"].tolist()]
    torch.manual_seed(int(g["seed"]))
    saes, objs, groups = T.make_saes(cfgs, Loader(), device="cpu")
    assert len(saes) == len(objs) == 2 and all(grp["lr"] == 0.0 for grp in groups)
    for i, sae in enumerate(saes):
        torch.testing.assert_close(sae.W_enc.detach(), g[f"W_enc_{i}"], rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(sae.W_dec.detach(), g[f"W_dec_{i}"], rtol=1e-6, atol=1e-7)
        assert torch.equal(sae.b_enc.detach(), g[f"b_enc_{i}"])
        torch.testing.assert_close(sae.W_dec.detach().norm(dim=1), torch.ones(s), rtol=1e-5, atol=1e-6)
        assert torch.equal(sae.W_enc.detach(), sae.W_dec.detach().T)
        assert sae.W_enc.data_ptr() != sae.W_dec.data_ptr()
    # too few samples for the dictionary is an error, as in the reference
    small = type("L", (), {"n_samples": s - 1, "__iter__": lambda self: iter(())})()
    with pytest.raises(ValueError, match="needs at least as many activation rows"):
        T.make_saes(cfgs, small, device="cpu")


def test_data_parallel_ranks_take_the_same_steps(tmp_path):
    """with uneven shard shares the ranks of a data-parallel run took different numbers of steps with
    different ragged batches, and the limiter counted local rows against the global n_train (world x too many steps).
    Now: every rank yields the same batch-size sequence, and as many steps as one process with the global batch does
    on the same number of rows per epoch."""
    from saev_amd import data
    from saev_amd.utils import scheduling as S

    rng = np.random.default_rng(2)
    acts = rng.standard_normal((37, 1, 4, 8)).astype(np.float32)
    d = data.write_shards(tmp_path, acts, layers=(11,), cls_token=False, max_tokens_per_shard=4 * 5)
    cfg = data.ShuffledConfig(shards=d, layer=11, batch_size=16, seed=3)
    for resident in (True, False):
        seqs = []
        for rank in range(2):
            dl = data.ShuffledDataLoader(cfg, device="cpu", rank=rank, world_size=2, resident=resident)
            lim = S.BatchLimiter(dl, 400, rows_scale=2)
            seqs.append([len(b["act"]) for b in lim])
            assert len(lim) == 25
            dl.shutdown()
        assert seqs[0] == seqs[1] and len(seqs[0]) >= len(lim)
        n_epoch = data.ShuffledDataLoader(cfg, device="cpu", rank=0, world_size=2).n_epoch

        class One:  # one process, global batch 16, 2 * n_epoch rows per epoch
            batch_size, drop_last = 16, False

            def __iter__(self):
                for lo in range(0, 2 * n_epoch, 16):
                    yield {"act": torch.zeros(min(16, 2 * n_epoch - lo), 1)}

        assert len(seqs[0]) == sum(1 for _ in S.BatchLimiter(One(), 400))


def _tiny_vit_setup(device="cpu", n_img=37):
    from saev_amd import data
    from saev_amd.data.vit import VisionTransformer

    torch.manual_seed(0)
    vit = VisionTransformer(d_model=32, depth=4, heads=4, patch=4, image=16).to(device).eval()  # 16 patches + CLS
    rec = data.ActivationRecorder(vit, vit.blocks, layers=(1, 3), content_tokens_per_example=16, cls_token=True)
    imgs = torch.randn(n_img, 3, 16, 16)

    def images():
        for lo in range(0, n_img, 5):
            yield imgs[lo : lo + 5], torch.arange(lo, min(lo + 5, n_img))

    return data, rec, imgs, images


def test_extraction_feed_delivers_every_token_once_with_the_hooked_values():
    """BASELINE configs[4] hand-off on CPU (the host logic: hooks, token selection, reservoir bookkeeping): every
    (example, token) of the selected layer arrives exactly once per epoch, in a different order each epoch, carrying
    the activation the hooked block produced; `special` / `all` token choices and drop_last as in the shuffled loader."""
    data, rec, imgs, images = _tiny_vit_setup()
    with torch.no_grad():
        full = rec(imgs)[1].clone()  # (37, 2 layers, 17 tokens, 32)
    assert full.shape == (37, 2, 17, 32)
    feed = data.ExtractionFeed(data.ExtractConfig(layer=3, batch_size=64, buffer_size=3, seed=1), rec, images, n_examples=37,
                               d_model=32, device="cpu")
    assert feed.n_samples == 37 * 16 and len(feed) == 10 and feed.metadata.n_examples == 37
    orders = []
    for epoch in range(2):
        seen, sizes = {}, []
        for b in feed:
            sizes.append(len(b["act"]))
            assert b["act"].dtype == torch.float32 and b["example_idx"].dtype == torch.int32 and b["token_idx"].dtype == torch.int32
            assert 0.0 <= feed.reservoir.fill() <= 1.0
            for a, e, t in zip(b["act"], b["example_idx"].tolist(), b["token_idx"].tolist()):
                assert (e, t) not in seen
                seen[(e, t)] = True
                torch.testing.assert_close(a, full[e, 1, t + 1], rtol=1e-5, atol=1e-6)  # content token t sits after CLS
        assert len(seen) == 37 * 16 and sizes == [64] * 9 + [16]
        orders.append(list(seen))
    assert orders[0] != orders[1]
    cls = data.ExtractionFeed(data.ExtractConfig(layer=1, tokens="special", batch_size=8, buffer_size=4), rec, images,
                              n_examples=37, d_model=32, device="cpu")
    got = {e: a for b in cls for a, e in zip(b["act"], b["example_idx"].tolist())}
    assert len(got) == 37
    torch.testing.assert_close(got[5], full[5, 0, 0], rtol=1e-5, atol=1e-6)
    dl = data.ExtractionFeed(data.ExtractConfig(layer=3, tokens="all", batch_size=64, buffer_size=3, drop_last=True), rec, images,
                             n_examples=37, d_model=32, device="cpu")
    assert [len(b["act"]) for b in dl] == [64] * (37 * 17 // 64)
    with pytest.raises(ValueError, match="not in recorded layers"):
        data.ExtractionFeed(data.ExtractConfig(layer=2), rec, images, n_examples=37, d_model=32, device="cpu")


def test_warmup_and_scheduler_base_follow_the_reference():
    """reference utils/scheduling.py:9-39: the Scheduler base raises, Warmup ramps linearly for n_steps calls then stays."""
    from saev_amd.utils import scheduling as S

    w = S.Warmup(0.0, 1.0, 4)
    assert [w.step() for _ in range(6)] == [0.25, 0.5, 0.75, 1.0, 1.0, 1.0]
    assert repr(w) == "Warmup(init=0.0, final=1.0, n_steps=4)"
    assert isinstance(w, S.Scheduler) and isinstance(S.WarmupCosine(0.0, 2, 1.0, 10, 0.0), S.Scheduler)
    with pytest.raises(NotImplementedError):
        S.Scheduler().step()


def test_device_reservoir_rejects_oversized_blocks_with_a_value_error():
    """a block larger than the reservoir (or than its free room) used to hit a bare assert; under -O the negative
    slice start went on to corrupt the slot bookkeeping."""
    from saev_amd.data.extract import DeviceReservoir

    r = DeviceReservoir(8, 4, torch.device("cpu"), seed=0)
    with pytest.raises(ValueError, match="cannot enter a reservoir"):
        r.put(torch.zeros(9, 4), torch.zeros(9, dtype=torch.int32), torch.zeros(9, dtype=torch.int32))
    r.put(torch.zeros(6, 4), torch.zeros(6, dtype=torch.int32), torch.zeros(6, dtype=torch.int32))
    with pytest.raises(ValueError, match="reservoir overflow"):
        r.put(torch.zeros(3, 4), torch.zeros(3, dtype=torch.int32), torch.zeros(3, dtype=torch.int32))
    assert r.room() == 2 and r._n_filled == 6


def test_shard_validation_reports_every_bad_file_up_front(tmp_path):
    """ShardInfo.validate (reference data/shards.py:638-694): missing, empty, non-regular and truncated shard files are all
    named in one FileNotFoundError, and the loaders call it in their constructors' shard scan."""
    import numpy as np

    from saev_amd import data
    from saev_amd.data import shards as shards_lib

    acts = np.random.default_rng(0).standard_normal((40, 1, 4, 16)).astype(np.float32)
    d = data.write_shards(tmp_path / "cache", acts, layers=(3,), max_tokens_per_shard=4 * 8)  # five shards of eight examples
    md, info = shards_lib.Metadata.load(d), shards_lib.ShardInfo.load(d)
    assert len(info) == 5
    info.validate(d, md)  # a healthy cache passes
    names = [n for n, _ in info]
    (d / names[0]).unlink()
    (d / names[1]).write_bytes(b"")
    (d / names[2]).unlink(); (d / names[2]).mkdir()
    (d / names[3]).write_bytes((d / names[3]).read_bytes()[:100])
    with pytest.raises(FileNotFoundError) as exc:
        info.validate(d, md)
    msg = str(exc.value)
    assert "Shard validation failed" in msg
    for title, name in (("Missing files (1)", names[0]), ("Empty files (1)", names[1]), ("Not regular files (1)", names[2]),
                        ("Truncated files (1)", names[3])):
        assert title in msg and name in msg
    assert names[4] not in msg
    # without the metadata only the reference's four categories are checked
    with pytest.raises(FileNotFoundError) as exc:
        info.validate(d)
    assert "Truncated" not in str(exc.value)


def test_batch_entropy_matches_the_reference_on_fixture_g17():
    """The loader-coverage figures of the log block (reference utils/statistics.py:57-122, train.py:369-377)."""
    from saev_amd.utils.statistics import batch_entropy

    g = load_golden("g17_batch_entropy")
    for tag in "abcd":
        n_ex, n_tok = g[f"{tag}_support"].tolist()
        got = batch_entropy(g[f"{tag}_example_idx"], g[f"{tag}_token_idx"], n_ex, n_tok)
        keys = [str(k) for k in g[f"{tag}_keys"]]
        assert sorted(got) == keys
        for k, want in zip(keys, g[f"{tag}_vals"].tolist()):
            assert math.isclose(got[k], want, rel_tol=1e-12, abs_tol=1e-15), (tag, k, got[k], want)
    with pytest.raises(ValueError):
        batch_entropy(torch.zeros(3, dtype=torch.int32), torch.zeros(4, dtype=torch.int32), 5, 5)
    with pytest.raises(ValueError):
        batch_entropy(torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int32), 5, 5)
    with pytest.raises(ValueError):
        batch_entropy(torch.zeros(3, dtype=torch.int32), torch.zeros(3, dtype=torch.int32), 0, 5)


def test_loader_spread_covers_the_global_batch_and_skips_index_free_feeds():
    from saev_amd.framework import train as T

    class MD:
        n_examples, content_tokens_per_example = 10, 4

    class DL:
        metadata = MD()

    b = {"act": None, "example_idx": torch.tensor([0, 1, 1, 9], dtype=torch.int32), "token_idx": torch.tensor([0, 0, 3, 3], dtype=torch.int32)}
    m = T._loader_spread(b, DL())
    assert math.isclose(m["loader/example_coverage"], 0.3) and math.isclose(m["loader/token_coverage"], 0.5)
    assert math.isclose(m["loader/token_entropy"], math.log(2))
    assert T._loader_spread({"act": None}, DL()) == {} and T._loader_spread(b, object()) == {}
