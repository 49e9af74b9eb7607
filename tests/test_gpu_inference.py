"""framework/inference.worker_fn on the MI355X against what the reference's own inference pass wrote for the same
cache and checkpoint (fixtures G14, produced by oracle/gen_golden.py running the reference)."""

import json

import numpy as np
import pytest
import scipy.sparse
import torch

import sae_ref as R
from conftest import load_golden
from test_inference_host_cpu import write_cache

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["plain", "labels"])
def test_inference_artifacts_match_the_reference(tmp_path, tag):
    from saev_amd import disk, nn
    from saev_amd.data import Metadata, OrderedConfig
    from saev_amd.framework import inference

    g = load_golden(f"g14_inference_{tag}")
    d = write_cache(tmp_path, g)
    md = Metadata.load(d)
    S, D = g["p_W_dec"].shape
    sae = nn.SparseAutoencoder(nn.SparseAutoencoderConfig(
        d_model=D, d_sae=S, activation=nn.modeling.TopK(top_k=int(g["k"]), aux=nn.modeling.AuxK(k_aux=int(g["k_aux"])))))
    with torch.no_grad():
        for k in R.PARAM_ORDER:
            getattr(sae, k).copy_(g["p_" + k])
    runs_root = tmp_path / "saev" / "runs"
    runs_root.mkdir(parents=True)
    run = disk.Run.new("gpu00014", train_shards_dir=d, val_shards_dir=d, runs_root=runs_root)
    nn.dump(run.ckpt, sae)
    cfg = inference.Config(run=run.run_dir, data=OrderedConfig(shards=d, layer=11, batch_size=int(g["batch_size"])),
                           n_dists=int(g["n_dists"]), ignore_labels=g["ignore_labels"].tolist())
    assert inference.need_compute(cfg)[0]
    inference.worker_fn(cfg)
    out = run.inference / md.hash
    assert sorted(p.name for p in out.iterdir()) == ["config.json", "distributions.pt", "mean_values.pt", "metrics.json",
                                                      "sparsity.pt", "token_acts.npz"]
    assert not inference.need_compute(cfg)[0]

    csr = scipy.sparse.load_npz(out / "token_acts.npz")
    assert csr.shape == tuple(g["csr_shape"].tolist())
    assert csr.indices.dtype == np.int32 and csr.indptr.dtype == np.int32 and csr.data.dtype == np.float32
    np.testing.assert_array_equal(csr.indptr, g["csr_indptr"].numpy())
    np.testing.assert_array_equal(csr.indices, g["csr_indices"].numpy())  # no near-ties in this fixture
    np.testing.assert_allclose(csr.data, g["csr_data"].numpy(), rtol=1e-5, atol=1e-6)

    torch.testing.assert_close(torch.load(out / "mean_values.pt"), g["mean_values"], rtol=1e-5, atol=1e-6, equal_nan=True)
    torch.testing.assert_close(torch.load(out / "sparsity.pt"), g["sparsity"], rtol=1e-6, atol=0)
    torch.testing.assert_close(torch.load(out / "distributions.pt"), g["distributions"], rtol=1e-5, atol=1e-6)
    got = json.loads((out / "metrics.json").read_text())
    want = dict(zip(g["metrics_keys"].tolist(), g["metrics_vals"].tolist()))
    assert list(got) == list(want)
    for k, v in want.items():
        assert got[k] == pytest.approx(v, rel=1e-5), k
    assert isinstance(got["n_tokens"], int) and got["n_tokens"] == int(want["n_tokens"])

    # metrics-only mode writes just metrics.json and honours force_recompute
    (out / "metrics.json").unlink()
    cfg2 = inference.Config(run=run.run_dir, data=cfg.data, save=False, ignore_labels=cfg.ignore_labels)
    assert inference.need_compute(cfg2)[0]
    m = inference.worker_fn(cfg2)
    assert m.normalized_mse == pytest.approx(want["normalized_mse"], rel=1e-5)
    assert inference.need_compute(inference.Config(run=run.run_dir, data=cfg.data, save=False, force_recompute=True))[0]
