import pathlib
import sys

import numpy as np
import pytest
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name: str) -> dict:
    with np.load(GOLDEN / f"{name}.npz") as z:
        out = {}
        for k in z.files:
            a = z[k]
            if a.dtype.kind in "fiub":
                out[k] = torch.from_numpy(a.copy()) if a.ndim > 0 else a.item()
            else:
                out[k] = a
        return out


@pytest.fixture
def golden():
    return load_golden


@pytest.fixture(params=["f32", "f16x3", "f16r"], autouse=True)
def encoder_mode(request, monkeypatch):
    """GPU tests run once per encoder arithmetic (both are fp32-accurate); CPU tests ignore it."""
    if "gpu" not in request.keywords:
        if request.param != "f32":
            pytest.skip("encoder mode only matters on the GPU")
        yield request.param
        return
    monkeypatch.setenv("SAEV_AMD_ENCODER", request.param)
    yield request.param


@pytest.fixture
def dw_rows_route(monkeypatch):
    """Engines created inside the test use the row kernels for the weight gradients (SAEV_AMD_DW=rows), the route every ranged /
    two-pass / gathered backward takes: tests that assert BIT-identity between such a backward and the one-pass backward compare
    like with like (the one-pass default, the column slices, sums dval in another order; tests/test_gpu_dw_slices.py bounds the
    difference)."""
    monkeypatch.setenv("SAEV_AMD_DW", "rows")
