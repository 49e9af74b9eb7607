import pathlib
import sys

import numpy as np
import pytest
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

GOLDEN = ROOT / "tests" / "golden"


ENCODER_MODES = ("f32", "f16x3", "f16r")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "encoder_modes(*modes): the encoder arithmetics a GPU test is collected for (default: all three)")


def pytest_generate_tests(metafunc):
    """GPU tests are collected once per encoder arithmetic (f32 / f16x3 / f16r: all fp32-accurate) unless they name the
    modes that make sense for them with ``@pytest.mark.encoder_modes``; CPU tests are collected once.  Nothing is skipped
    at run time for the mode's sake, so a skip in the GPU log always means the environment."""
    if "encoder_mode" not in metafunc.fixturenames:
        return
    if metafunc.definition.get_closest_marker("gpu") is None:
        return
    mark = metafunc.definition.get_closest_marker("encoder_modes")
    modes = mark.args if mark is not None else ENCODER_MODES
    assert modes and all(m in ENCODER_MODES for m in modes), modes
    metafunc.parametrize("encoder_mode", list(modes), indirect=True)


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


@pytest.fixture(autouse=True)
def encoder_mode(request, monkeypatch):
    """The encoder arithmetic of the engines a GPU test creates (pytest_generate_tests above); CPU tests see "f32" and
    no environment change."""
    mode = getattr(request, "param", None)
    if mode is None:
        yield "f32"
        return
    monkeypatch.setenv("SAEV_AMD_ENCODER", mode)
    yield mode


@pytest.fixture
def dw_rows_route(monkeypatch):
    """Engines created inside the test use the row kernels for the weight gradients (SAEV_AMD_DW=rows), the route every ranged /
    two-pass / gathered backward takes: tests that assert BIT-identity between such a backward and the one-pass backward compare
    like with like (the one-pass default, the column slices, sums dval in another order; tests/test_gpu_dw_slices.py bounds the
    difference)."""
    monkeypatch.setenv("SAEV_AMD_DW", "rows")
