import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--slow", action="store_true", default=False,
                     help="also run the tests marked `slow` (wide seed sweeps, the full fuzz / perturbation / depth-front matrices); "
                          "the builder runs `pytest tests -m gpu --slow` and keeps the log under profiles/")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: the wide form of a test whose representative cases run without it (run with --slow)")


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
    """`slow` items leave the collection unless --slow is given: `-m gpu` (the driver's command) then runs one representative of
    every matrix and stays inside its time limit."""
    if config.getoption("--slow"):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("slow") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def slow(request):
    """True under --slow: tests that loop over seeds widen their lists."""
    return bool(request.config.getoption("--slow"))


def load_golden(name):
    """Returns a dict; keys 'group::name' are regrouped into nested dicts of tensors."""
    raw = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    out = {}
    for k in raw.files:
        v = raw[k]
        val = torch.from_numpy(v) if v.dtype.kind == "f" else v
        if "::" in k:
            grp, sub = k.split("::", 1)
            out.setdefault(grp, {})[sub] = val
        else:
            out[k] = val
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny) over the tensor: the 1e-4 parity gate's metric."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture
def kenv(monkeypatch):
    """KBN_* switches for A/B tests: the library reads the environment once at load time, so every change
    is followed by kbn_reload_env(); everything is restored (and re-read) when the test ends."""
    import kbnet_amd as kb

    class Env:
        def setenv(self, name, value):
            monkeypatch.setenv(name, value)
            kb.ops.reload_env()

        def delenv(self, name):
            monkeypatch.delenv(name, raising=False)
            kb.ops.reload_env()

    yield Env()
    monkeypatch.undo()
    kb.ops.reload_env()
