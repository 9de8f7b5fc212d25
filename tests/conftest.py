import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import mil_amd  # noqa: E402,F401  (registers the package alias)

GOLDEN = os.path.join(REPO, "tests", "golden")
SAMPLE_STRIDE = 97


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check_grad(name, got, gold, tol):
    """Compare a gradient tensor with its golden norm + strided sample (and full tensor if stored)."""
    g = got.detach().cpu().float()
    if name.endswith("k_proj.bias"):
        # softmax is invariant to a constant added to every key's score, so d/d(k_proj.bias) == 0 exactly;
        # both sides only hold rounding noise (1e-7 .. 1e-5 here)
        assert float(g.abs().max()) < 5e-5, name
        return
    if float(gold[name + ".norm"]) < 1e-7:       # mathematically-zero gradient (softmax bias): rounding noise only
        assert float(g.abs().max()) < 1e-6, name
        return
    if name in gold:
        assert rel_err(g, gold[name]) <= tol, (name, rel_err(g, gold[name]))
    samp = g.flatten()[::SAMPLE_STRIDE]
    ref = gold[name + ".sample"]
    scale = float(gold[name + ".norm"]) / max(1.0, float(g.numel()) ** 0.5)
    assert float((samp - ref).abs().max()) <= tol * max(scale * 30, 1e-12) + 1e-9, (
        name, float((samp - ref).abs().max()), scale)
    assert abs(float(g.norm()) - float(gold[name + ".norm"])) <= tol * float(gold[name + ".norm"]) + 1e-9, name
