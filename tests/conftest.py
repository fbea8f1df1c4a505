import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def golden_cases(prefix="apgd_"):
    return sorted(os.path.basename(p)[len("apgd_"):-4] for p in glob.glob(os.path.join(GOLDEN, f"{prefix}*.npz")))


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, f"apgd_{name}.npz"), allow_pickle=False)
    out = {k: d[k] for k in d.files}
    out["eps"] = float(out["eps"])
    out["n_iter"] = int(out["n_iter"])
    out["norm"] = str(out["norm"])
    out["soft"] = bool(out["soft"])
    out["channels_last"] = bool(out["channels_last"])
    out["loss"] = str(out["loss"]) if "loss" in out else "ce"
    return out


def bits_equal(a, b):
    """Bit-for-bit equality of two fp32 arrays (NaN payloads and signed zeros included)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
