"""The FGSM restatement (oracle/fgsm_oracle.py) against the vectors recorded from the reference's own ``fgsm_train``
(tests/golden/make_fgsm_golden.py): start point and result bit for bit, for every flag combination the trainer can produce."""
import glob
import os

import numpy as np
import pytest

from oracle import apgd_oracle as O
from oracle import fgsm_oracle as FO

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "fgsm_*.npz")))


def test_fixtures_exist():
    assert len(CASES) >= 8


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[5:-4] for p in CASES])
def test_fgsm_oracle_reproduces_the_reference_bit_for_bit(path):
    f = np.load(path)
    rep = O.ReplayModel(f["logits"][None], f["grad"][None])
    out, fed = FO.fgsm_train_oracle(rep, f["x"], f["y"], float(f["eps"]), t=f["t"], alpha=float(f["alpha"]), use_rs=bool(f["use_rs"]),
                                    noise_level=float(f["noise_level"]), skip_projection=bool(f["skip_projection"]))
    assert np.array_equal(fed, f["x_fed"]), "start point differs from what the reference fed to the model"
    assert np.array_equal(out, f["x_adv"])
    if not bool(f["skip_projection"]):
        xc = f["x"]
        assert out.min() >= 0 and out.max() <= 1 and np.all(np.abs(out - xc) <= np.float32(f["eps"]) * (1 + 1e-6) + np.abs(xc - np.clip(xc, 0, 1)))
