"""The L1 restatement (oracle/apgd_l1_oracle.py) against trajectories recorded from the reference's ``apgd_train(norm='L1')``:
with the recorded logits / gradients replayed, every iterate handed to the model and the four outputs agree to fp32 summation-order
noise (the L1 projection sums and prefix-sums thousands of terms); acc exactly."""
import glob
import os

import numpy as np
import pytest

from oracle import apgd_oracle as O
from oracle import apgd_l1_oracle as L1

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "apgd_l1_*.npz")))


def test_fixtures_exist():
    assert len(CASES) >= 7


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_l1_oracle_follows_the_reference_trajectory(path):
    f = np.load(path)
    K = int(f["n_iter"])
    rep = O.ReplayModel(f["logits"], f["grads"])
    xb, acc, lb, xba, tr = L1.apgd_train_l1_oracle(rep, f["x"], f["y"], float(f["eps"]), K, is_train=bool(f["is_train"]), keep_trace=True)
    fed = np.stack(tr.x_adv_fed)
    assert fed.shape == f["x_adv_fed"].shape
    err = np.abs(fed - f["x_adv_fed"]).reshape(K + 1, -1).max(1)
    assert float(err.max()) <= 2e-6, err
    assert np.array_equal(acc, f["acc"])
    np.testing.assert_allclose(lb, f["loss_best"], rtol=1e-6, atol=1e-7)
    assert float(np.abs(xb - f["x_best"]).max()) <= 2e-6 and float(np.abs(xba - f["x_best_adv"]).max()) <= 2e-6
    # the invariants for images that start inside the box (the reference projects around the UNCLAMPED x, :248-250, so a pixel
    # that starts outside [0, 1] and is never touched stays outside): L1 ball and box
    if f["x"].min() >= 0 and f["x"].max() <= 1:
        d = (xb - f["x"]).reshape(xb.shape[0], -1)
        assert xb.min() >= 0 and xb.max() <= 1 and float(np.abs(d).sum(1).max()) <= float(f["eps"]) * (1 + 1e-5)


def test_l1_projection_lands_in_the_ball_and_the_box():
    g = np.random.default_rng(0)
    x = g.random((6, 3, 10, 10), dtype=np.float32)
    y = (g.standard_normal((6, 3, 10, 10)) * 0.3).astype(np.float32) * (g.random((6, 3, 10, 10)) < 0.2)
    for eps in (0.5, 3.0, 50.0):
        d = L1.l1_projection(x, y.astype(np.float32), eps)
        z = x + y.astype(np.float32) + d
        assert z.min() >= -1e-6 and z.max() <= 1 + 1e-6
        assert float(np.abs(y + d).reshape(6, -1).sum(1).max()) <= eps * (1 + 1e-5) + 1e-5
