"""TEST INFRASTRUCTURE (like everything under oracle/): record what the device model hands the HIP attack - logits, and the input
gradient either as fp32 (through autograd) or as the int8 signs the stem kernel writes into the attack's gradient-sign sink - so that
the numpy oracle (apgd_oracle.ReplayModel) can be driven by identical numbers and must then reproduce the attack bit for bit.

Why both modes are replayed separately instead of being compared with each other: library backward kernels (MIOpen's convolution
backward-data in the ConvStem, observed on convnext_iso: one gradient element in ~25 % of the runs differs in the last bit) are not
run-to-run reproducible, and one flipped sign of a tiny gradient moves a pixel by a whole step (tools/determinism_check.py)."""
import contextlib

import numpy as np
import torch


def _merge_chunks(items, B):
    """Records of a call that ran as batch chunks (one entry per chunk, in batch order) -> one entry per model call."""
    out, cur = [], []
    for a in items:
        cur.append(a)
        n = sum(c.shape[0] for c in cur)
        assert n <= B, "chunk records do not add up to the batch"
        if n == B:
            out.append(cur[0] if len(cur) == 1 else np.concatenate(cur, 0))
            cur = []
    assert not cur
    return out


def record_attack(R, model, x, y, norm, eps, K, autocast=True, sink=False, splits=1, **kw):
    """Run R.apgd_train on cuda tensors x, y; returns (outputs, logits [K+1, B, classes], grads [K+1, *x.shape]) as numpy.
    sink=True: the product default (int8 signs straight from the stem kernel when the model surface supports it); the recorded
    'gradient' is then the sign tensor itself, which is all the Linf update reads (autopgd_train_clean.py:221).
    splits > 1: the eager form of what a captured two-stream attack replays - the model calls as batch chunks on their own
    streams, every GEMM on cnx_gemm_nt (apgd._apgd_core(splits=, attack_gemm=True)); the chunks' records are joined per call."""
    rec = {"logits": [], "grads": []}
    last = {}

    class Tap(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            if sink:
                last["g"] = g
            else:
                rec["grads"].append(g.detach().float().cpu().numpy())
            return g

    class Rec(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, t):
            out = self.m(Tap.apply(t) if t.requires_grad else t)
            rec["logits"].append(out.detach().float().cpu().numpy())
            return out

    cls = R.ops.grad_sign_sink
    orig_exit = cls.__exit__

    def exit_and_record(self, *exc):
        # (the product may hand the update kernel its signs in a blocked order - include/apgd_hip.h APGD_I8_BLK; the oracle reads
        #  element order)
        src = R.ops.signs_to_linear(self.signs) if self.signs is not None else last.get("g")
        rec["grads"].append(src.detach().float().cpu().numpy().reshape(tuple(self.x_in.shape)))
        rec.setdefault("sink_used", []).append(self.signs is not None)
        return orig_exit(self, *exc)

    saved = R.apgd.USE_SIGN_SINK
    try:
        R.apgd.USE_SIGN_SINK = bool(sink)
        if sink:
            cls.__exit__ = exit_and_record
        ctx = torch.autocast("cuda", dtype=torch.bfloat16) if autocast else contextlib.nullcontext()
        with ctx:
            if splits > 1:
                assert not kw
                out = R.apgd._apgd_core(Rec(model).eval(), x, y, norm, eps, K, 0, splits=splits, attack_gemm=True)
            else:
                out = R.apgd_train(Rec(model).eval(), x, y, norm=norm, eps=eps, n_iter=K, **kw)
        torch.cuda.synchronize()
    finally:
        cls.__exit__ = orig_exit
        R.apgd.USE_SIGN_SINK = saved
    if splits > 1:
        rec["logits"] = _merge_chunks(rec["logits"], x.shape[0])
        rec["grads"] = _merge_chunks(rec["grads"], x.shape[0])
    return out, np.stack(rec["logits"]), np.stack(rec["grads"]), rec.get("sink_used", [])


def check_replay(O, out, logits, grads, x, y, norm, eps, K):
    """The oracle driven by the recorded numbers must give the HIP attack's outputs bit for bit."""
    xb, acc, lb, xba = out
    oxb, oacc, olb, oxba, _ = O.apgd_train_oracle(O.ReplayModel(logits, grads), x.cpu().numpy(), y.cpu().numpy(), norm, eps, K)
    assert np.array_equal(xb.cpu().numpy(), oxb), "x_best differs from the oracle replay"
    assert np.array_equal(xba.cpu().numpy(), oxba), "x_best_adv differs from the oracle replay"
    assert np.array_equal(acc.cpu().numpy(), oacc), "acc differs from the oracle replay"
    np.testing.assert_allclose(lb.cpu().numpy(), olb, rtol=1e-5, atol=3e-7)
