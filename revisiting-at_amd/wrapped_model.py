"""``WrappedModel`` — the attack boundary of the reference (``/root/reference/main.py:260-301``).

Same constructor, ``forward(x, y=None)``, ``set_perturb(mode)`` and the same ``base_model.``
state-dict prefix, so checkpoints and the trainer's call ``self.model(images, target)``
(``main.py:985-989``) work unchanged.  ``perturb`` is any callable
``perturb(model, x, y) -> tensor | tuple``; the MI355X path injects
``functools.partial(apgd_train, ...)`` from ``revisiting_at_amd.config.build_perturb``.
"""
import torch.nn as nn


class WrappedModel(nn.Module):
    """Generates the adversarial perturbation inside the forward pass."""

    def __init__(self, base_model, perturb, verbose=False):
        # `verbose` is kept for signature compatibility; the reference's un-synchronised wall-clock prints
        # (main.py:280-287) are diagnostics, not interface, and are not reproduced
        super().__init__()
        self.base_model = base_model
        self.perturb = perturb
        self.perturb_input = False
        self.verbose = verbose

    def forward(self, x, y=None):
        if self.perturb_input:
            assert y is not None                               # main.py:276
            self.base_model.eval()                             # attack runs in eval mode (main.py:279)
            z = self.perturb(self.base_model, x, y)            # main.py:283
            self.base_model.train()                            # main.py:289
            if isinstance(z, (tuple, list)):
                z = z[0]                                       # x_best (main.py:291-292)
            return self.base_model(z)
        return self.base_model(x)

    def set_perturb(self, mode):
        self.perturb_input = mode
