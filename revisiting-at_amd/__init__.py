"""revisiting-at_amd — MI355X-native APGD adversarial-training inner loop.

Drop-in for the ``adv.attack=apgd`` path of nmndeep/revisiting-at
(``main.py:260-301, 831-844`` + ``autopgd_train_clean.py``).  Import as
``revisiting_at_amd`` (see the shim ``revisiting_at_amd.py`` at the repo root).
"""
from . import _lib
from . import ops, architecture
from .apgd import apgd_train, checkpoint_schedule, criterion_names
from .fgsm import fgsm_train
from . import aa_eval, graphed
from .aa_eval import apgd_attack, run_standard_evaluation, robust_accuracy
from .wrapped_model import WrappedModel
from .config import AdvConfig, build_perturb, wrap_model_for_at
from .architecture import get_new_model, normalize_model
from .train_step import ATTrainStep, create_optimizer, setup_distributed
from . import mixup, checkpoint
from .mixup import Mixup, SoftTargetCrossEntropy

__version__ = "0.1.0"
__all__ = ["apgd_train", "fgsm_train", "checkpoint_schedule", "criterion_names", "WrappedModel", "AdvConfig",
           "build_perturb", "wrap_model_for_at", "get_new_model", "normalize_model", "ATTrainStep", "create_optimizer",
           "setup_distributed", "_lib"]
