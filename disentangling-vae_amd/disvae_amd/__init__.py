"""disvae_amd -- MI355X-native drop-in for the training path of ``disvae``
(YannDubs/disentangling-vae): same names as disvae/__init__.py:1-3."""
from .models.vae import init_specific_model
from .training import Trainer
from .evaluate import Evaluator

__all__ = ["init_specific_model", "Trainer", "Evaluator"]
