"""disvae_amd -- MI355X-native drop-in for the training path of ``disvae``
(YannDubs/disentangling-vae): same names as disvae/__init__.py:1-3."""
import os as _os

# A training iteration runs on up to five HIP streams (the caller's, the weight-gradient stream, and -- sharded batches -- the
# exchange stream, the communication stream and RCCL's own).  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4); streams that share a queue serialise, which costs a sharded 128-image iteration a third of its time
# (profiles/r05_v14_hw_queues.txt).  The variable is read when HIP initialises (the first device call of the process), so the
# default is raised here, at import; a value already in the environment is respected.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .models.vae import init_specific_model  # noqa: E402
from .training import Trainer  # noqa: E402
from .evaluate import Evaluator  # noqa: E402

__all__ = ["init_specific_model", "Trainer", "Evaluator"]
