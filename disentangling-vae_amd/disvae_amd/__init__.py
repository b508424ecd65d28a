"""disvae_amd -- MI355X-native drop-in for the training path of ``disvae``
(YannDubs/disentangling-vae): same names as disvae/__init__.py:1-3."""
import os as _os

# A training iteration runs on up to five HIP streams (the caller's, the weight-gradient stream, and -- sharded batches -- the
# exchange stream, the communication stream and RCCL's own).  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4); streams that share a queue serialise, which costs a sharded 128-image iteration a third of its time
# (profiles/r05_v14_hw_queues.txt).  The variable is read when HIP initialises (the first device call of the process), so the
# default is raised here, at import; a value already in the environment is respected.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    import sys as _sys
    _t = _sys.modules.get("torch")
    if _t is not None and _t.cuda.is_initialized():
        import warnings as _w
        _w.warn("disvae_amd was imported after HIP was initialised (a torch.cuda call came first): GPU_MAX_HW_QUEUES keeps HIP's "
                "default of 4 hardware queues and the streams of a sharded iteration share queues (about a third slower at 128 "
                "images per GPU).  Import disvae_amd first or export GPU_MAX_HW_QUEUES=8.", RuntimeWarning, stacklevel=2)

# Threading contract (as the reference's: a single Python thread): ONE trainer per device and process.  The side,
# exchange and communication streams are per (device, process) -- created once through dvae_stream_create, shared by every
# engine on that device and never destroyed; two trainers driven from different host threads on the same device would
# interleave their launches on them.

from .models.vae import init_specific_model  # noqa: E402
from .training import Trainer  # noqa: E402
from .evaluate import Evaluator  # noqa: E402

__all__ = ["init_specific_model", "Trainer", "Evaluator"]
