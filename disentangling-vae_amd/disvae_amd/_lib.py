"""ctypes binding of libdvae_hip.so (the C-ABI declared in include/dvae_hip.h).

The library is the product: there is NO CPU or PyTorch fallback.  If the shared object is
missing or a symbol cannot be resolved, importing an op fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVAE_HIP_LIB", os.path.join(_HERE, "..", "lib", "libdvae_hip.so"))

NCHW, NHWC = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY02, ACT_SIGMOID = 0, 1, 2, 3
REC = {"bernoulli": 0, "gaussian": 1, "laplace": 2}
LOSS_BETAH, LOSS_BETAB, LOSS_BTCVAE, LOSS_FACTOR = 0, 1, 2, 3
# scalar slots (dvae_hip.h)
S_LOSS, S_REC, S_KL, S_KL0, S_MI, S_TC, S_DWKL, S_KLW, S_DTC, NSCAL = 0, 1, 2, 3, 19, 20, 21, 22, 23, 32
C_INV_B, C_ANNEAL, C_BETA, C_ALPHA, C_GAMMA, C_CAP, NCOEF = 0, 1, 2, 3, 4, 5, 8
REC_NPART = 2048
NPACK = 32
MAX_LATENT_DIM = 16          # DVAE_MAX_D
BTCVAE_MAX_LATENT_DIM = 16   # DVAE_BTCVAE_MAX_D
ROWSTATS = 32                # DVAE_ROWSTATS

_p = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_long

# name -> argtypes (all return int unless listed in _RESTYPE)
SIGNATURES = {
    "dvae_version": [],
    "dvae_last_error": [],
    "dvae_conv4s2_fwd": [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "dvae_conv4s2_dgrad": [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "dvae_conv4s2_wgrad": [_p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p],
    "dvae_convT4s2_fwd": [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "dvae_convT4s2_dgrad": [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "dvae_convT4s2_wgrad": [_p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p],
    "dvae_convT4s2_sigmoid_recon_fwd": [_p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    "dvae_conv_wgrad_ws_floats": [],
    "dvae_stage_weights": [_p, _i, _p, _i, _p, _p, _p, _p],
    "dvae_convT3_fwd_staged": [_p, _p, _p, _p, _i, _p, _p, _i, _p, _p, _i, _i, _p],
    "dvae_conv32_down": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "dvae_conv32_up": [_p, _i, _p, _p, _p, _p, _i, _i, _i, _p],
    "dvae_conv1_fwd_bits": [_p, _i, _p, _p, _p, _p, _i, _i, _p],
    "dvae_conv32_up_bits": [_p, _p, _p, _p, _p, _p, _i, _i, _p],
    "dvae_convT3_dgrad_bits": [_p, _p, _p, _p, _i, _i, _p],
    "dvae_fc_chain_fwd": [_p, _p],
    "dvae_fc_chain_bwd": [_p, _p],
    "dvae_reparam_kl_blocks": [_i],
    "dvae_kl_finish": [_p, _i, _p, _i, _p],
    "dvae_u8_to_f32": [_p, _p, _l, _p],
    "dvae_u8_fused_supported": [_i, _i, _i],
    "dvae_conv4s2_fwd_u8": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "dvae_conv4s2_wgrad_u8": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p],
    "dvae_convT4s2_sigmoid_recon_fwd_u8": [_p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    "dvae_relayout": [_p, _i, _p, _i, _i, _i, _i, _p],
    "dvae_linear_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    "dvae_linear_dgrad": [_p, _p, _p, _i, _p, _i, _i, _i, _p, _p],
    "dvae_linear_wgrad": [_p, _p, _p, _p, _i, _i, _i, _p, _p],
    "dvae_linear_wgrad_grouped": [_p, _i, _p],
    "dvae_reparam_kl_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    "dvae_reparam_kl_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    "dvae_recon_loss": [_p, _p, _l, _i, _p, _p, _p, _i, _p],
    "dvae_kl_normal_bwd": [_p, _p, _p, _p, _p, _i, _i, _p],
    "dvae_reduce_sum": [_p, _l, ctypes.c_float, _p, _p],
    "dvae_sigmoid_bwd": [_p, _p, _p, _l, _p],
    "dvae_btcvae_fwd": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    "dvae_btcvae_bwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p],
    "dvae_permute_dims": [_p, _p, _p, _i, _i, _p],
    "dvae_latent_entropy_ws_floats": [_l, _i, _i],
    "dvae_latent_entropy": [_p, _p, _p, _l, _i, _i, _p, _p, _p],
    "dvae_disc_losses": [_p, _i, _p, _p, _p, _p, _p],
    "dvae_loss_pack": [_p, _p, _i, _p, _i, _p, _p, _p],
    "dvae_loss_finalize": [_i, _p, _i, _i, _p, _p, _p],
    "dvae_loss_epilogue": [_i, _p, _p, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p],
    "dvae_set_coef": [_p] + [ctypes.c_float] * 8 + [_p],
    "dvae_comm_load": [ctypes.c_char_p],
    "dvae_comm_unique_id": [_p],
    "dvae_comm_init": [_p, _p, _i, _i],
    "dvae_comm_destroy": [_p],
    "dvae_comm_world": [_p],
    "dvae_comm_rank": [_p],
    "dvae_comm_allreduce": [_p, _p, _l, _p],
    "dvae_comm_allgather": [_p, _p, _p, _l, _p],
    "dvae_comm_reducescatter": [_p, _p, _p, _l, _p],
    "dvae_comm_broadcast": [_p, _p, _l, _i, _p],
    "dvae_comm_group_start": [],
    "dvae_comm_group_end": [],
    "dvae_add": [_p, _p, _p, _l, _p],
    "dvae_stream_order": [_p, _p],
}
_RESTYPE = {"dvae_last_error": ctypes.c_char_p, "dvae_conv_wgrad_ws_floats": ctypes.c_size_t,
            "dvae_latent_entropy_ws_floats": ctypes.c_size_t}


FCW_MAX = 8
KL_MAX_BLOCKS = 1024                 # DVAE_KL_MAX_BLOCKS
KL_FLOATS = 16 + KL_MAX_BLOCKS * 16   # DVAE_KL_FLOATS
FC_CHAIN_ROWS = 8                    # batch rows per workgroup of dvae_fc_chain_*: ceil(n / 8) KL partial blocks


class ConvImageDesc(ctypes.Structure):
    """dvae_conv_image_desc (include/dvae_hip.h)."""
    _fields_ = [("w", _p), ("img_down", _p), ("img_up", _p)]


class FcImageDesc(ctypes.Structure):
    """dvae_fc_image_desc (include/dvae_hip.h)."""
    _fields_ = [("w", _p), ("img_fwd", _p), ("img_bwd", _p), ("N", _i), ("K", _i)]


class ThinImageDesc(ctypes.Structure):
    """dvae_thin_image_desc (include/dvae_hip.h)."""
    _fields_ = [("w", _p), ("img_pairs", _p), ("C", _i)]


def thin_pair_floats(C):
    """DVAE_THIN_PAIR_FLOATS(C)."""
    return 48 if C == 3 else 16


class FcChainFwdArgs(ctypes.Structure):
    """dvae_fc_chain_fwd_args (include/dvae_hip.h)."""
    _fields_ = [(n_, _p) for n_ in ("a_flat", "w_e1", "w_e2", "w_ml", "w_d1", "w_d2", "w_d3", "b_e1", "b_e2", "b_ml",
                                    "b_d1", "b_d2", "b_d3", "eps", "h1", "h2", "ml", "mu", "logvar", "z", "kl_part",
                                    "d1", "d2", "d3")] + [(n_, _i) for n_ in ("n_enc", "n_kl", "n_dec", "D")]


class FcChainBwdArgs(ctypes.Structure):
    """dvae_fc_chain_bwd_args (include/dvae_hip.h)."""
    _fields_ = [(n_, _p) for n_ in ("gd3", "w_d3", "w_d2", "w_d1", "w_ml", "w_e2", "w_e1", "d2", "d1", "h2", "h1",
                                    "a_flat", "mu", "logvar", "eps", "dz2", "dz3", "dmu_x", "dlv_x", "scal", "coef",
                                    "gd2", "gd1", "dz", "dml", "gh2", "gh1", "ga_flat")] + [("n", _i), ("D", _i)]


def struct_of(cls, **kw):
    """ctypes struct with the given fields (device pointers as ints / None, sizes as ints) -> (struct, its address).
    Keep the struct alive while a recorded launch plan may replay the call."""
    st = cls()
    for k, v in kw.items():
        setattr(st, k, v)
    return st, ctypes.addressof(st)


class LinearWgradDesc(ctypes.Structure):
    """dvae_linear_wgrad_desc (include/dvae_hip.h)."""
    _fields_ = [("x", _p), ("dy", _p), ("dw", _p), ("db", _p), ("M", _i), ("K", _i), ("N", _i)]


def wgrad_descs(problems):
    """[(x, dy, dw, db, M, K, N), ...] (device pointers as ints) -> (host array, its address).  The array must stay
    alive as long as a recorded launch plan may replay the call: callers keep it."""
    arr = (LinearWgradDesc * len(problems))()
    for d, (x, dy, dw, db, M, K, N) in zip(arr, problems):
        d.x, d.dy, d.dw, d.db, d.M, d.K, d.N = x, dy, dw, db, M, K, N
    return arr, ctypes.addressof(arr)


class DvaeHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is None:
        path = os.path.abspath(LIB_PATH)
        if not os.path.exists(path):
            raise DvaeHipError(
                "libdvae_hip.so not found at %s -- build it with `python disentangling-vae_amd/build.py` "
                "(there is no CPU / PyTorch fallback for the training-step kernels)" % path)
        h = ctypes.CDLL(path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the symbol is missing: fail loudly
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, ctypes.c_int)
        _lib = h
    return _lib


ALLOC_GEN = [0]   # bumped whenever the engine (re)allocates device buffers: recorded plans hold raw pointers


def note_alloc():
    ALLOC_GEN[0] += 1


_REC = None   # launch list being recorded (graph.py: "plan" replay), or None


def call(name, *args):
    """Call an int-returning entry point, raise on a non-zero status."""
    h = lib()
    fn = getattr(h, name)
    if _REC is not None:
        # frozen copy of the launch: arguments pre-converted to their ctypes so that a replay is
        # one foreign call per launch with no Python-side marshalling
        _REC.append((fn, tuple(None if a is None else t(a) for t, a in zip(fn.argtypes, args)), name))
    rc = fn(*args)
    if rc != 0:
        raise DvaeHipError("%s failed (%d): %s" % (name, rc, h.dvae_last_error().decode()))


def record_py(fn, *args):
    """Run a host-side callable that belongs to the launch sequence (stream fork/join, a torch
    copy) and keep it in the recorded plan."""
    if _REC is not None:
        _REC.append((fn, args, None))
    return fn(*args)


def begin_record():
    global _REC
    _REC = []


def end_record():
    global _REC
    plan, _REC = _REC, None
    return plan


def replay(plan):
    for fn, args, name in plan:
        rc = fn(*args)
        if name is not None and rc != 0:
            raise DvaeHipError("%s failed (%d): %s" % (name, rc, lib().dvae_last_error().decode()))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
