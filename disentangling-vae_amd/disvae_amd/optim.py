"""``optimizer.step()`` of a stock ``torch.optim.Adam`` (main.py:208, losses.py:238) as ONE launch of libdvae_hip.so.

The optimizer stays the caller's torch object: param groups, hyper-parameters (read every step, so a changed
``param_groups[i]["lr"]`` takes effect) and the state tensors ``step`` / ``exp_avg`` / ``exp_avg_sq`` are torch's, in torch's
layout -- ``optimizer.state_dict()`` saves and loads exactly what a stock Adam's does.  Only the arithmetic of ``step()``
moves: ``dvae_adam_step`` (csrc/adam.hip) updates every parameter of the optimizer in one launch.  Through torch the same
step costs the host 100-150 us of Python per iteration (state gathering per parameter, grouping by device and dtype, two
multi-tensor launches; profiles/r05_v10_host_profile.txt) -- at 128 images per GPU that is a third of the iteration.

Taken only when it is exactly the same update: ``type(optimizer) is torch.optim.Adam`` (no subclass), fp32 device
parameters with fp32 gradients, amsgrad / maximize / capturable / differentiable / decoupled_weight_decay off, float
learning rate, no step hooks and no LR scheduler wrapped around ``step`` (those observe calls of ``optimizer.step`` itself).
Anything else -> ``optimizer.step()``.
"""
import ctypes
import weakref

import torch

from . import _lib
from ._lib import call
from ._debug import knob

_RUNNERS = weakref.WeakKeyDictionary()      # optimizer -> _NativeAdam (or False: not eligible)


def step(optimizer):
    """What the native training iteration calls instead of ``optimizer.step()`` (training.py:158, losses.py:307-308)."""
    r = _RUNNERS.get(optimizer)
    if r is None:
        r = _RUNNERS[optimizer] = _NativeAdam(optimizer) if eligible(optimizer) else False
    if r is False or not r.step():
        optimizer.step()


def _global_hooks():
    try:
        from torch.optim import optimizer as _om
        return bool(getattr(_om, "_global_optimizer_pre_hooks", None) or getattr(_om, "_global_optimizer_post_hooks", None))
    except Exception:   # noqa
        return True


def eligible(optimizer):
    if knob("DVAE_NATIVE_ADAM", "1") == "0":          # A/B switch (DVAE_DEBUG=1 only)
        return False
    if type(optimizer) is not torch.optim.Adam:
        return False
    # observers of optimizer.step(): registered hooks, an LR scheduler's call counter patched over the bound method
    if getattr(optimizer, "_optimizer_step_pre_hooks", None) or getattr(optimizer, "_optimizer_step_post_hooks", None):
        return False
    if "step" in vars(optimizer) or _global_hooks():
        return False
    n = 0
    for g in optimizer.param_groups:
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            return False
        if g.get("decoupled_weight_decay"):
            return False
        if isinstance(g.get("lr"), torch.Tensor) or any(isinstance(b_, torch.Tensor) for b_ in g.get("betas", ())):
            return False
        for p in g["params"]:
            if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                return False
            n += 1
    return n > 0


class _NativeAdam:
    def __init__(self, optimizer):
        self.opt = weakref.ref(optimizer)
        self.groups = None        # per param group: [ctypes table, its address, n tensors, host step count]
        self._sig = None
        # a step taken by torch itself (the fallback below, the caller's own optimizer.step() between native iterations) moves
        # the device-side step counts: forget the host copy, the next native step reads them back
        me = weakref.ref(self)

        def on_torch_step(*_a, **_k):
            r = me()
            if r is not None:
                r._sig = None
        self._hook = optimizer.register_step_post_hook(on_torch_step)

    def _observed(self, opt):
        """Somebody else now listens to optimizer.step() (hooks, an LR scheduler): those calls must happen."""
        return (len(opt._optimizer_step_pre_hooks) > 0 or len(opt._optimizer_step_post_hooks) != 1 or "step" in vars(opt)
                or _global_hooks())

    def _signature(self, opt):
        """Cheap per-step check that the tables still describe the optimizer: same parameters, gradients and state tensors
        (load_state_dict, add_param_group, zero_grad(set_to_none=True), a moved model all change one of these)."""
        sig = []
        state = opt.state
        for g in opt.param_groups:
            for p in g["params"]:           # EVERY parameter: a middle one whose .grad was dropped or replaced, or whose state entry
                st = state.get(p)           # was swapped, would otherwise leave a stale device pointer in the table (28 tensors)
                gr = p.grad
                if gr is None or not st:
                    return None
                sig.append((p.data_ptr(), gr.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                            st["step"].data_ptr()))
            sig.append(len(g["params"]))
        return tuple(sig)

    def _build(self, opt):
        groups = []
        for g in opt.param_groups:
            ps = g["params"]
            steps = set()
            tab = (_lib.AdamTensor * len(ps))()
            for e, p in zip(tab, ps):
                if p.grad is None or p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or p.grad.device != p.device:
                    return None
                st = opt.state[p]
                if len(st) == 0:                       # torch's lazy state initialisation of a fused Adam (adam.py:_init_group)
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if "max_exp_avg_sq" in st:
                    return None
                s_ = st["step"]
                if not torch.is_tensor(s_) or s_.device != p.device or s_.dtype != torch.float32:
                    # a foreach / single-tensor Adam keeps `step` on the host: move it where the fused forms keep it
                    s_ = st["step"] = torch.as_tensor(float(s_), dtype=torch.float32, device=p.device)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous() and m.dtype == v.dtype == torch.float32 and m.device == p.device):
                    return None
                e.p, e.g, e.m, e.v, e.step, e.n = p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), s_.data_ptr(), p.numel()
                steps.add(s_)
            # the step count lives on the device (torch's fused convention); read once per rebuild, tracked on the host after
            vals = {float(x) for x in torch.stack(list(steps)).tolist()} if steps else {0.0}
            if len(vals) != 1:
                return None                            # parameters at different step counts: torch's per-tensor bias corrections
            groups.append([tab, ctypes.addressof(tab), len(ps), int(vals.pop())])
        return groups

    def step(self):
        opt = self.opt()
        if opt is None or self._observed(opt):
            return False
        sig = self._signature(opt)
        if sig is None or sig != self._sig:
            self.groups = self._build(opt)
            if self.groups is None:
                self._sig = None
                return False
            self._sig = self._signature(opt)
        stream = torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
        for ent, g in zip(self.groups, opt.param_groups):
            b1, b2 = g["betas"]
            call("dvae_adam_step", ent[1], ent[2], float(ent[3] + 1), float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                 float(g["weight_decay"]), stream)
            ent[3] += 1            # only once the launch is enqueued: a refused call leaves host and device counts equal
        return True
