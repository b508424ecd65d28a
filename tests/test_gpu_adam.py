"""-m gpu: ``disvae_amd.optim.step`` -- the update of a stock torch.optim.Adam as one launch of libdvae_hip.so
(dvae_adam_step, csrc/adam.hip) -- against torch's own Adam (main.py:208, losses.py:238 build the optimizers;
training.py:158, losses.py:307-308 step them):
  * element-wise equal to torch's CPU Adam fed with the same gradients to <= 1 ulp of the parameter (the same bar as
    tests/test_gpu_timed_config.py holds torch's fused GPU Adam to), over several steps, tensor shapes, beta pairs and weight decay;
  * the optimizer object stays a stock one: state_dict() loads into a CPU torch.optim.Adam and back, steps taken by torch
    itself in between (optimizer.step() called by the user) keep the bias corrections right, changed learning rates count;
  * everything that is not exactly that update falls back to optimizer.step()."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import *  # noqa
from disvae_amd import optim


def _ulp(a, b):
    return ((a - b).abs() / (b.abs() * 2.0 ** -23 + 1e-9)).max().item()


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(sum(torch.Size(s).numel() for s in shapes) + 3, generator=g)
    out, off = [], 1                       # views at a 4-byte offset: the scalar path of the kernel
    for s in shapes:
        n = torch.Size(s).numel()
        out.append(base[off:off + n].view(s))
        off += n
    return out


SHAPES = [(32, 3, 4, 4), (32,), (256, 512), (1,), (20, 256), (4099,), (1000, 1000)]


@pytest.mark.parametrize("betas,wd,lr", [((0.9, 0.999), 0.0, 5e-4), ((0.5, 0.9), 0.0, 1e-5), ((0.9, 0.999), 0.01, 1e-3)])
def test_native_adam_equals_torch_cpu_adam(betas, wd, lr):
    cpu = [torch.nn.Parameter(p.clone()) for p in _params(1, SHAPES)]
    flat = torch.cat([p.detach().reshape(-1) for p in cpu]).to(DEV)
    pad = torch.zeros(1, device=DEV)        # unaligned arena: parameter views start 4 bytes into the buffer
    arena = torch.cat((pad, flat))
    gpu, off = [], 1
    for p in cpu:
        gpu.append(torch.nn.Parameter(arena[off:off + p.numel()].view(p.shape)))
        off += p.numel()
    oc = torch.optim.Adam(cpu, lr=lr, betas=betas, weight_decay=wd)
    og = torch.optim.Adam(gpu, lr=lr, betas=betas, weight_decay=wd)
    assert optim.eligible(og)
    gen = torch.Generator().manual_seed(7)
    for step in range(6):
        if step == 3:                       # a changed learning rate is read at the next step
            for o in (oc, og):
                o.param_groups[0]["lr"] = lr * 0.5
        for pc, pg in zip(cpu, gpu):
            g = torch.randn(pc.shape, generator=gen) * (10.0 ** ((step % 3) - 1))
            pc.grad, pg.grad = g.clone(), g.to(DEV)
        oc.step()
        if step == 4:
            og.step()                       # torch itself steps once in between: the host step count must follow
        else:
            optim.step(og)
        assert optim._RUNNERS[og] is not False
        for k, (pc, pg) in enumerate(zip(cpu, gpu)):
            d = _ulp(pg.detach().cpu(), pc.detach())
            assert d <= 1.0, "step %d tensor %d: %.2f ulp" % (step, k, d)
            sc, sg = oc.state[pc], og.state[pg]
            assert float(sg["step"]) == float(sc["step"]) == step + 1
            check(sg["exp_avg"], sc["exp_avg"], rtol=1e-6, atol_rel=1e-7, what="exp_avg %d" % k)
            check(sg["exp_avg_sq"], sc["exp_avg_sq"], rtol=1e-6, atol_rel=1e-7, what="exp_avg_sq %d" % k)
        with torch.no_grad():               # re-synchronise: the comparison stays a one-step one
            for pc, pg in zip(cpu, gpu):
                pg.copy_(pc.detach().to(DEV))
    # the optimizer is still a stock one: its state loads into a CPU Adam over clones of the parameters, and back
    sd = copy.deepcopy(og.state_dict())
    clones = [torch.nn.Parameter(p.detach().cpu().clone()) for p in gpu]
    o2 = torch.optim.Adam(clones, lr=lr, betas=betas, weight_decay=wd)
    o2.load_state_dict(sd)
    assert float(o2.state[clones[0]]["step"]) == 6
    og.load_state_dict(sd)                  # new state tensors: the runner rebuilds its table
    for pg in gpu:
        pg.grad = torch.ones_like(pg)
    before = [p.detach().clone() for p in gpu]
    optim.step(og)
    assert float(og.state[gpu[0]]["step"]) == 7
    assert all(not torch.equal(a, b.detach()) for a, b in zip(before, gpu))


def test_native_adam_falls_back_when_the_update_is_not_the_stock_one():
    mk = lambda: [torch.nn.Parameter(torch.randn(64, device=DEV))]
    assert not optim.eligible(torch.optim.Adam(mk(), amsgrad=True))
    assert not optim.eligible(torch.optim.Adam(mk(), maximize=True))
    assert not optim.eligible(torch.optim.AdamW(mk()))
    assert not optim.eligible(torch.optim.SGD(mk(), lr=0.1))
    assert not optim.eligible(torch.optim.Adam([torch.nn.Parameter(torch.randn(8))]))            # CPU parameters
    assert not optim.eligible(torch.optim.Adam(mk(), lr=torch.tensor(1e-3), foreach=False))
    o = torch.optim.Adam(mk())
    torch.optim.lr_scheduler.StepLR(o, 10)                                                     # patches optimizer.step
    assert not optim.eligible(o)
    o = torch.optim.Adam(mk())
    calls = []
    o.register_step_post_hook(lambda *a, **k: calls.append(1))
    assert not optim.eligible(o)
    p = mk()
    o = torch.optim.Adam(p, amsgrad=True)
    p[0].grad = torch.ones_like(p[0])
    optim.step(o)                                                                              # runs torch's own step
    assert "max_exp_avg_sq" in o.state[p[0]] and optim._RUNNERS[o] is False
    # hooks registered AFTER the native path was taken are honoured from then on
    p = mk()
    o = torch.optim.Adam(p)
    p[0].grad = torch.ones_like(p[0])
    optim.step(o)
    o.register_step_post_hook(lambda *a, **k: calls.append(2))
    optim.step(o)
    assert calls == [2] and float(o.state[p[0]]["step"]) == 2
