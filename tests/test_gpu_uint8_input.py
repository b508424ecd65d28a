"""-m gpu: the on-device uint8 input pipeline vs the reference's ToTensor + the oracle.

utils/datasets.py:204-213 (dSprites: imgs * 255 -> ToTensor) and :282-291 (CelebA: imread -> ToTensor) hand the model
float32 = uint8 / 255.  Here the batch stays uint8 in HBM and the kernels that read the input image divide on the fly."""
from collections import defaultdict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import *  # noqa
from gpu_util import _lib  # noqa
from oracle import disvae_oracle as O
from disvae_amd.models.vae import init_specific_model
from disvae_amd.models.losses import get_loss_f
from disvae_amd.training import Trainer
from disvae_amd.data import DeviceImageLoader

HP = dict(rec_dist="bernoulli", reg_anneal=10000, betaH_B=4, betaB_initC=0, betaB_finC=25,
          betaB_G=1000, factor_G=6.4, latent_dim=10, lr_disc=1e-4, btcvae_A=1, btcvae_B=6.4, btcvae_G=1)


def to_tensor(u8):
    """torchvision.transforms.ToTensor on an already-CHW uint8 batch: .float().div(255)."""
    return u8.to(torch.float32).div(255)


def test_u8_to_f32_is_totensor():
    g = torch.Generator().manual_seed(0)
    for n in (16, 4096 + 16 * 3 + 5, 3 * 64 * 64 * 7):
        u8 = torch.randint(0, 256, (n,), dtype=torch.uint8, generator=g)
        out = torch.empty(n, device=DEV)
        ud = u8.to(DEV)
        call("dvae_u8_to_f32", ptr(ud), ptr(out), n, stream())
        assert torch.equal(out.cpu(), to_tensor(u8))            # all 256 values, bit for bit


@pytest.mark.parametrize("N,C", [(5, 3), (3, 1), (300, 3), (1024, 3)])
def test_u8_kernels_equal_fp32_kernels_bitwise(N, C):
    """conv1 forward, conv1 weight gradient and the fused likelihood with a uint8 image == the same kernels on
    ToTensor(image), bit for bit (the conversion is the only difference and it is exact)."""
    g = torch.Generator().manual_seed(N)
    u8 = torch.randint(0, 256, (N, C, 64, 64), dtype=torch.uint8, generator=g)
    if C == 1:
        u8 = (u8 > 127).to(torch.uint8) * 255                   # dSprites: binary images * 255
    x = to_tensor(u8)
    ud, xd = keep(u8.to(DEV)), dev(x)
    w = dev((torch.rand(32, C, 4, 4, generator=g) - 0.5) * 0.4)
    b = dev((torch.rand(32, generator=g) - 0.5) * 0.2)
    s = stream()
    y8, y32 = torch.empty(N, 32, 32, 32, device=DEV), torch.empty(N, 32, 32, 32, device=DEV)
    call("dvae_conv4s2_fwd_u8", ptr(ud), ptr(w), ptr(b), ptr(y8), N, C, 64, 64, 32, _lib.ACT_RELU, s)
    call("dvae_conv4s2_fwd", ptr(xd), _lib.NCHW, ptr(w), ptr(b), ptr(y32), _lib.NHWC, N, C, 64, 64, 32, _lib.ACT_RELU, s)
    assert torch.equal(y8, y32)
    ref = torch.relu(F.conv2d(x.double(), w.cpu().double(), b.cpu().double(), stride=2, padding=1))
    check(from_nhwc(y8, N, 32, 32, 32), ref, rtol=1e-5, atol_rel=2e-6, what="u8 conv1 fwd vs ToTensor + torch")
    dy = dev(torch.rand(N, 32, 32, 32, generator=g) - 0.5)      # NHWC gradient of conv1's output
    ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=DEV)
    dw8, db8, dw32, db32 = (torch.full(sh, 7.0, device=DEV) for sh in ((32, C, 4, 4), (32,), (32, C, 4, 4), (32,)))
    call("dvae_conv4s2_wgrad_u8", ptr(ud), ptr(dy), ptr(dw8), ptr(db8), N, C, 64, 64, 32, ptr(ws), s)
    call("dvae_conv4s2_wgrad", ptr(xd), _lib.NCHW, ptr(dy), _lib.NHWC, ptr(dw32), ptr(db32), N, C, 64, 64, 32, ptr(ws), s)
    if N < 192:
        assert torch.equal(dw8, dw32) and torch.equal(db8, db32)
    else:
        # large fp32 batches take the wave-specialised kernel (conv_thin_ws.hip: 16x16x4 tiles, another fixed summation order
        # over the 1024 N pixels): the same sums, equal to fp32 rounding of a sum of that length
        refw = torch.autograd.functional.vjp(lambda ww: F.conv2d(x.double(), ww, None, stride=2, padding=1),
                                             w.cpu().double(), from_nhwc(dy, N, 32, 32, 32).double())[1]
        for got in (dw8, dw32):
            check(got, refw, what="conv1 weight gradient")
        check(db32, dy.double().sum((0, 1, 2)), what="conv1 bias gradient")
        check(db8, dy.double().sum((0, 1, 2)), what="conv1 bias gradient (u8)")
    # fused last decoder layer: the target is the input image
    a = dev(torch.relu(torch.rand(N, 32, 32, 32, generator=g) - 0.3))
    wt = dev((torch.rand(32, C, 4, 4, generator=g) - 0.5) * 0.4)
    bt = dev((torch.rand(C, generator=g) - 0.5) * 0.2)
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / N
    coefd = dev(coef)
    outs = []
    for u in (True, False):
        recon, gl = torch.empty(N, C, 64, 64, device=DEV), torch.empty(N, C, 64, 64, device=DEV)
        parts = torch.full((_lib.REC_NPART,), 7.0, device=DEV)
        if u:
            call("dvae_convT4s2_sigmoid_recon_fwd_u8", ptr(a), ptr(wt), ptr(bt), ptr(ud), ptr(recon), ptr(gl), _lib.REC["bernoulli"],
                 ptr(coefd), ptr(parts), N, 32, 32, 32, C, s)
        else:
            call("dvae_convT4s2_sigmoid_recon_fwd", ptr(a), _lib.NHWC, ptr(wt), ptr(bt), ptr(xd), ptr(recon), ptr(gl),
                 _lib.REC["bernoulli"], ptr(coefd), ptr(parts), N, 32, 32, 32, C, s)
        outs.append((recon, gl, parts))
    for t8, t32 in zip(*outs):
        assert torch.equal(t8, t32)


@pytest.mark.parametrize("loss,img,B", [("btcvae", (3, 64, 64), 12), ("btcvae", (1, 64, 64), 33), ("factor", (3, 64, 64), 10),
                                         ("betaH", (1, 32, 32), 8)])
def test_fused_step_on_uint8_batch_vs_totensor_oracle(loss, img, B):
    """One training iteration fed with a uint8 pixel batch == the oracle fed with ToTensor(batch) (the reference's
    data path), and bit-identical to the native step fed with the fp32 image.  32x32 images (MNIST geometry) take the
    one-pass dvae_u8_to_f32 route instead of the fused kernels."""
    seed, n_data, lr = 77, 202599, 5e-4
    gen = torch.Generator().manual_seed(seed + 1)
    u8 = torch.randint(0, 256, (B,) + img, dtype=torch.uint8, generator=gen)
    x = to_tensor(u8)
    hp = dict(HP, n_data=n_data)
    results = []
    for feed in (u8, x):
        torch.manual_seed(seed)
        model = init_specific_model("Burgess", img, 10)
        opt = torch.optim.Adam(model.parameters(), lr=lr)
        loss_f = get_loss_f(loss, device=torch.device(DEV), **hp)
        model.to(DEV).train()
        g2 = torch.Generator().manual_seed(seed + 2)
        storer = defaultdict(list)
        if loss == "factor":
            Bh = B // 2
            eps1, eps2 = torch.randn(Bh, 10, generator=g2), torch.randn(Bh, 10, generator=g2)
            perms = torch.stack([torch.randperm(Bh, generator=g2) for _ in range(10)])
            out = loss_f.call_optimize(feed.to(DEV), model, opt, storer, noise=(dev(eps1), dev(eps2), perms))
        else:
            eps = torch.randn(B, 10, generator=g2)
            out = loss_f.fused_step(feed.to(DEV), model, opt, storer, eps=dev(eps))
        results.append((out.item(), {k: p.grad.clone() for k, p in model.named_parameters()}, dict(storer)))
    assert results[0][0] == results[1][0]
    for k in results[0][1]:
        assert torch.equal(results[0][1][k], results[1][1][k]), k
    # vs the oracle on ToTensor(batch)
    torch.manual_seed(seed)
    p0 = O.init_vae_params(img, 10)
    st = O.LossState(steps_anneal=HP["reg_anneal"])
    if loss == "factor":
        d0 = O.init_disc_params(10)
        ref_loss, ref_logs, _, _, _ = O.factor_iteration_grads(hp, st, O.clone_params(p0, requires_grad=True),
                                                               O.clone_params(d0, requires_grad=True), x, eps1, eps2, list(perms))
    else:
        ref_loss, ref_logs, _, _ = O.train_iteration_grads(loss, hp, st, O.clone_params(p0, requires_grad=True), x, eps)
    np.testing.assert_allclose(results[0][0], ref_loss.item(), rtol=2e-5)
    for k in ref_logs:
        np.testing.assert_allclose(results[0][2][k][0], ref_logs[k].item(), rtol=5e-5, atol=1e-6, err_msg=k)


def test_device_loader_trains_through_the_trainer_api(tmp_path):
    """DeviceImageLoader (uint8 set resident in HBM, batches gathered on the device) through Trainer.__call__
    (training.py:64-102), ragged last batch included; and the reference-style autograd path accepts uint8 too."""
    import logging
    N, img, B = 70, (1, 64, 64), 32
    gen = torch.Generator().manual_seed(5)
    imgs = (torch.rand(N, 64, 64, generator=gen) > 0.8).to(torch.uint8).numpy()      # dSprites-like binary [N,H,W]
    loader = DeviceImageLoader(imgs, batch_size=B, shuffle=True, labels=torch.arange(N), device=DEV)
    assert len(loader) == 3 and len(loader.dataset) == N
    torch.manual_seed(11)
    model = init_specific_model("Burgess", img, 10)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    loss_f = get_loss_f("btcvae", device=torch.device(DEV), n_data=N, **HP)
    tr = Trainer(model, opt, loss_f, device=torch.device(DEV), logger=logging.getLogger("u8"), save_dir=str(tmp_path),
                 is_progress_bar=False)
    tr(loader, epochs=3, checkpoint_every=10)
    assert loss_f.n_train_steps == 9
    assert all(torch.isfinite(p).all() for p in model.parameters())
    seen = sorted(int(v) for _, lab in loader for v in lab)
    assert seen == list(range(N))                                   # every image exactly once per epoch
    batch, _ = next(iter(loader))
    assert batch.dtype == torch.uint8 and batch.is_cuda and int(batch.max()) == 255
    # reference control flow (model(x) -> loss_f(...)) on a uint8 batch
    model.train()
    recon, latent_dist, z = model(batch)
    l = loss_f(batch, recon, latent_dist, True, None, latent_sample=z)
    assert torch.isfinite(l)
