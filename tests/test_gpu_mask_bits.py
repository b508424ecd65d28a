"""-m gpu: ReLU masks as bit planes (include/dvae_hip.h, dvae_*_bits) -- the forward kernels of conv1 and convT2 emit one
uint32 per output pixel (bit c = [channel c > 0]) and the input-gradient kernels of conv2 and convT3 read it instead of the
134 MB fp32 activation.  Every entry point must be BIT-IDENTICAL to its fp32-mask twin (same kernels, same accumulator
chains: only where the sign comes from differs), at sizes where every persistent workgroup loops, and the training step
built on them must equal the step on fp32 masks bit for bit.  Reference lines: encoders.py:73-77, decoders.py:77-82 under
training.py:157 (relu backward)."""
from collections import defaultdict

import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import *  # noqa
from gpu_util import _lib  # noqa
from test_gpu_fused_core import _rand, _stage


def _pack_bits(act_nhwc):
    """[P, 32] fp32 (device) -> int32 [P]: bit c = act[p][c] > 0."""
    w = (act_nhwc.reshape(-1, 32) > 0).to(torch.int64) << torch.arange(32, device=act_nhwc.device, dtype=torch.int64)
    v = w.sum(1)
    return torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32)


@pytest.mark.parametrize("N,C,u8", [(3, 3, False), (520, 3, False), (70, 1, False), (520, 3, True), (9, 1, True)])
def test_conv1_forward_emits_the_bit_plane(N, C, u8):
    g = torch.Generator().manual_seed(N + C)
    if u8:
        x = torch.randint(0, 256, (N, C, 64, 64), generator=g, dtype=torch.uint8)
        xd = keep(x.to(DEV))
    else:
        x = torch.rand(N, C, 64, 64, generator=g)
        xd = dev(x)
    w, b = dev(_rand(32, C, 4, 4, seed=1, scale=0.3)), dev(_rand(32, seed=2, scale=0.2))
    y_ref = torch.empty(N, 32, 32, 32, device=DEV)
    if u8:
        call("dvae_conv4s2_fwd_u8", ptr(xd), ptr(w), ptr(b), ptr(y_ref), N, C, 64, 64, 32, _lib.ACT_RELU, stream())
    else:
        call("dvae_conv4s2_fwd", ptr(xd), _lib.NCHW, ptr(w), ptr(b), ptr(y_ref), _lib.NHWC, N, C, 64, 64, 32, _lib.ACT_RELU, stream())
    y = torch.full((N, 32, 32, 32), 7.0, device=DEV)
    bits = torch.full((N * 1024,), 0x55555555, dtype=torch.int32, device=DEV)
    call("dvae_conv1_fwd_bits", ptr(xd), int(u8), ptr(w), ptr(b), ptr(y), ptr(bits), N, C, stream())
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    assert torch.equal(bits, _pack_bits(y_ref))
    assert 0.1 < (y_ref > 0).float().mean().item() < 0.9        # the mask is not trivial


@pytest.mark.parametrize("N", [2, 300, 1030])
def test_conv32_up_bits_forward_and_input_gradient(N):
    w = dev(_rand(32, 32, 4, 4, seed=3, scale=0.1))
    img_d, img_u = torch.empty(16384, device=DEV), torch.empty(16384, device=DEV)
    _stage([(w, img_d, img_u)])
    small = dev(_rand(N, 16, 16, 32, seed=4))
    bias = dev(_rand(32, seed=5, scale=0.1))
    # forward form (convT2): output + its bit plane
    ref = torch.empty(N, 32, 32, 32, device=DEV)
    call("dvae_conv32_up", ptr(small), _lib.NHWC, ptr(img_u), ptr(bias), None, ptr(ref), N, 16, _lib.ACT_RELU, stream())
    out = torch.full((N, 32, 32, 32), 7.0, device=DEV)
    bits = torch.full((N * 1024,), 0x33333333, dtype=torch.int32, device=DEV)
    call("dvae_conv32_up_bits", ptr(small), ptr(img_u), ptr(bias), None, ptr(out), ptr(bits), N, _lib.ACT_RELU, stream())
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert torch.equal(bits, _pack_bits(ref))
    # input-gradient form (conv2's): masked by the bit plane of an activation == masked by the activation itself
    act = dev(_rand(N, 32, 32, 32, seed=6))                 # ~half of the units off
    abits = _pack_bits(act)
    gref = torch.empty(N, 32, 32, 32, device=DEV)
    call("dvae_conv32_up", ptr(small), _lib.NHWC, ptr(img_u), None, ptr(act), ptr(gref), N, 16, _lib.ACT_NONE, stream())
    gout = torch.full((N, 32, 32, 32), 7.0, device=DEV)
    call("dvae_conv32_up_bits", ptr(small), ptr(img_u), None, ptr(abits), ptr(gout), None, N, _lib.ACT_NONE, stream())
    torch.cuda.synchronize()
    assert torch.equal(gout, gref)
    assert (gref == 0).float().mean().item() > 0.3


@pytest.mark.parametrize("N,C", [(2, 3), (300, 3), (1030, 3), (130, 1)])
def test_convT3_input_gradient_from_the_bit_plane(N, C):
    w = dev(_rand(32, C, 4, 4, seed=7, scale=0.2))
    dy = dev(_rand(N, C, 64, 64, seed=8))
    act = dev(_rand(N, 32, 32, 32, seed=9))
    abits = _pack_bits(act)
    ref = torch.empty(N, 32, 32, 32, device=DEV)
    call("dvae_convT4s2_dgrad", ptr(dy), _lib.NCHW, ptr(w), ptr(act), ptr(ref), _lib.NHWC, N, 32, 32, 32, C, stream())
    out = torch.full((N, 32, 32, 32), 7.0, device=DEV)
    call("dvae_convT3_dgrad_bits", ptr(dy), ptr(w), ptr(abits), ptr(out), N, C, stream())
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert (ref == 0).float().mean().item() > 0.3


@pytest.mark.parametrize("loss,B,C", [("btcvae", 96, 3), ("factor", 64, 1), ("btcvae", 520, 3)])
def test_training_step_on_bit_planes_equals_the_step_on_fp32_masks(loss, B, C):
    """Two models from the same seed, one with engine.mask_bits switched off (fp32 masks): after two iterations on the same
    batch and noise every parameter and the loss are bit-identical."""
    from disvae_amd.models.vae import init_specific_model
    from disvae_amd.models.losses import get_loss_f
    hp = dict(rec_dist="bernoulli", reg_anneal=10000, betaH_B=4, betaB_initC=0, betaB_finC=25, betaB_G=1000, factor_G=6.4,
              latent_dim=10, lr_disc=1e-4, btcvae_A=1, btcvae_B=6.4, btcvae_G=1, n_data=737280)
    g = torch.Generator().manual_seed(11)
    data = torch.rand(B, C, 64, 64, generator=g).to(DEV)
    Bh = B // 2
    noise = {"btcvae": [torch.randn(B, 10, generator=g).to(DEV) for _ in range(2)],
             "factor": [(torch.randn(Bh, 10, generator=g).to(DEV), torch.randn(Bh, 10, generator=g).to(DEV),
                         torch.stack([torch.randperm(Bh, generator=g) for _ in range(10)])) for _ in range(2)]}[loss]
    results = []
    for bits in (True, False):
        torch.manual_seed(5)
        model = init_specific_model("Burgess", (C, 64, 64), 10).to(DEV).train()
        assert model.engine.mask_bits
        model.engine.mask_bits = bits
        opt = torch.optim.Adam(model.parameters(), lr=5e-4)
        torch.manual_seed(6)
        loss_f = get_loss_f(loss, device=torch.device(DEV), **hp)
        losses = []
        for it in range(2):
            if loss == "factor":
                out = loss_f.call_optimize(data, model, opt, defaultdict(list), noise=noise[it])
            else:
                out = loss_f.fused_step(data, model, opt, defaultdict(list), eps=noise[it])
            losses.append(out.clone())
        torch.cuda.synchronize()
        results.append((torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    (l1, p1), (l0, p0) = results
    assert torch.equal(l1, l0), (l1, l0)
    for k in p1:
        assert torch.equal(p1[k], p0[k]), k
