"""On-device input pipeline: the whole image set resident in HBM as uint8, batches gathered on the
device, ToTensor's ``/ 255`` fused into the kernels that read the input image.

What it replaces: ``utils/datasets.py`` ``DSprites.__getitem__`` (:194-213: ``imgs[idx] * 255`` -> ``ToTensor``),
``CelebA.__getitem__`` (:273-291: ``imread`` -> ``ToTensor``) and ``get_dataloaders`` (:46-71:
``DataLoader(dataset, batch_size, shuffle=True, pin_memory=...)`` with ``num_workers=0``) for data that is already in
memory.  At the ~0.75 M images/s of the native training step the reference's per-item Python ``__getitem__`` + fp32
host-to-device copy is three orders of magnitude too slow; a 64x64x3 data set of 202 599 images is 2.5 GB as uint8 (of
288 GB of HBM).

``DeviceImageLoader`` is iterable like the reference's ``DataLoader``: ``len()`` = number of batches (no drop_last: the
last batch may be smaller, datasets.py:67-71), every item is ``(batch, labels)`` with ``batch`` a uint8 ``[B,C,H,W]``
device tensor (pixel values 0..255, NCHW = ToTensor's output order).  The native ``Trainer`` / loss plugins accept
such batches directly (``engine.input``): conv1 forward, conv1 weight gradient and the likelihood target read 1 byte per
pixel and divide by 255 on the fly, bit-identical to ``ToTensor`` followed by the fp32 kernels.
"""
import numpy as np
import torch


def to_uint8_nchw(images):
    """Images as the reference's datasets hold them -> uint8 [N,C,H,W] tensor (host).

    * dSprites ``imgs`` (datasets.py:148,206): [N,H,W] with values {0,1} -> pixels {0,255}, C = 1;
    * HWC uint8 arrays as ``imread`` returns them (datasets.py:284): [N,H,W,C] -> permuted to [N,C,H,W]
      (what ToTensor does per item);
    * already [N,C,H,W] uint8: unchanged."""
    t = torch.as_tensor(np.asarray(images))
    if t.dtype != torch.uint8:
        raise TypeError("expected uint8 pixel data, got %s" % t.dtype)
    if t.dim() == 3:                                   # dSprites: binary [N,H,W]
        if int(t.max()) <= 1:
            t = t * 255                                # datasets.py:206
        return t.unsqueeze(1).contiguous()
    if t.dim() != 4:
        raise ValueError("expected [N,H,W], [N,H,W,C] or [N,C,H,W], got %s" % (tuple(t.shape),))
    if t.shape[-1] in (1, 3) and t.shape[1] not in (1, 3):
        t = t.permute(0, 3, 1, 2)                      # HWC -> CHW (ToTensor)
    return t.contiguous()


class _DatasetView:
    def __init__(self, loader):
        self._loader = loader

    def __len__(self):
        return self._loader.n_images


class DeviceImageLoader:
    """uint8 image set resident on the device + batch iterator (see the module docstring).

    images: anything ``to_uint8_nchw`` accepts, or a uint8 [N,C,H,W] device tensor; labels: optional [N, ...] tensor
    (dSprites ``lat_values``); without labels the second item of a batch is 0 like CelebA's placeholder (datasets.py:291).
    shuffle: the order of an epoch is drawn exactly as ``DataLoader(dataset, batch_size, shuffle=True)`` draws it
    (datasets.py:67-71) -- the iterator's base seed and the RandomSampler's seed from the global CPU generator, then
    ``torch.randperm(n)`` from a generator of its own -- so the same ``torch.manual_seed`` gives the same batches and leaves
    the CPU generator in the same state as the reference's loader (tests/test_host_logic.py).  The permutation is moved to
    the device ONCE per epoch; a batch is one on-device gather.
    rank / world_size: data parallelism (disvae_amd.parallel).  Every rank draws the SAME permutation (the CPU seed is
    shared: FactorVAE's permute_dims needs that anyway) and takes rows [rank*B, (rank+1)*B) of each global batch of
    world_size*B images, the rank order the global-batch estimators assume; ``len()`` counts global batches.  A ragged last
    global batch is split evenly (up to world_size - 1 images of the epoch are dropped so that all ranks see the same
    batch shape, which the collectives require)."""

    def __init__(self, images, batch_size=64, shuffle=True, labels=None, device="cuda", rank=0, world_size=1):
        if isinstance(images, torch.Tensor) and images.is_cuda:
            if images.dtype != torch.uint8 or images.dim() != 4:
                raise TypeError("device image sets must be uint8 [N,C,H,W]")
            self.images = images.contiguous()
        else:
            self.images = to_uint8_nchw(images).to(device)
        self.labels = None if labels is None else torch.as_tensor(labels)
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        if not (0 <= int(rank) < int(world_size)):
            raise ValueError("rank %r outside world_size %r" % (rank, world_size))
        self.rank, self.world_size = int(rank), int(world_size)
        self.dataset = _DatasetView(self)              # ``len(loader.dataset)`` = images, as main.py:201 logs it

    @property
    def n_images(self):
        return self.images.shape[0]

    def _global_batches(self):
        """[(first index into the epoch's order, images per rank)] of every global batch."""
        n, bg = self.n_images, self.batch_size * self.world_size
        out = []
        for i in range(0, n, bg):
            per = min(self.batch_size, (n - i) // self.world_size)
            if per > 0:
                out.append((i, per))
        return out

    def __len__(self):
        """number of batches (training.py:118), the last one possibly smaller."""
        return len(self._global_batches())

    @staticmethod
    def epoch_order(n):
        """The index order ``DataLoader(shuffle=True, num_workers=0)`` iterates an n-item data set in, with the same
        consumption of the global CPU generator (torch/utils/data/dataloader.py: the iterator's base seed;
        sampler.py RandomSampler.__iter__: a seed for a private generator, then randperm)."""
        torch.empty((), dtype=torch.int64).random_()                       # _BaseDataLoaderIter._base_seed (unused here)
        seed = int(torch.empty((), dtype=torch.int64).random_().item())   # RandomSampler.__iter__
        g = torch.Generator()
        g.manual_seed(seed)
        return torch.randperm(n, generator=g)

    def __iter__(self):
        dev = self.images.device
        order = order_dev = None
        if self.shuffle:
            order = self.epoch_order(self.n_images)
            order_dev = order.to(dev)                  # ONE host-to-device copy per epoch
        for i, per in self._global_batches():
            lo = i + self.rank * per
            if order is None:
                batch = self.images[lo:lo + per]
                lab = 0 if self.labels is None else self.labels[lo:lo + per]
            else:
                batch = self.images.index_select(0, order_dev[lo:lo + per])   # gather on the device: uint8, B x C x H x W bytes
                lab = 0 if self.labels is None else self.labels[order[lo:lo + per]]
            yield batch, lab
