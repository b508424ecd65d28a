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
    shuffle: a fresh ``torch.randperm`` of the data set per epoch from the torch CPU generator, like
    ``DataLoader(shuffle=True)``'s RandomSampler (seeded by ``torch.manual_seed``)."""

    def __init__(self, images, batch_size=64, shuffle=True, labels=None, device="cuda"):
        if isinstance(images, torch.Tensor) and images.is_cuda:
            if images.dtype != torch.uint8 or images.dim() != 4:
                raise TypeError("device image sets must be uint8 [N,C,H,W]")
            self.images = images.contiguous()
        else:
            self.images = to_uint8_nchw(images).to(device)
        self.labels = None if labels is None else torch.as_tensor(labels)
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.dataset = _DatasetView(self)              # ``len(loader.dataset)`` = images, as main.py:201 logs it

    @property
    def n_images(self):
        return self.images.shape[0]

    def __len__(self):
        """number of batches (training.py:118), the last one possibly smaller."""
        return (self.n_images + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.n_images
        order = torch.randperm(n) if self.shuffle else None
        dev = self.images.device
        for i in range(0, n, self.batch_size):
            if order is None:
                batch = self.images[i:i + self.batch_size]
                lab = 0 if self.labels is None else self.labels[i:i + self.batch_size]
            else:
                idx = order[i:i + self.batch_size]
                batch = self.images.index_select(0, idx.to(dev))       # gather on the device: uint8, B x C x H x W bytes
                lab = 0 if self.labels is None else self.labels[idx]
            yield batch, lab
