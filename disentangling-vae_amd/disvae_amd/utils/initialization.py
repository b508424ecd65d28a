"""Weight initialisation with the reference's RNG stream (disvae/utils/initialization.py:33-61,
vae.py:87-88): every layer first draws torch's default init in registration order (weight
kaiming_uniform_(a=sqrt(5)), bias U(+-1/sqrt(fan_in))), then ``apply(weights_init)`` re-draws
every weight with kaiming_uniform_(nonlinearity='relu') in the same order.  Run on the CPU
generator, once, so that identical seeds give bit-identical initial weights."""
import math

import torch


def reference_init_(arena, names):
    """names: layer prefixes in registration order; fills arena.flat (CPU) in place."""
    for n in names:
        w = arena.view(n + ".weight")
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        fan_in = w.size(1) * (w[0][0].numel() if w.dim() > 2 else 1)
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        torch.nn.init.uniform_(arena.view(n + ".bias"), -bound, bound)
    for n in names:
        torch.nn.init.kaiming_uniform_(arena.view(n + ".weight"), nonlinearity="relu")
