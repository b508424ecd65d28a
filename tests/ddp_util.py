"""Test transport for the 2-ranks-on-ONE-GPU tests: gloo (RCCL refuses two ranks on one device) with device tensors staged
through the host.  The product communicators (disvae_amd.parallel.Comm / RcclComm) have no such branch."""
import torch
import torch.distributed as dist

from disvae_amd._lib import record_on_stream
from disvae_amd.parallel import Comm


class HostStagedComm(Comm):
    def _stage(self, fn, out, t):
        def run():
            ho, ht = out.cpu().contiguous(), t.cpu().contiguous()
            fn(ho.view(-1), ht.view(-1))          # gloo's *_tensor collectives take flat buffers
            out.copy_(ho)
        record_on_stream(run)                     # recordable like the product's collectives (launch-plan replay)

    def all_gather_into(self, out, t):
        self._stage(lambda o, i: dist.all_gather_into_tensor(o, i, group=self.group), out, t)

    def reduce_scatter_into(self, out, t):
        self._stage(lambda o, i: dist.reduce_scatter_tensor(o, i, op=dist.ReduceOp.SUM, group=self.group), out, t)
