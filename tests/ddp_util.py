"""Test transport for the 2-ranks-on-ONE-GPU tests: gloo (RCCL refuses two ranks on one device) with device tensors staged
through the host.  The product communicators (disvae_amd.parallel.Comm / RcclComm) have no such branch."""
import torch
import torch.distributed as dist

from disvae_amd._lib import record_on_stream
from disvae_amd.parallel import Comm, _Done


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

    # gloo's own handling of device tensors (async_op on a second CUDA stream of its choosing) raced once in ~10 runs with two
    # ranks on one GPU (a discriminator gradient read before its all-reduce had landed): the sum-all-reduces are staged through
    # the host like the other two collectives, synchronously -- numerics of the sharded step are what these tests check, the
    # overlap of the asynchronous form belongs to the product transports (torch NCCL / RcclComm)
    def all_reduce(self, t):
        def run():
            h = t.cpu().contiguous()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h.view_as(t))
        record_on_stream(run)
        return t

    def all_reduce_async(self, t):
        self.all_reduce(t)
        return _Done()
