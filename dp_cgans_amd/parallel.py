"""Data parallelism for the training step: one process per GPU, torch.distributed (backend "nccl"
is RCCL on ROCm) over xGMI.  What crosses ranks per step: BatchNorm column sums
(fp64, 2*g_l numbers per Residual block, forward and backward), the flat discriminator gradient
buffer, the flat generator gradient buffer.  Parameters, optimiser state and the sampler tables are
replicated; each rank draws its own batch shard from its own Philox streams, while the DP noise
comes from a stream all ranks share, so it is added exactly once and replicas stay identical."""
import os

import torch
import torch.distributed as dist


class Dist(object):
    def __init__(self, backend=None, device=None):
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        # the GPU this rank computes on (RCCL binds its communicator to the current device)
        self.device_index = self.local_rank if device is None or torch.device(device).index is None \
            else torch.device(device).index
        # DPCGAN_FORCE_DIST=1 runs the collectives even with a single rank: that is how the RCCL path
        # (process group, all-reduce on the device, graph segments cut at the collectives) is
        # exercised on a one-GPU box
        self.active = self.world > 1 or os.environ.get('DPCGAN_FORCE_DIST') == '1'
        self.owns_group = False       # True when THIS object created the process group (then shutdown() destroys it)
        if self.active and not dist.is_initialized():
            self.owns_group = True
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            if backend is None:
                backend = os.environ.get('DPCGAN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
            if backend == 'nccl':
                torch.cuda.set_device(self.device_index)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
        self.backend = dist.get_backend() if self.active else None

    def allreduce(self, t):
        """In-place sum over ranks.  RCCL reduces device buffers directly; under gloo (CPU rehearsal,
        or several ranks sharing one GPU in the tests) device tensors are staged through the host."""
        if self.active:
            if t.is_cuda and self.backend == 'gloo':
                h = t.detach().cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def barrier(self):
        if self.active:
            if self.backend == 'nccl':
                dist.barrier(device_ids=[self.device_index])
            else:
                dist.barrier()

    def max_float(self, x, device):
        t = torch.tensor([x], dtype=torch.float64, device='cpu' if self.backend == 'gloo' else device)
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def broadcast_int(self, value, device):
        """Rank 0's 63-bit integer on every rank (a sum in which only rank 0 contributes)."""
        if not self.active:
            return int(value)
        t = torch.tensor([int(value) if self.rank == 0 else 0], dtype=torch.int64,
                         device='cpu' if self.backend == 'gloo' else device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def shutdown(self):
        """Destroy the process group if this object created it (a group the caller set up stays the caller's); the
        object then behaves as a single rank: no collectives."""
        if self.active and self.owns_group and dist.is_initialized():
            dist.destroy_process_group()
            self.owns_group = False
            self.active = False
