"""Data-parallel training across the GPUs of one node: one process per GPU, gradients summed with RCCL over xGMI.

The reference is single-device (train.py:68).  Sharding rule: images are independent units; rank r
takes a contiguous slice of the global batch, weights are replicated, the local loss is the mean over the local
images, and the only exchange is an all-reduce(sum) of the flat gradient buffer of the 52 trainable tensors
(11,578,222 fp32 = 46.3 MB) scaled by 1/world -- which equals the reference's mean over the global batch.  BatchNorm
statistics stay per-GPU (the reference has no SyncBN); running stats can be averaged with sync_bn_stats().

xGMI is point-to-point (7 links per GPU), so the payload goes out as a few large contiguous buckets rather than 52
small tensors: per-call latency, not bandwidth, is what a 46 MB payload would otherwise pay 52 times.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torch.distributed rendezvous from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, world, local)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('SSD_FORCE_DEVICE') is not None:      # rehearsal: every rank on the same device
        local = int(os.environ['SSD_FORCE_DEVICE'])
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # SSD_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the data-parallel path without RCCL
            backend = os.environ.get('SSD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(global_batch, rank, world):
    """Contiguous image slice [lo, hi) of rank `rank` (rank r gets images [32r, 32r+32))."""
    per = global_batch // world
    if per * world != global_batch:
        raise ValueError(f'global batch {global_batch} is not divisible by world size {world}')
    return rank * per, (rank + 1) * per


class GradAllReduce:
    """Engine.dist_hook: sums the flat gradient buffer over ranks with a few large all-reduce calls, overlapped with the
    end of backward: `start(flat, lo, hi)` is called as soon as flat[lo:hi] is final (everything except the first tail
    stage, whose wgrad is the largest and runs last); `finish(flat)` reduces the rest and waits.  torch's NCCL(RCCL)
    process group orders each collective after the kernels already enqueued on the current stream and runs it on its own
    stream, so the early buckets travel over xGMI while the last wgrad kernels compute.
    The 1/world factor is applied by the optimizer (FusedAdam grad_scale) so no extra pass over the buffer is needed;
    with average=True the buffer itself is scaled (stock torch.optim users)."""

    def __init__(self, group=None, n_buckets=2, average=False):
        self.group, self.n_buckets, self.average = group, max(1, n_buckets), average
        self._works, self._done_from = [], None

    def bucket_ranges(self, n, lo=0):
        step = -(-(n - lo) // self.n_buckets)
        step = max(4, -(-step // 4) * 4)
        return [(a, min(n, a + step)) for a in range(lo, n, step)]

    def _active(self):
        return dist.is_initialized() and dist.get_world_size(self.group) > 1

    def start(self, flat, lo, hi):
        if not self._active():
            return
        assert hi == flat.numel() and lo % 4 == 0
        self._done_from = lo
        for a, b in self.bucket_ranges(hi, lo):
            self._works.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self, flat):
        if not self._active():
            return
        end = flat.numel() if self._done_from is None else self._done_from
        if end > 0:
            self._works.append(dist.all_reduce(flat[:end], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in self._works:
            w.wait()
        self._works, self._done_from = [], None
        if self.average:
            flat.mul_(1.0 / dist.get_world_size(self.group))

    def __call__(self, flat):
        self.finish(flat)


def attach(net, n_buckets=2, average=False, group=None):
    """Install the gradient all-reduce on an SSD module; returns the world size (use 1/world as FusedAdam grad_scale
    unless average=True)."""
    hook = GradAllReduce(group=group, n_buckets=n_buckets, average=average)
    net._dist_hook = hook                # kept on the module: finetune_backbone() rebuilds the engine and must not drop it
    net._engine().dist_hook = hook
    return dist.get_world_size(group) if dist.is_initialized() else 1


@torch.no_grad()
def broadcast_module(net, src=0, group=None):
    """Make every rank start from rank `src`'s model: parameters AND buffers (BatchNorm running statistics, step counters).
    SSD() initialises with the process's own RNG state, which differs per process unless seeded, and the reference has no
    notion of ranks; without this each replica would train a different model on averaged gradients.  One flat message per
    dtype (a few large sends, as for the gradients)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    by_dtype = {}
    for t in list(net.parameters()) + list(net.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, ts in by_dtype.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src=src, group=group)
        o = 0
        for t in ts:
            t.detach().copy_(flat[o:o + t.numel()].view_as(t))
            o += t.numel()
    if hasattr(net, 'hip_weights_changed'):
        net.hip_weights_changed()            # raw copies above: packed / Winograd-domain weight images must be rebuilt


@torch.no_grad()
def broadcast_optimizer_state(opt, src=0, group=None):
    """Same for a resumed optimizer: Adam moments and step counts of rank `src` (FusedAdam.state_dict layout = torch's)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for g in opt.param_groups:
        for p in g['params']:
            st = opt.state.get(p)
            if not st:
                continue
            for k in ('exp_avg', 'exp_avg_sq'):
                if k in st:
                    dist.broadcast(st[k], src=src, group=group)
            mine = st.get('step', 0)
            step = torch.tensor([float(mine)], device=p.device)
            dist.broadcast(step, src=src, group=group)
            st['step'] = torch.tensor(step.item()) if torch.is_tensor(mine) else int(step.item())
    if hasattr(opt, 'invalidate'):
        opt.invalidate()


def global_mean(value, device=None, group=None):
    """Mean over ranks of a per-rank Python float (epoch loss): every rank gets the same number, so decisions taken from it
    (checkpoint or not -> a collective inside) cannot diverge between ranks."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    return t.item() / dist.get_world_size(group)


@torch.no_grad()
def sync_bn_stats(net, group=None):
    """Average BatchNorm running statistics over ranks (checkpoint time); ~22k floats, one call."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    bufs = [b for n, b in net.named_buffers() if n.endswith('running_mean') or n.endswith('running_var')]
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.all_reduce(flat, group=group)
    flat /= dist.get_world_size(group)
    o = 0
    for b in bufs:
        b.copy_(flat[o:o + b.numel()].view_as(b))
        o += b.numel()
