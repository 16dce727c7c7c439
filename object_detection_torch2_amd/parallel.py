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
        # dmabuf IPC: on this driver RCCL's peer mappings (hipIpcGetMemHandle) fail in the legacy mode; must be in the environment
        # before the first HIP call of the process, whichever entry point (bench.py, train.py under torchrun) got here
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
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

    def __init__(self, group=None, n_buckets=2, average=False, force=False):
        self.group, self.n_buckets, self.average = group, max(1, n_buckets), average
        self.force = force         # issue the collectives even in a one-rank group (RCCL rehearsal on a one-GPU box: tests/test_gpu_dist.py)
        self._works, self._done_from, self._extra = [], None, []

    def bucket_ranges(self, n, lo=0):
        step = -(-(n - lo) // self.n_buckets)
        step = max(4, -(-step // 4) * 4)
        return [(a, min(n, a + step)) for a in range(lo, n, step)]

    def _active(self):
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force)

    def start(self, flat, lo, hi):
        if not self._active():
            return
        assert hi == flat.numel() and lo % 4 == 0
        self._done_from = lo
        for a, b in self.bucket_ranges(hi, lo):
            self._works.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def launch(self, t):
        """Asynchronous all-reduce(sum) of one more final, contiguous gradient tensor that lives outside the engine's flat buffer (the
        classifier stack of VGG16: 411 + 67 MB).  Large tensors travel as they are (one call each); the collective is ordered after the
        kernels already enqueued on the current stream and is waited for by `wait_launched()` or the next `finish()`."""
        if not self._active():
            return
        if not t.is_contiguous():
            raise ValueError('GradAllReduce.launch: contiguous tensors only')
        self._works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if self.average:
            self._extra.append(t)

    def wait_launched(self):
        """Make the current stream wait for every collective launched so far (a no-op when finish() already did)."""
        for w in self._works:
            w.wait()
        self._works = []
        if self.average and self._extra:
            torch._foreach_mul_(self._extra, 1.0 / dist.get_world_size(self.group))
        self._extra = []

    def finish(self, flat):
        if not self._active():
            return
        end = flat.numel() if self._done_from is None else self._done_from
        if end > 0:
            self._works.append(dist.all_reduce(flat[:end], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._done_from = None
        self.wait_launched()
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
    """Same for a resumed optimizer: Adam moments and step counts of rank `src` (FusedAdam.state_dict layout = torch's).
    The typical caller is a resume where ONLY rank `src` loaded a checkpoint, so the other ranks have no state at all: the
    sequence of collectives must not depend on what a rank happens to hold.  Every rank therefore first materialises zero
    moments and step 0 for every parameter of the optimizer (what Adam itself would create lazily), then takes part in the same
    two broadcasts: one flat message with all moments, one with all step counts."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    fused = hasattr(opt, 'invalidate')           # FusedAdam keeps `step` as a Python int, torch.optim.Adam as a float32 scalar tensor
    params = [p for g in opt.param_groups for p in g['params']]
    if not params:
        return
    moments = []
    for p in params:
        st = opt.state[p]                        # defaultdict: creates the empty record
        for k in ('exp_avg', 'exp_avg_sq'):
            if k not in st:
                st[k] = torch.zeros_like(p, memory_format=torch.preserve_format)
            moments.append(st[k])
        if 'step' not in st:
            st['step'] = 0 if fused else torch.tensor(0.0, dtype=torch.float32)
    by_dev = {}
    for t in moments:
        by_dev.setdefault((t.device, t.dtype), []).append(t)
    for (_, _), ts in sorted(by_dev.items(), key=lambda kv: str(kv[0])):
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src=src, group=group)
        o = 0
        for t in ts:
            t.copy_(flat[o:o + t.numel()].view_as(t))
            o += t.numel()
    steps = torch.tensor([float(opt.state[p]['step']) for p in params], dtype=torch.float64, device=params[0].device)
    dist.broadcast(steps, src=src, group=group)
    for p, v in zip(params, steps.tolist()):
        mine = opt.state[p]['step']
        opt.state[p]['step'] = torch.tensor(v, dtype=mine.dtype, device=mine.device) if torch.is_tensor(mine) else int(v)
    if fused:
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
