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
        if 'HSA_ENABLE_IPC_MODE_LEGACY' not in os.environ:
            if torch.cuda.is_initialized():
                # the runtime read its environment at the first HIP call: setting the variable now would change nothing
                import warnings
                warnings.warn('init_from_env: the GPU was initialised before HSA_ENABLE_IPC_MODE_LEGACY=0 could be set; RCCL peer mappings '
                              '(hipIpcGetMemHandle) fail in the legacy IPC mode on this driver -- export it in the launching environment',
                              RuntimeWarning, stacklevel=2)
            os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # SSD_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the data-parallel path without RCCL
            backend = os.environ.get('SSD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def host_cores():
    """CPU cores this process may actually use: cgroup quota if set (a GPU box gives one GPU's share), else affinity."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16) if n > 64 else n      # unrestricted many-core host: use the documented 16-core share of one GPU


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_local_ranks(n, command):
    """Start `command` (argv list) n times as fresh rank processes of one node -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, one per GPU -- from a process that has not touched the GPU and never does (never by replacing itself).  Rank 0's
    stdout is relayed, the other ranks' goes to stderr; the first failing rank ends the job with its exit code (its peers would
    wait in a collective forever).  Used by `bench.py --gpus N` and `train.py --device-count N` when no launcher set WORLD_SIZE."""
    import subprocess
    import sys
    import time
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL peer mappings fail without it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, host_cores() // n)))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(command), env=e, stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                if code != 0:
                    rc = code if code > 0 else 128 - code
                    print(f'{os.path.basename(command[1]) if len(command) > 1 else command[0]}: rank {procs.index(p)} exited with code {code}; '
                          'stopping the other ranks', file=sys.stderr)
    for p in live:
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def shard_batch(global_batch, rank, world):
    """Contiguous image slice [lo, hi) of rank `rank` (rank r gets images [32r, 32r+32))."""
    per = global_batch // world
    if per * world != global_batch:
        raise ValueError(f'global batch {global_batch} is not divisible by world size {world}')
    return rank * per, (rank + 1) * per


class GradAllReduce:
    """Engine.dist_hook: sums the flat gradient buffer over ranks with a few large collectives, overlapped with the
    end of backward: `start(flat, lo, hi)` is called as soon as flat[lo:hi] is final (everything except the first tail
    stage, whose wgrad is the largest and runs last); `finish(flat)` reduces the rest and waits.  torch's NCCL(RCCL)
    process group orders each collective after the kernels already enqueued on the current stream and runs it on its own
    stream, so the early buckets travel over xGMI while the last wgrad kernels compute.
    The 1/world factor is applied by the optimizer (FusedAdam grad_scale) so no extra pass over the buffer is needed;
    with average=True the buffer itself is scaled (stock torch.optim users).

    mode 'allreduce' (default): one all-reduce per bucket; the algorithm (ring / tree) is RCCL's choice.
    mode 'rs_ag' (SURVEY.md 8(e)): reduce-scatter + all-gather per bucket, in place -- on a fully connected xGMI node each rank
        then exchanges 1/world of the bucket with every peer over its own link (7 links at once) instead of passing the whole
        bucket around a ring bound by one link: 2 * (46.3 MB / 8) / 153 GB/s = 0.08 ms against 0.53 ms for the 8-GPU node.
        The few elements past the last multiple of `world` go through a (tiny) all-reduce.
    payload 'bf16' (optional, 23 MB instead of 46 MB): the bucket is rounded to bf16, summed in bf16 and widened back; the sum of
        `world` bf16 values carries ~3 significant digits, which Adam's normalised step tolerates but which is NOT the reference's
        arithmetic -- off by default.
    Chains of dependent device operations (convert -> collective -> convert, reduce-scatter -> all-gather) are issued on one
    communication stream behind an event of the caller's stream, so the caller's stream never waits before `finish`."""

    def __init__(self, group=None, n_buckets=2, average=False, force=False, mode='allreduce', payload='fp32'):
        if mode not in ('allreduce', 'rs_ag'):
            raise ValueError(f"GradAllReduce: mode must be 'allreduce' or 'rs_ag', got {mode!r}")
        if payload not in ('fp32', 'bf16'):
            raise ValueError(f"GradAllReduce: payload must be 'fp32' or 'bf16', got {payload!r}")
        self.group, self.n_buckets, self.average = group, max(1, n_buckets), average
        self.mode, self.payload = mode, payload
        self.force = force         # issue the collectives even in a one-rank group (RCCL rehearsal on a one-GPU box: tests/test_gpu_dist.py)
        self._works, self._done_from, self._extra = [], None, []
        self._comm, self._events, self._stage = {}, [], {}
        self.log = []              # (kind, elements, bytes on the wire per rank) of the last step's collectives: bench.py prints it

    def bucket_ranges(self, n, lo=0):
        step = -(-(n - lo) // self.n_buckets)
        step = max(4, -(-step // 4) * 4)
        return [(a, min(n, a + step)) for a in range(lo, n, step)]

    def _active(self):
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force)

    def describe(self):
        """What a first multi-GPU run needs to diagnose itself: ranks the communicator really has, backend, mode, payload."""
        ok = dist.is_initialized()
        return dict(ranks_seen=dist.get_world_size(self.group) if ok else 1, backend=dist.get_backend(self.group) if ok else None,
                    mode=self.mode, payload=self.payload, buckets_early=self.n_buckets, collectives_last_step=list(self.log))

    # ---- one bucket ------------------------------------------------------------------------------------------------
    def _plain(self):
        return self.mode == 'allreduce' and self.payload == 'fp32'

    def _chain(self, t):
        """convert / reduce-scatter / all-gather chain for one contiguous fp32 bucket `t`, in issue order on the current stream."""
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        buf = t
        if self.payload == 'bf16':
            st = self._stage.get((t.device, t.numel()))
            if st is None:
                st = torch.empty(t.numel(), device=t.device, dtype=torch.bfloat16)
                self._stage[(t.device, t.numel())] = st
            st.copy_(t)
            buf = st
        esz = buf.element_size()
        if self.mode == 'rs_ag' and buf.numel() >= world:
            chunk = buf.numel() // world
            body, rest = buf[:chunk * world], buf[chunk * world:]
            mine = body[rank * chunk:(rank + 1) * chunk]
            dist.reduce_scatter_tensor(mine, body, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(body, mine, group=self.group)
            self.log.append(('reduce_scatter+all_gather', body.numel(), 2 * (world - 1) * chunk * esz))
            if rest.numel():
                dist.all_reduce(rest, op=dist.ReduceOp.SUM, group=self.group)
                self.log.append(('all_reduce(remainder)', rest.numel(), rest.numel() * esz))
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            self.log.append(('all_reduce', buf.numel(), buf.numel() * esz))
        if buf is not t:
            t.copy_(buf)

    def _issue(self, t):
        if self._plain():
            self._works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.log.append(('all_reduce', t.numel(), t.numel() * 4))
            return
        if not t.is_cuda:
            self._chain(t)
            return
        comm = self._comm.get(t.device)
        if comm is None:
            comm = self._comm[t.device] = torch.cuda.Stream(device=t.device, priority=-1)
        comm.wait_event(torch.cuda.current_stream(t.device).record_event())
        with torch.cuda.stream(comm):
            self._chain(t)
            self._events.append(comm.record_event())

    # ---- hook points -----------------------------------------------------------------------------------------------
    def start(self, flat, lo, hi):
        self.log = []
        if not self._active():
            return
        assert hi == flat.numel() and lo % 4 == 0
        self._done_from = lo
        for a, b in self.bucket_ranges(hi, lo):
            self._issue(flat[a:b])

    def launch(self, t):
        """Asynchronous all-reduce(sum) of one more final, contiguous gradient tensor that lives outside the engine's flat buffer (the
        classifier stack of VGG16: 411 + 67 MB).  Large tensors travel as they are (one call each); the collective is ordered after the
        kernels already enqueued on the current stream and is waited for by `wait_launched()` or the next `finish()`."""
        if not self._active():
            return
        if not t.is_contiguous():
            raise ValueError('GradAllReduce.launch: contiguous tensors only')
        self._issue(t.view(-1))
        if self.average:
            self._extra.append(t)

    def wait_launched(self):
        """Make the current stream wait for every collective launched so far (a no-op when finish() already did)."""
        for w in self._works:
            w.wait()
        self._works = []
        for ev in self._events:
            torch.cuda.current_stream().wait_event(ev)
        self._events = []
        if self.average and self._extra:
            torch._foreach_mul_(self._extra, 1.0 / dist.get_world_size(self.group))
        self._extra = []

    def finish(self, flat):
        if not self._active():
            return
        end = flat.numel() if self._done_from is None else self._done_from
        if self._done_from is None:
            self.log = []
        if end > 0:
            self._issue(flat[:end])
        self._done_from = None
        self.wait_launched()
        if self.average:
            flat.mul_(1.0 / dist.get_world_size(self.group))

    def __call__(self, flat):
        self.finish(flat)


def attach(net, n_buckets=2, average=False, group=None, mode='allreduce', payload='fp32'):
    """Install the gradient all-reduce on an SSD module; returns the world size (use 1/world as FusedAdam grad_scale
    unless average=True).  mode / payload: see GradAllReduce."""
    hook = GradAllReduce(group=group, n_buckets=n_buckets, average=average, mode=mode, payload=payload)
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
    if any(g.get('amsgrad') for g in opt.param_groups):
        # Adam(amsgrad=True) keeps a third moment (max_exp_avg_sq) and skips its lazy initialisation once ANY state exists: creating
        # only the two moments below would make its next step() fail with a KeyError
        raise NotImplementedError('broadcast_optimizer_state: amsgrad optimizers are not supported (max_exp_avg_sq is not broadcast)')
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
    limit = 64 << 20                             # elements per message (256 MB of fp32): bounded staging copy, same sequence on every rank
    for (_, _), ts in sorted(by_dev.items(), key=lambda kv: str(kv[0])):
        group_ts, count = [], 0
        for t in ts + [None]:
            if t is not None and (not group_ts or count + t.numel() <= limit):
                group_ts.append(t); count += t.numel()
                continue
            if len(group_ts) == 1 and group_ts[0].is_contiguous():
                dist.broadcast(group_ts[0], src=src, group=group)        # a large tensor travels as it is (VGG16 fc1 moments: 411 MB)
            else:
                flat = torch.cat([g_.reshape(-1) for g_ in group_ts])
                dist.broadcast(flat, src=src, group=group)
                o = 0
                for g_ in group_ts:
                    g_.copy_(flat[o:o + g_.numel()].view_as(g_))
                    o += g_.numel()
            group_ts, count = ([t], t.numel()) if t is not None else ([], 0)
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
