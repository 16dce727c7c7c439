"""The train step of the reference loop (train.py:115-122: zero_grad -> forward -> loss -> backward -> optimizer.step) as replayed
HIP graphs: host time per step drops from ~3.6 ms of Python + ctypes (≈270 launches) to 0.6 ms (two graph launches).

Shape of a step (MI355X-first, not a capture of the eager call sequence):

    backbone stream :  BB(i+1)  ─ frozen VGG16-bn of the NEXT batch: one graph, a linear chain of ≈45 launches
    caller's stream :  T(i)     ─ detector heads + trainable tail forward, MultiBox loss, tail backward, fused Adam of THIS batch:
                                  one graph, by default a linear chain too (`linear_tail`)

The two chains never wait for each other inside a step: what T reads of BB (z + BatchNorm scale / shift of conv_4_3 for det_4_3,
the last frozen activation for conv_6_1) exists twice, and consecutive batches alternate between the copies (`parity`), so BB(i+1)
writes copy (i+1)%2 while T(i) reads copy i%2.  Two events per step order them: T(i) waits for BB(i); BB(i+2) waits for T(i).

Measured on one MI355X (scripts/graph_probe.py, graph_stream_probe.py): host enqueue 3.6-4.0 -> 0.6 ms per step;
device time 10.4 ms (fp32) / 5.3 ms (bf16) per step against 9.7 / 4.8 ms for the eager engine, so this mode is opt-in (`--graph`):
it buys host time, not throughput.  Two findings shape it.  (1) A tail captured WITH its internal concurrency (heads / weight
gradients / repacking on a forked stream) replays on streams the runtime creates at instantiation with normal priority; which
hardware queue they land on relative to the backbone stream decides between 10.4 and 14.1 ms (fp32; 5.8 .. 9.7 bf16) for the SAME
graph launched on different caller streams -- a lottery.  A linear tail replays entirely on the caller's stream: the same time on
every stream tried.  (2) Why the device time is higher than eager's although the two chains overlap MORE here (rocprofv3 kernel trace of both
modes, one step: 4.9 ms with two or more kernels running against 2.0 ms): kernels that share the chip stretch each other -- the tail's
208 launches are busy for 7.4 ms here against 5.9 ms eager (4.3 ms alone), the backbone's for 8.6 against 7.6 (7.35 alone).  The
densely packed tail meets the backbone's heaviest layers (conv_1_2 .. conv_3_1, persistent grids on 224 CUs); moving the det_4_3
head behind the backbone in the EAGER engine changes nothing (9.65 vs 9.62 ms), so it is the packing, not a placement.

Data parallelism: the gradient all-reduce stays an eager RCCL call between graph segments -- the engine's hook points cut T into
T1 (everything up to the last weight gradient) | all-reduce of flat[cut:] | T2 (conv_6_1's weight gradient) | all-reduce of
flat[:cut] + wait | T3 (Adam).  BB(i+1) is already queued on its own stream and keeps the chip busy across the cuts.

Results are bit-identical to the eager step: same kernels, same per-stream order, same buffers (tests/test_gpu_graph_step.py).
"""
import os

import torch

from . import _lib
from .model import topology as T


_DBG = set(filter(None, os.environ.get('SSD_GS_DEBUG', '').split(',')))      # timing experiments only (scripts/graph_probe.py)


class _Cut:
    """Stand-in for the engine's dist_hook during capture: each hook point ends the running capture segment, notes the eager
    collective to issue there at replay time, and begins the next segment."""

    def __init__(self, recorder):
        self.rec = recorder

    def start(self, flat, lo, hi):
        self.rec.cut(('start', lo, hi))

    def finish(self, flat):
        self.rec.cut(('finish',))


class _Recorder:
    def __init__(self, pool):
        self.pool, self.items, self.g = pool, [], None

    def begin(self):
        self.g = torch.cuda.CUDAGraph()
        self.g.capture_begin(pool=self.pool, capture_error_mode='thread_local')

    def end(self):
        self.g.capture_end()
        self.items.append(self.g)
        self.g = None

    def cut(self, op):
        self.end()
        self.items.append(op)
        self.begin()


class GraphedTrainStep:
    """runner = GraphedTrainStep(net, optimizer, default_bboxes, images_shape, targets_shape)

        runner.feed(images_0)
        for i in range(steps):
            if i + 1 < steps:
                runner.feed(images_{i+1})          # backbone of the next batch, on its own stream
            loss = runner.step(targets_i)          # this batch's tail forward + loss + backward + Adam; 0-dim tensor, valid
                                                   # until the next step() (clone / accumulate it on the stream)
    `optimizer` must be the FusedAdam over net.train_params(); the runner owns p.grad (views of the engine's flat gradient buffer:
    do not call optimizer.zero_grad()).  The model must be in train mode with its VGG part frozen (the reference's configuration).
    Capturing does not change parameters, buffers or optimizer state (the warm-up steps it needs are rolled back)."""

    def __init__(self, net, optimizer, default_bboxes, images_shape, targets_shape, alpha=1.0, warmup=2, linear_tail=None):
        eng = net._engine()
        self.linear_tail = (os.environ.get('SSD_GS_LINEAR', '1') == '1') if linear_tail is None else bool(linear_tail)
        if not net.training:
            raise RuntimeError('GraphedTrainStep: the module must be in train mode (the reference never calls eval())')
        if not (0 < eng.tail_start < len(eng.stages)) or eng.headless:
            raise RuntimeError('GraphedTrainStep needs the frozen-backbone configuration (not finetune_backbone / VGG16)')
        if not hasattr(optimizer, 'enqueue'):
            raise TypeError('GraphedTrainStep: optimizer must be object_detection_torch2_amd.optim.FusedAdam')
        self.net, self.opt, self.eng = net, optimizer, eng
        self.lib = _lib.load()
        dfb = _lib.f32c(default_bboxes)
        _lib.require_gpu(dfb)
        self.dev = dev = dfb.device
        N, C3, H, W = images_shape
        if C3 != 3 or tuple(targets_shape)[0] != N or tuple(targets_shape)[2] != net.num_classes + 4:
            raise ValueError(f'GraphedTrainStep: images {tuple(images_shape)} / targets {tuple(targets_shape)} do not fit (N, 3, H, W) / (N, G, {net.num_classes + 4})')
        self.N, self.G, self.row = N, tuple(targets_shape)[1], net.num_classes + 4
        self.alpha, self.threshold = float(alpha), float(T.MATCH_THRESHOLD)
        f = dict(device=dev, dtype=torch.float32)
        self._x = torch.zeros(N, 3, H, W, **f)
        self._targets = torch.zeros(N, self.G, self.row, **f)
        self._df = dfb
        P = dfb.shape[0]
        self._terms = torch.empty(3, N, P, **f)
        self._cnt = torch.empty(N, device=dev, dtype=torch.int32)
        self._sel = torch.empty(N, 8, **f)
        self._loss = torch.zeros(1, **f)
        self._one = torch.ones(1, **f)
        self._grad_out = torch.empty(N, P, self.row, **f)
        self.P = P
        self._bb_stream = eng._side_stream(dev, which=2)
        self._bb_done = [torch.cuda.Event(), torch.cuda.Event()]
        self._t_done = [None, None]
        self._fed = []                       # parities of fed, not yet consumed batches (oldest first)
        self._next_parity = 0
        self._real_hook = eng.dist_hook
        # p.grad = views of the flat gradient buffer (what autograd leaves there in the eager step)
        flat, views, params, names = eng._flat_grad(dev)
        for n, q in zip(names, params):
            q.grad = views[n]
        self._flat = flat
        self._capture(warmup)

    # ------------------------------------------------------------------------------------------------ pieces (eager or under capture)
    def _backbone(self, parity):
        self.eng.forward(self._x, save=False, phase='backbone', parity=parity)

    def _tail(self, parity, work):
        eng, L, st = self.eng, self.lib, _lib.stream
        out = eng.forward(None, save=True, phase='tail', parity=parity)
        N, P, G, row = self.N, self.P, self.G, self.row
        _lib.check(L.ssdhip_multibox_loss_fwd(_lib.ptr(out), _lib.ptr(self._targets), _lib.ptr(self._df), N, P, G, row - 4, self.alpha, self.threshold,
                                              _lib.ptr(self._terms), _lib.ptr(self._cnt), _lib.ptr(self._sel), _lib.ptr(self._loss), st()), 'SSD.loss')
        _lib.check(L.ssdhip_multibox_loss_bwd(_lib.ptr(out), _lib.ptr(self._targets), _lib.ptr(self._df), N, P, G, row - 4, self.alpha, self.threshold,
                                              _lib.ptr(self._terms), _lib.ptr(self._sel), _lib.ptr(self._one), _lib.ptr(self._grad_out), st()),
                   'SSD.loss backward')
        eng.backward(self._grad_out)
        self.opt.enqueue(work)
        return out

    # ------------------------------------------------------------------------------------------------ capture
    def _state_tensors(self, work):
        ts = [t for t in self.net.parameters()] + [t for t in self.net.buffers()]
        for plan, params in work:
            ts += [plan['step'], plan['hyper']]
            for p in params:
                ts += [self.opt.state[p]['exp_avg'], self.opt.state[p]['exp_avg_sq']]
        return ts + [self._flat]

    @torch.no_grad()
    def _capture(self, warmup):
        eng, dev = self.eng, self.dev
        work = self.opt.prepare()
        if not work:
            raise RuntimeError('GraphedTrainStep: the optimizer holds no parameter with a gradient')
        self._work = work
        state = self._state_tensors(work)
        snap = [t.detach().clone() for t in state]
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        eng._phased = True
        hook, eng.dist_hook = eng.dist_hook, None
        single = eng.single_stream
        try:
            with torch.cuda.stream(cap):
                # eager passes through exactly the code that is captured next: lazy buffers of both parities, packed weight images,
                # optimizer plan, first-launch work of every kernel.  They train on zeros; everything they touch is rolled back.
                for _ in range(max(1, warmup)):
                    for parity in (0, 1):
                        self._backbone(parity)
                        self._tail(parity, work)
                torch.cuda.synchronize(dev)
                for t, s in zip(state, snap):
                    t.detach().copy_(s)
                torch.cuda.synchronize(dev)
                self._bb_graphs, self._tail_items = [], []
                for parity in (0, 1):
                    rec = _Recorder(torch.cuda.graph_pool_handle())
                    rec.begin()
                    self._backbone(parity)
                    rec.end()
                    self._bb_graphs.append(rec.items[0])
                eng.single_stream = single or self.linear_tail     # linear tail: heads / weight gradients / repacking in line on one stream
                for parity in (0, 1):
                    # every trainable layer's packed image is stale by construction at this point of a step (Adam just ran): make the
                    # capture see it that way, so the repacking launches are part of the graph
                    self.net._hip_epoch += 1
                    rec = _Recorder(torch.cuda.graph_pool_handle())
                    eng.dist_hook = _Cut(rec) if hook is not None else None
                    rec.begin()
                    self._tail(parity, work)
                    rec.end()
                    eng.dist_hook = None
                    self._tail_items.append(rec.items)
                torch.cuda.synchronize(dev)
        finally:
            eng.dist_hook = hook
            eng._phased = False
            eng.single_stream = single
        torch.cuda.current_stream(dev).wait_stream(cap)
        self.net._hip_epoch += 1             # whatever an eager forward finds packed is not what the restored weights say

    # ------------------------------------------------------------------------------------------------ replay
    def feed(self, images, ready_event=None):
        """Enqueue the frozen backbone for `images` -- the batch the next unpaired step() trains on -- on the backbone stream."""
        if len(self._fed) >= 2:
            raise RuntimeError('GraphedTrainStep.feed: two batches are already in flight; call step() first')
        if tuple(images.shape) != tuple(self._x.shape):
            raise ValueError(f'GraphedTrainStep was captured for images {tuple(self._x.shape)}, got {tuple(images.shape)}')
        _lib.require_gpu(images)
        p = self._next_parity
        bb = self._bb_stream
        bb.wait_event(ready_event if ready_event is not None else torch.cuda.current_stream(self.dev).record_event())
        if self._t_done[p] is not None and 'nowait_t' not in _DBG:
            bb.wait_event(self._t_done[p])           # the tail that read this parity's hand-over buffers
        images.record_stream(bb)
        with torch.cuda.stream(bb):
            if 'nocopy' not in _DBG:
                self._x.copy_(images)
            self._bb_graphs[p].replay()
            self._bb_done[p].record(bb)
        self._fed.append(p)
        self._next_parity = p ^ 1

    def step(self, targets):
        """Tail forward + MultiBox loss + backward + Adam for the oldest fed batch, on the caller's stream.  Returns the loss (0-dim
        view of a static tensor)."""
        if not self._fed:
            raise RuntimeError('GraphedTrainStep.step without a fed batch: call feed(images) first')
        g = targets.shape[1] if targets.dim() == 3 else -1
        if targets.dim() != 3 or targets.shape[0] != self.N or targets.shape[2] != self.row or not (0 <= g <= self.G):
            raise ValueError(f'GraphedTrainStep was captured for targets {tuple(self._targets.shape)} (fewer ground-truth rows are '
                             f'padded with zero rows), got {tuple(targets.shape)}')
        _lib.require_gpu(targets)
        p = self._fed.pop(0)
        main = torch.cuda.current_stream(self.dev)
        if 'trace' in _DBG and not getattr(self, '_traced', False):
            self._traced = True
            print(f'[graph_step] step on stream {main} priority {main.priority}; backbone stream priority {self._bb_stream.priority}', flush=True)
        if 'nowait_bb' not in _DBG:
            main.wait_event(self._bb_done[p])
        if 'nocopy' in _DBG:
            pass
        elif g == self.G:
            self._targets.copy_(targets)
        else:
            self._targets.zero_()                    # zero rows are padding: they never match a prior (ssd.py:231-250)
            self._targets[:, :g].copy_(targets)
        work = self._work
        self.opt.refresh(work)                       # lr / weight decay / grad_scale changes reach the device before the replay
        hook = self._real_hook
        for item in self._tail_items[p]:
            if isinstance(item, tuple):
                if item[0] == 'start':
                    hook.start(self._flat, item[1], item[2])
                else:
                    hook.finish(self._flat)
            else:
                item.replay()
        if self._t_done[p] is None:
            self._t_done[p] = torch.cuda.Event()
        self._t_done[p].record(main)
        self.opt.committed(work)                     # host step counts, parameter versions, net._hip_epoch (on_step)
        self.opt._opt_called = True                  # what torch's lr schedulers look at to tell "optimizer.step() ran" (train.py:154)
        if self.opt._on_step is None:
            self.net._hip_epoch += 1
        return self._loss[0]

    def drain(self):
        """Join both chains into the caller's stream (before eager use of the module between graph steps)."""
        if self._fed:
            raise RuntimeError('GraphedTrainStep.drain: a fed batch was never trained on')
        torch.cuda.current_stream(self.dev).wait_stream(self._bb_stream)
        self._bb_stream.wait_stream(torch.cuda.current_stream(self.dev))
