"""Fused multi-tensor Adam on libssdhip.so: same update rule as torch.optim.Adam(lr, weight_decay) used by the
reference harness (train.py:97,122) -- L2 decay added to the gradient, bias-corrected, no amsgrad -- as ONE kernel
launch over all trainable tensors.  Step count / bias corrections live in device memory (graph-replayable)."""
import numpy as np
import torch

from . import _lib

_CHUNK = 8192         # elements per block: 11.6 M parameters -> ~1,450 blocks (65,536 gave 219 blocks for 256 CUs: 3.7 TB/s)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0, on_step=None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError('FusedAdam: invalid hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale))
        self._on_step = on_step
        self._plans = {}

    def invalidate(self):
        """Forget the cached launch plans (device pointer table, device step counter): the next step() rebuilds them from
        `self.state`.  Needed whenever state tensors are replaced behind the optimizer's back."""
        self._plans = {}

    def load_state_dict(self, state_dict):
        """torch's loader replaces exp_avg / exp_avg_sq with new tensors and new step values: a plan cached before it would keep
        writing into the old (freed) moment buffers and keep counting from the old step."""
        super().load_state_dict(state_dict)
        self.invalidate()

    def _plan(self, gi, group, params):
        for p in params:
            st = self.state[p]
            if 'exp_avg' not in st:
                st['step'] = 0
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr())
                    for p in params)
        plan = self._plans.get(gi)
        if plan is not None and plan['key'] == key:
            return plan
        dev = params[0].device
        rows, sizes, blocks = [], [], []
        for ti, p in enumerate(params):
            st = self.state[p]
            for t in (p, p.grad, st['exp_avg'], st['exp_avg_sq']):
                if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                    raise ValueError('FusedAdam: contiguous fp32 parameters, gradients and moments on one device only')
            rows.append([p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()])
            sizes.append(p.numel())
            blocks += [[ti, c] for c in range((p.numel() + _CHUNK - 1) // _CHUNK)]
        b1, b2 = group['betas']
        # the device step counter always restarts from the host-side per-parameter step (kept current by step(), replaced by
        # load_state_dict), like torch.optim.Adam; one shared counter: the tensors of a group step together
        step0 = int(max(int(self.state[p].get('step', 0)) for p in params))
        plan = dict(key=key, n_blocks=len(blocks), gi=gi,
                    table=torch.tensor(rows, dtype=torch.int64, device=dev),
                    sizes=torch.tensor(sizes, dtype=torch.int64, device=dev),
                    blockmap=torch.tensor(blocks, dtype=torch.int32, device=dev),
                    hyper=torch.tensor([group['lr'], b1, b2, group['eps'], group['weight_decay'], 1.0, 1.0, group['grad_scale']],
                                       dtype=torch.float32, device=dev),
                    step=torch.full((1,), step0, dtype=torch.int32, device=dev),
                    host=(group['lr'], group['weight_decay'], group['grad_scale']))
        self._plans[gi] = plan
        return plan

    @torch.no_grad()
    def prepare(self):
        """Host side of a step, safe OUTSIDE a graph capture only: launch plans exist (device pointer tables are uploaded when a
        tensor was replaced) and the device copy of lr / weight_decay / grad_scale is current (lr scheduler, train.py:98,154).
        Returns [(plan, params)] for `enqueue`."""
        work = []
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group['params'] if p.grad is not None]
            if not params:
                continue
            _lib.require_gpu(*params)
            plan = self._plan(gi, group, params)
            host = (group['lr'], group['weight_decay'], group['grad_scale'])
            if host != plan['host']:
                plan['hyper'][0] = group['lr']; plan['hyper'][4] = group['weight_decay']; plan['hyper'][7] = group['grad_scale']
                plan['host'] = host
            work.append((plan, params))
        return work

    def refresh(self, work):
        """The per-step part of `prepare` for a caller that holds `work` across steps (a captured graph): hyper-parameters only.
        Raises when the plan it was given is no longer the optimizer's (state replaced: load_state_dict / invalidate)."""
        for plan, _ in work:
            if self._plans.get(plan['gi']) is not plan:
                raise RuntimeError('FusedAdam: the launch plan held by a captured step is stale (optimizer state was replaced)')
            group = self.param_groups[plan['gi']]
            host = (group['lr'], group['weight_decay'], group['grad_scale'])
            if host != plan['host']:
                plan['hyper'][0] = group['lr']; plan['hyper'][4] = group['weight_decay']; plan['hyper'][7] = group['grad_scale']
                plan['host'] = host

    @staticmethod
    def enqueue(work):
        """Device side: two launches per group (step counter / bias corrections, then the update of every tensor); nothing else,
        so a HIP graph may hold it (graph_step.py)."""
        lib = _lib.load()
        st = _lib.stream()
        for plan, _ in work:
            _lib.check(lib.ssdhip_adam_tick(_lib.ptr(plan['step']), _lib.ptr(plan['hyper']), st), 'adam_tick')
            _lib.check(lib.ssdhip_adam_multi_tensor(_lib.ptr(plan['table']), _lib.ptr(plan['sizes']), _lib.ptr(plan['blockmap']),
                                                    plan['n_blocks'], _CHUNK, _lib.ptr(plan['hyper']), st), 'adam')

    def committed(self, work):
        """Host bookkeeping after `enqueue` ran (or after the replay of a graph that holds it)."""
        for _, params in work:
            for p in params:
                self.state[p]['step'] += 1
            # the kernel wrote through raw pointers: tell torch (and every cache keyed on `_version`, e.g. the engine's packed and
            # Winograd-domain weight images) that the parameters changed, with or without an on_step callback
            torch.autograd.graph.increment_version(params)
        if self._on_step is not None:
            self._on_step()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work = self.prepare()
        self.enqueue(work)
        self.committed(work)
        return loss
