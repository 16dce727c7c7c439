"""Fused multi-tensor Adam on libssdhip.so: same update rule as torch.optim.Adam(lr, weight_decay) used by the
reference harness (train.py:97,122) -- L2 decay added to the gradient, bias-corrected, no amsgrad -- as ONE kernel
launch over all trainable tensors.  Step count / bias corrections live in device memory (graph-replayable)."""
import numpy as np
import torch

from . import _lib

_CHUNK = 65536


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0, on_step=None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError('FusedAdam: invalid hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale))
        self._on_step = on_step
        self._plans = {}

    def _plan(self, gi, group, params):
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in params)
        plan = self._plans.get(gi)
        if plan is not None and plan['key'] == key:
            return plan
        dev = params[0].device
        rows, sizes, blocks = [], [], []
        for ti, p in enumerate(params):
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                raise ValueError('FusedAdam: contiguous fp32 parameters and gradients only')
            st = self.state[p]
            if 'exp_avg' not in st:
                st['step'] = 0
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            rows.append([p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()])
            sizes.append(p.numel())
            blocks += [[ti, c] for c in range((p.numel() + _CHUNK - 1) // _CHUNK)]
        b1, b2 = group['betas']
        old = plan
        plan = dict(key=key, n_blocks=len(blocks),
                    table=torch.tensor(rows, dtype=torch.int64, device=dev),
                    sizes=torch.tensor(sizes, dtype=torch.int64, device=dev),
                    blockmap=torch.tensor(blocks, dtype=torch.int32, device=dev),
                    hyper=old['hyper'] if old else torch.tensor([group['lr'], b1, b2, group['eps'], group['weight_decay'], 1.0, 1.0,
                                                                 group['grad_scale']], dtype=torch.float32, device=dev),
                    # device step counter; resumes from the checkpointed per-parameter step (load_state_dict), like torch.optim.Adam
                    step=old['step'] if old else torch.full((1,), int(max(int(self.state[p].get('step', 0)) for p in params)),
                                                            dtype=torch.int32, device=dev),
                    host=old['host'] if old else (group['lr'], group['weight_decay'], group['grad_scale']))
        self._plans[gi] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group['params'] if p.grad is not None]
            if not params:
                continue
            _lib.require_gpu(*params)
            plan = self._plan(gi, group, params)
            host = (group['lr'], group['weight_decay'], group['grad_scale'])
            if host != plan['host']:            # lr scheduler (train.py:98,154) or user changed a hyper-parameter
                plan['hyper'][0] = group['lr']; plan['hyper'][4] = group['weight_decay']; plan['hyper'][7] = group['grad_scale']
                plan['host'] = host
            st = _lib.stream()
            _lib.check(lib.ssdhip_adam_tick(_lib.ptr(plan['step']), _lib.ptr(plan['hyper']), st), 'adam_tick')
            _lib.check(lib.ssdhip_adam_multi_tensor(_lib.ptr(plan['table']), _lib.ptr(plan['sizes']), _lib.ptr(plan['blockmap']),
                                                    plan['n_blocks'], _CHUNK, _lib.ptr(plan['hyper']), st), 'adam')
            for p in params:
                self.state[p]['step'] += 1
        if self._on_step is not None:
            self._on_step()
        return loss
