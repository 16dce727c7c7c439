"""`VGG16` with the reference's module API (src/model/vgg16.py:7-130) on the MI355X path: SURVEY 8(f)-4, the classification
pre-training that produces `result/classification/weights.pth` for `SSD(weights_path_vgg16=...)` (train.py:69-74, ssd.py:25).

Kept from the reference: constructor `(weights_path, num_classes, transfer_learning)`, `.features` as the same `nn.Sequential`
(state_dict keys `features.<idx>.*`), `.classifier` (1000-way) and `.classifier2` (num_classes-way) stacks, `forward`, `normalize`,
`loss`, `_initialize_weights`; weights files interchange with the reference's.

The reference cannot run this path as shipped (recorded in DESIGN.md §6); what this module does about each point:
  * `train.py:97` calls `net.train_params()`, which the reference's VGG16 lacks -> provided (every parameter that requires grad,
    of the stack `forward` uses);
  * no `torch.hub` fetch (vgg16.py:68) -> without a weights file the feature stack is kaiming-initialised;
  * `Linear(512 * 7 * 7, ...)` (vgg16.py:44) needs a 7x7 final map, which these pools give for 184..215-px inputs (not for the
    scripts' default 300) -> any other size raises the same kind of error torch would (ValueError here);
  * `VGG16.loss` (vgg16.py:127-129) is the mean of sum(targets * log_softmax) WITHOUT the minus sign -> `loss(outputs, targets)`
    is that, literally (parity); `loss(..., nll=True)` is the negative log-likelihood that training needs (this package's
    train.py uses it and says so).
Everything runs on libssdhip.so: the feature stack on the SSD engine (un-frozen mode: conv / BN / ReLU / pool forward and
backward), the Linear layers on the implicit-GEMM kernels as 1x1 convolutions, ReLU + Dropout and the loss as small HIP passes.
No CPU fallback.
"""
import ctypes
import pathlib

import torch
import torch.nn as nn

from .. import _lib
from . import topology as T
from .engine import Engine

_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M_P', 512, 512, 512, 'M', 512, 512, 512, 'M_P']


def _rup(v, m):
    return (v + m - 1) // m * m


def _hash_key(seed, stream):
    return (seed * 0x9E3779B1 + stream * 0x85EBCA77 + 0x165667B1) & 0xFFFFFFFF


class _Linear:
    """One nn.Linear on the implicit-GEMM kernels: rows = batch, 1x1 'pixels', K = in_features (multiple of 32)."""

    def __init__(self, mod):
        self.mod = mod
        self.cout, self.cin = mod.weight.shape
        self.co_pad = _rup(self.cout, 32)
        self.packed = self.packed_d = None
        self.key = self.key_d = None

    def _pack(self, lib, dgrad):
        w = self.mod.weight
        key = (w.data_ptr(), w._version)
        if not dgrad:
            rows, KP = _rup(self.cout, 128), _rup(self.cin, 32)
            if self.packed is None or self.packed.device != w.device:
                self.packed, self.key = torch.empty(rows * KP, device=w.device, dtype=torch.float32), None
            if self.key != key:
                _lib.check(lib.ssdhip_pack_conv_weight(_lib.ptr(w), _lib.ptr(self.packed), self.cout, self.cin, 1, 1, rows, self.cin, KP, 0,
                                                       _lib.stream()), 'pack linear weight')
                self.key = key
            return self.packed, KP
        rows, KPd = _rup(self.cin, 128), self.co_pad
        if self.packed_d is None or self.packed_d.device != w.device:
            self.packed_d, self.key_d = torch.empty(rows * KPd, device=w.device, dtype=torch.float32), None
        if self.key_d != key:
            _lib.check(lib.ssdhip_pack_conv_weight(_lib.ptr(w), _lib.ptr(self.packed_d), self.cout, self.cin, 1, 1, rows, self.co_pad, KPd, 1,
                                                   _lib.stream()), 'pack linear weight (dgrad)')
            self.key_d = key
        return self.packed_d, KPd

    def forward(self, lib, x):           # x (N, cin) -> (N, cout)
        N = x.shape[0]
        packed, KP = self._pack(lib, False)
        y = torch.empty(N, self.cout, device=x.device, dtype=torch.float32)
        p = _lib.ConvParams()
        p.x, p.w, p.bias, p.y = x.data_ptr(), packed.data_ptr(), self.mod.bias.data_ptr(), y.data_ptr()
        p.y_batch_stride = self.cout
        p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = N, 1, 1, self.cin, 1, 1, self.cout
        p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = 1, 1, 1, 0, KP, 0, 0, 0
        _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), _lib.stream()), 'linear forward')
        return y

    def backward(self, lib, x, dy, need_dx, ws_holder):     # dy (N, cout) -> dW, db (, dx (N, cin))
        N = x.shape[0]
        dyp = dy
        if self.co_pad != self.cout:                         # the gradient kernels take rows padded to 32 channels
            dyp = torch.zeros(N, self.co_pad, device=dy.device, dtype=torch.float32)
            dyp[:, :self.cout] = dy
        dw, db = torch.empty_like(self.mod.weight), torch.empty_like(self.mod.bias)
        q = _lib.WgradParams()
        q.N, q.Hi, q.Wi, q.Cin, q.Ho, q.Wo, q.Cout = N, 1, 1, self.cin, 1, 1, self.cout
        q.R, q.S, q.stride, q.pad, q.dy_row, q.dy_batch_stride, q.dw_cin = 1, 1, 1, 0, self.co_pad, self.co_pad, 0
        q.x, q.dy, q.dw_oihw, q.dbias = x.data_ptr(), dyp.data_ptr(), dw.data_ptr(), db.data_ptr()
        need = _lib.check(lib.ssdhip_conv2d_wgrad_ws_bytes(ctypes.byref(q)), 'wgrad_ws_bytes')
        if ws_holder.get('ws') is None or ws_holder['ws'].numel() * 4 < need:
            ws_holder['ws'] = torch.empty((need + 3) // 4, device=x.device, dtype=torch.float32)
        q.ws = ws_holder['ws'].data_ptr()
        _lib.check(lib.ssdhip_conv2d_wgrad(ctypes.byref(q), _lib.stream()), 'linear weight gradient')
        dx = None
        if need_dx:
            packed_d, KPd = self._pack(lib, True)
            dx = torch.empty(N, self.cin, device=x.device, dtype=torch.float32)
            p = _lib.ConvParams()
            p.x, p.w, p.y, p.y_batch_stride = dyp.data_ptr(), packed_d.data_ptr(), dx.data_ptr(), self.cin
            p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = N, 1, 1, self.co_pad, 1, 1, self.cin
            p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = 1, 1, 1, 0, KPd, 1, 0, 0
            _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), _lib.stream()), 'linear data gradient')
        return dw, db, dx


class _VGGForward(torch.autograd.Function):
    """logits = net(x).  Inputs: x and the parameters that require grad (so autograd routes their gradients)."""

    @staticmethod
    def forward(ctx, net, x, *params):
        ctx.net = net
        ctx.token = object()
        net._token = ctx.token
        return net._run(x, save=True)

    @staticmethod
    def backward(ctx, grad_out):
        net = ctx.net
        if net._token is not ctx.token:
            raise RuntimeError('VGG16 backward: a newer forward replaced the saved activations of this graph')
        grads = net._run_backward(grad_out)
        return (None, None) + tuple(grads)


class _DotLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, targets, sign):
        lib = _lib.load()
        N, C = outputs.shape
        rowval = torch.empty(N, device=outputs.device, dtype=torch.float32)
        loss = torch.empty(1, device=outputs.device, dtype=torch.float32)
        _lib.check(lib.ssdhip_log_softmax_dot_fwd(_lib.ptr(outputs), _lib.ptr(targets), N, C, float(sign), _lib.ptr(rowval), _lib.ptr(loss),
                                                  _lib.stream()), 'VGG16.loss')
        ctx.save_for_backward(outputs, targets)
        ctx.sign = float(sign)
        return loss[0]

    @staticmethod
    def backward(ctx, grad_loss):
        outputs, targets = ctx.saved_tensors
        N, C = outputs.shape
        g = _lib.f32c(grad_loss).reshape(1)
        scratch = torch.empty(1, device=outputs.device, dtype=torch.float32)
        dz = torch.empty_like(outputs)
        _lib.check(_lib.load().ssdhip_log_softmax_dot_bwd(_lib.ptr(outputs), _lib.ptr(targets), N, C, ctx.sign, _lib.ptr(g), _lib.ptr(scratch),
                                                          _lib.ptr(dz), _lib.stream()), 'VGG16.loss backward')
        return dz, None, None


class VGG16(nn.Module):
    def __init__(self, weights_path: pathlib.Path = None, num_classes: int = 20, transfer_learning: bool = False):
        """Same arguments as the reference (vgg16.py:8)."""
        super(VGG16, self).__init__()
        self.transfer_learning = transfer_learning
        self.num_classes = num_classes
        self.mean = torch.Tensor([0.485, 0.456, 0.406])
        self.std = torch.Tensor([0.229, 0.224, 0.225])
        layers, in_channels = [], 3
        for v in _CFG:                                                       # vgg16.py:22-39
            if v == 'M':
                layers += [nn.MaxPool2d(kernel_size=2, stride=2)]
            elif v == 'M_P':
                layers += [nn.MaxPool2d(kernel_size=2, stride=2, padding=1)]
            else:
                layers += [nn.Conv2d(in_channels=in_channels, out_channels=v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                in_channels = v
        self.features = nn.Sequential(*layers)

        def stack(n_out):                                                    # vgg16.py:42-61
            return nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(inplace=True), nn.Dropout(),
                                 nn.Linear(4096, 4096), nn.ReLU(inplace=True), nn.Dropout(), nn.Linear(4096, n_out))
        self.classifier = stack(1000)
        self.classifier2 = stack(num_classes)

        # engine plumbing (same contract as model.SSD)
        self.topo = T.VGG16_FEATURES if not transfer_learning else T.VGG16_FEATURES_FROZEN
        self.detectors = nn.ModuleDict()
        self.compute_dtype = 'fp32'
        self._hip_epoch = 0
        self._tune_all = not transfer_learning
        self._dist_hook = None
        self._eng = None
        self._token = None
        self._saved = None
        self._lin = None
        self._ws = {}
        self._features_trainable = not transfer_learning
        self.dropout_seed = 0            # the dropout masks of forward call k use the counter-hash streams (seed, 2k), (seed, 2k + 1)
        self._calls = 0

        if weights_path and pathlib.Path(weights_path).exists():
            print('weights loaded.')
            self.load_state_dict(torch.load(pathlib.Path(weights_path).as_posix(), weights_only=True))
        else:
            # the reference fetches torchvision's pretrained vgg16_bn through torch.hub here (vgg16.py:68): unavailable offline
            for m in self.features:
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                    nn.init.constant_(m.bias, 0)
            self._initialize_weights()
        if self.transfer_learning:                                           # vgg16.py:72-75
            for p in self.features.parameters():
                p.requires_grad = False

    def _initialize_weights(self):
        for m in self.classifier2.modules():                                 # vgg16.py:77-83
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    # ---- engine contract -----------------------------------------------------------------------------------
    def feature_lookup(self):
        """conv_L_i / bn_L_i -> module of the reference's Sequential (the engine's layer names)."""
        out, idx, stage, sub = {}, 0, 1, 1
        for v in _CFG:
            if isinstance(v, str):
                idx, stage, sub = idx + 1, stage + 1, 1
                continue
            out[f'conv_{stage}_{sub}'], out[f'bn_{stage}_{sub}'] = self.features[idx], self.features[idx + 1]
            idx, sub = idx + 3, sub + 1
        return out

    def named_train_params(self):
        """Engine names of the trainable feature parameters (all of them unless transfer_learning), module order."""
        if self.transfer_learning and not self._features_trainable:
            return
        for name, layer in self.feature_lookup().items():
            for pn, param in layer.named_parameters():
                yield f'features.{name}.{pn}', param

    def hip_weights_changed(self):
        self._hip_epoch += 1

    def _engine(self) -> Engine:
        if self._eng is None:
            self._eng = Engine(self)
            self._eng.dist_hook = self._dist_hook
        return self._eng

    def _head(self):
        return self.classifier2 if self.transfer_learning else self.classifier

    def unfreeze_features(self):
        """Extension: keep the num_classes-way stack of transfer_learning=True (vgg16.py:96-99) but train the feature stack too.
        The reference offers only 'classifier (1000-way) + trainable features' or 'classifier2 + frozen features'; with 20-class
        VOC targets (dataset.py:47) the first cannot even compute its loss ((N,20) * (N,1000)), the second pre-trains nothing of the
        backbone.  This package's train.py --purpose classification uses classifier2 + trainable features."""
        if not self.transfer_learning:
            raise RuntimeError('unfreeze_features() is for transfer_learning=True models (the other mode trains the features already)')
        for p in self.features.parameters():
            p.requires_grad = True
        self._features_trainable = True
        self.topo, self._tune_all, self._eng = T.VGG16_FEATURES, True, None
        return self

    def train_params(self):
        """Not in the reference (train.py:97 needs it): the trainable parameters of the stack forward() uses."""
        for p in list(self.features.parameters()) + list(self._head().parameters()):
            if p.requires_grad:
                yield p

    # ---- forward / backward on the HIP path ----------------------------------------------------------------
    def normalize(self, x: torch.Tensor) -> torch.Tensor:
        """vgg16.py:103-115 as a stand-alone op (the forward fuses it with the NHWC conversion)."""
        from .ssd import SSD
        return SSD.normalize(self, x)

    def _run(self, x, save):
        lib = _lib.load()
        eng = self._engine()
        fmap = eng.forward(x, save=save)                                     # (N, h, w, 512) NHWC, after the last pool
        N, h, w, C = fmap.shape
        if (h, w, C) != (7, 7, 512):
            raise ValueError(f'VGG16: the classifier takes a 512 x 7 x 7 feature map (vgg16.py:44), this input gives {C} x {h} x {w}; '
                             f'use an input size of 184..215 px')
        head = self._head()
        if self._lin is None or self._lin[0] is not head:
            self._lin = (head, [_Linear(head[0]), _Linear(head[3]), _Linear(head[6])])
        lins = self._lin[1]
        flat = torch.empty(N, C * h * w, device=fmap.device, dtype=torch.float32)
        _lib.check(lib.ssdhip_nhwc_flatten(_lib.ptr(fmap), _lib.ptr(flat), N, h, w, C, 0, _lib.stream()), 'flatten')   # x.view(N, -1)
        training = self.training
        keys = (_hash_key(self.dropout_seed, 2 * self._calls), _hash_key(self.dropout_seed, 2 * self._calls + 1))
        self._calls += 1
        z1 = lins[0].forward(lib, flat)
        a1 = torch.empty_like(z1)
        _lib.check(lib.ssdhip_relu_dropout(_lib.ptr(z1), None, _lib.ptr(a1), z1.numel(), head[2].p, keys[0], int(training), _lib.stream()), 'relu+dropout')
        z2 = lins[1].forward(lib, a1)
        a2 = torch.empty_like(z2)
        _lib.check(lib.ssdhip_relu_dropout(_lib.ptr(z2), None, _lib.ptr(a2), z2.numel(), head[5].p, keys[1], int(training), _lib.stream()), 'relu+dropout')
        out = lins[2].forward(lib, a2)
        self._saved = dict(flat=flat, z1=z1, a1=a1, z2=z2, a2=a2, keys=keys, training=training, shape=(N, h, w, C)) if save else None
        return out

    def _run_backward(self, grad_out):
        lib = _lib.load()
        sv, self._saved = self._saved, None
        if sv is None:
            raise RuntimeError('VGG16 backward called without a saved forward')
        head, lins = self._lin
        g = _lib.f32c(grad_out)
        # data parallelism: the classifier stack's gradients are separate tensors (fc1 alone is 411 MB); each goes into the same
        # asynchronous all-reduce hook as the feature gradients the moment its weight-gradient launch is enqueued, so the collectives
        # travel while the rest of the backward computes; the engine's finish() (or wait_launched below) waits for all of them
        hook = self._dist_hook
        reduce_ = hook.launch if hook is not None else (lambda t: None)
        dw3, db3, da2 = lins[2].backward(lib, sv['a2'], g, True, self._ws)
        reduce_(dw3); reduce_(db3)
        dz2 = torch.empty_like(da2)
        _lib.check(lib.ssdhip_relu_dropout(_lib.ptr(sv['z2']), _lib.ptr(da2), _lib.ptr(dz2), dz2.numel(), head[5].p, sv['keys'][1],
                                           int(sv['training']), _lib.stream()), 'relu+dropout backward')
        dw2, db2, da1 = lins[1].backward(lib, sv['a1'], dz2, True, self._ws)
        reduce_(dw2); reduce_(db2)
        dz1 = torch.empty_like(da1)
        _lib.check(lib.ssdhip_relu_dropout(_lib.ptr(sv['z1']), _lib.ptr(da1), _lib.ptr(dz1), dz1.numel(), head[2].p, sv['keys'][0],
                                           int(sv['training']), _lib.stream()), 'relu+dropout backward')
        dw1, db1, dflat = lins[0].backward(lib, sv['flat'], dz1, self._features_trainable, self._ws)
        reduce_(dw1); reduce_(db1)
        head_grads = {id(head[0].weight): dw1, id(head[0].bias): db1, id(head[3].weight): dw2, id(head[3].bias): db2,
                      id(head[6].weight): dw3, id(head[6].bias): db3}
        feat_grads = {}
        if self._features_trainable:
            N, h, w, C = sv['shape']
            dmap = torch.empty(N, h, w, C, device=dflat.device, dtype=torch.float32)
            _lib.check(lib.ssdhip_nhwc_flatten(_lib.ptr(dflat), _lib.ptr(dmap), N, h, w, C, 1, _lib.stream()), 'flatten backward')
            grads, params = self._engine().backward(dmap)
            feat_grads = {id(p): gr.clone() for p, gr in zip(params, grads)}    # views of the engine's flat buffer: detach from it
        if hook is not None:
            hook.wait_launched()             # frozen features: nobody called finish(); otherwise nothing is left to wait for
        return [head_grads.get(id(p), feat_grads.get(id(p))) for p in self._autograd_params()]

    def _autograd_params(self):
        return [p for p in list(self.features.parameters()) + list(self._head().parameters()) if p.requires_grad]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(N, 3, S, S) float[0,1] -> (N, 1000 or num_classes) (vgg16.py:85-101)."""
        _lib.require_gpu(x)
        params = self._autograd_params()
        if torch.is_grad_enabled() and params:
            return _VGGForward.apply(self, x, *params)
        return self._run(x, save=False)

    def loss(self, outputs: torch.Tensor, targets: torch.Tensor, nll: bool = False) -> torch.Tensor:
        """vgg16.py:117-130, literally: `torch.sum(targets * log_softmax(outputs), dim=1).mean()` -- a log-likelihood, the minus
        sign of a loss is missing in the reference.  nll=True returns its negative (what an optimizer must minimise)."""
        _lib.require_gpu(outputs, targets)
        if outputs.dim() != 2 or tuple(outputs.shape) != tuple(targets.shape):
            raise ValueError(f'loss: outputs {tuple(outputs.shape)} and targets {tuple(targets.shape)} must both be (N, C)')
        out = outputs if (outputs.is_contiguous() and outputs.dtype == torch.float32) else _lib.f32c(outputs)
        return _DotLoss.apply(out, _lib.f32c(targets), -1.0 if nll else 1.0)
