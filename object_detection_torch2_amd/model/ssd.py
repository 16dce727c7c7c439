"""`SSD` with the reference's module API (src/model/ssd.py:8-328), executed by hand-written HIP kernels on MI355X.

Kept from the reference: constructor signature; `.num_classes`, `.default_bboxes` (CPU fp32 (P,4)),
`.normalize`, `.features` / `.detectors` ModuleDicts with the same keys, parameter shapes and state_dict layout (173
entries, OIHW fp32, so weights.pth files interchange); `.forward`, `.loss`, `.train_params`.

Deliberate deviations, all recorded in DESIGN.md:
  * forward implements the documented contract (N, P, 4+C) (ssd.py:93); the literal code never fires a detector
    (ssd.py:102 tests a features key against detector keys) and returns (N, 0, 4+C);
  * no torch.hub fetch (vgg16.py:68): without a local VGG16 weights file the frozen backbone is kaiming-initialised;
  * tensors must live on the GPU: there is no CPU fallback (RuntimeError), the CPU restatement is test-only (oracle/).
"""
import pathlib

import torch
import torch.nn as nn

from .. import _lib, torch_ops
from . import topology as T
from .engine import Engine


class _SSDForward(torch.autograd.Function):
    """outputs = net(x).  Inputs: x and the trainable parameters (so autograd routes their gradients);
    backward runs the hand-written tail backward and returns views of the flat gradient buffer."""

    @staticmethod
    def forward(ctx, net, x, *params):
        ctx.net = net
        ctx.token = object()
        net._engine_token = ctx.token
        ev, net._ready_event = net._ready_event, None
        return net._engine().forward(x, save=True, ready_event=ev)

    @staticmethod
    def backward(ctx, grad_out):
        net = ctx.net
        if net._engine_token is not ctx.token:
            raise RuntimeError('SSD backward: a newer forward replaced the saved activations of this graph '
                               '(one live training graph per module)')
        eng = net._engine()
        # p.grad may alias the flat buffer from the previous step (autograd steals the views); keep "+=" semantics
        flat = eng._flat['buf'] if eng._flat is not None else None
        alias = flat is not None and any(p.grad is not None and flat.data_ptr() <= p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4
                                         for _, p in net.named_train_params())
        old = flat.clone() if alias else None
        grads, _ = eng.backward(grad_out)
        if alias:
            new = eng._flat['buf'].clone()
            eng._flat['buf'].copy_(old)
            names, params, offs, _ = eng.grad_layout()
            grads = [new[o:o + q.numel()].view(q.shape) for q, o in zip(params, offs)]
        return (None, None) + tuple(grads)


class _MultiBoxLoss(torch.autograd.Function):
    """SSD.loss (ssd.py:181-229) on torch.ops.ssdhip.multibox_loss_fwd / _bwd (csrc/torch_ops.cpp -> ssdhip_multibox_loss_*)."""

    @staticmethod
    def forward(ctx, outputs, targets, dboxes, alpha, threshold):
        loss, terms, sel = torch_ops.load().multibox_loss_fwd(outputs, targets, dboxes, float(alpha), float(threshold))
        ctx.save_for_backward(outputs, targets, dboxes, terms, sel)
        ctx.alpha, ctx.threshold = float(alpha), float(threshold)
        return loss[0]

    @staticmethod
    def backward(ctx, grad_loss):
        outputs, targets, dboxes, terms, sel = ctx.saved_tensors
        g = _lib.f32c(grad_loss).reshape(1)
        grad = torch_ops.load().multibox_loss_bwd(outputs, targets, dboxes, ctx.alpha, ctx.threshold, terms, sel, g)
        return grad, None, None, None, None


class SSD(nn.Module):
    def __init__(self, num_classes: int, weights_path: pathlib.Path = None, weights_path_vgg16: pathlib.Path = None, imsize: int = 300):
        """Same arguments as the reference (ssd.py:9).  weights_path_vgg16: local VGG16 state-dict (the reference's
        vgg16.py:64-66 branch); its `features.N.*` entries initialise the frozen backbone.
        imsize (extension, default = the reference's 300): 512 builds the 24,564-prior SSD512 variant (topology.SSD512)."""
        super(SSD, self).__init__()
        self.num_classes = num_classes
        self.topo = T.topology_for(imsize)
        self._hip_epoch = 0
        self._eng = None
        self._tune_all = False
        self._ready_event = None
        self.compute_dtype = 'fp32'      # 'bf16': see set_compute_dtype
        self._engine_token = None
        self._dist_hook = None           # parallel.attach: gradient all-reduce hook, handed to every engine this module builds
        self._default_bboxes = None      # generated lazily by the HIP kernel (first access needs the GPU)

        features = nn.ModuleDict()
        for l in T.feature_plan(self.topo):
            if l['kind'] == 'conv':
                m = nn.Conv2d(in_channels=l['cin'], out_channels=l['cout'], kernel_size=l['k'], stride=l['stride'], padding=l['pad'])
            elif l['kind'] == 'bn':
                m = nn.BatchNorm2d(l['c'])
            elif l['kind'] == 'act':
                m = nn.ReLU(inplace=True)
            else:
                m = nn.MaxPool2d(kernel_size=2, stride=2, padding=l['pad'])
            if l.get('frozen'):
                for p in m.parameters():
                    p.requires_grad = False          # ssd.py:31-32
            features[l['name']] = m
        self.features = features
        self.detectors = nn.ModuleDict({
            h['name']: nn.Conv2d(in_channels=h['cin'], out_channels=h['cout'], kernel_size=3, padding=1) for h in T.head_plan(num_classes, self.topo)})

        self._initialize_weights()
        if weights_path_vgg16 and pathlib.Path(weights_path_vgg16).exists():
            self._load_vgg16(pathlib.Path(weights_path_vgg16))
        if weights_path and pathlib.Path(weights_path).exists():
            print('weights loaded.')
            self.load_state_dict(torch.load(pathlib.Path(weights_path).as_posix(), weights_only=True))

    # ---- construction helpers ------------------------------------------------------------------------------
    def _initialize_weights(self):
        """ssd.py:135-158 for extras + detectors.  The frozen VGG part gets the same kaiming init here because the
        reference's pretrained fetch (vgg16.py:68) is unavailable offline; load real weights via weights_path*."""
        for m in list(self.features.values()) + list(self.detectors.values()):
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _load_vgg16(self, path):
        """Map a reference VGG16 state-dict (`features.<idx>.*`, vgg16.py:25-39) onto conv_L_i / bn_L_i."""
        sd = torch.load(path.as_posix(), weights_only=True)
        own = self.state_dict()
        idx, stage, sub = 0, 1, 1
        with torch.no_grad():
            for v in self.topo.vgg_cfg:
                if isinstance(v, str):
                    idx, stage, sub = idx + 1, stage + 1, 1
                    continue
                for suffix in ('weight', 'bias'):
                    own[f'features.conv_{stage}_{sub}.{suffix}'].copy_(sd[f'features.{idx}.{suffix}'])
                for suffix in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
                    own[f'features.bn_{stage}_{sub}.{suffix}'].copy_(sd[f'features.{idx + 1}.{suffix}'])
                idx, sub = idx + 3, sub + 1
        print('weights loaded.')

    @property
    def default_bboxes(self) -> torch.Tensor:
        if self._default_bboxes is None:
            self._default_bboxes = self._get_default_bboxes()
        return self._default_bboxes

    def _get_default_bboxes(self) -> torch.Tensor:
        """(P, 4) CPU fp32 default boxes (ssd.py:108-133), generated by the HIP kernel when a GPU is present.
        Callers move it with `.to(device)` like the reference (train.py:81)."""
        if not torch.cuda.is_available():
            raise RuntimeError('SSD needs an MI355X: default boxes are generated by libssdhip.so (no CPU fallback)')
        return self.default_bboxes_on(torch.device('cuda', torch.cuda.current_device()), self.topo).cpu()

    @staticmethod
    def default_bboxes_on(device, topo=T.SSD300) -> torch.Tensor:
        maps = [int(v) for m in topo.box_maps for v in m]
        return torch_ops.load().default_boxes(maps, float(topo.s_min), float(topo.s_max), torch.device(device))

    def normalize(self, x: torch.Tensor) -> torch.Tensor:
        """vgg16.py:103-115 as a stand-alone op (the forward fuses it with the NHWC conversion)."""
        _lib.require_gpu(x)
        y = torch_ops.load().normalize_nchw_to_nhwc4(_lib.f32c(x))
        return y[..., :3].permute(0, 3, 1, 2)

    # ---- reference API -------------------------------------------------------------------------------------
    def _engine(self) -> Engine:
        if self._eng is None:
            self._eng = Engine(self)
            self._eng.dist_hook = self._dist_hook      # survives a rebuilt launch plan (finetune_backbone after parallel.attach)
        return self._eng

    def hip_weights_changed(self):
        """Tell the engine that trainable weights were updated outside torch's version counter (FusedAdam)."""
        self._hip_epoch += 1

    def forward(self, x: torch.Tensor, ready_event: torch.cuda.Event = None) -> torch.Tensor:
        """(N, 3, 300, 300) float[0,1] -> (N, P, 4 + C) on x.device (ssd.py:86-106, documented contract).
        ready_event (extension, optional): an event after which `x` is complete.  The frozen backbone runs on its own stream; given
        this event it starts as soon as x is ready instead of queueing behind everything on the caller's stream -- the
        backbone of step i+1 then overlaps the tail backward and optimizer of step i (same results, see engine.py)."""
        _lib.require_gpu(x)
        params = [p for _, p in self.named_train_params()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            self._ready_event = ready_event
            return _SSDForward.apply(self, x, *params)
        return self._engine().forward(x, save=False, ready_event=ready_event)

    def graphed_forward(self, example: torch.Tensor):
        """Extension: capture `forward` for inputs shaped like `example` in a HIP graph (inference, no autograd) and return
        `run(x) -> outputs`: ≈90 launches with their cross-stream events become one.  Measured (scripts/graph_latency.py): no
        throughput gain at batch 2-8 -- eager launches are asynchronous and the device time (1.8 ms at batch 2) exceeds the
        host's enqueue time -- so this is for callers whose host thread is the bottleneck.  `outputs` is the graph's static output tensor (overwritten by the next call); parameters are read at replay
        time, so weight updates made with stock torch ops are seen, but call this again after `load_state_dict` or an optimizer
        step that repacks weights (`hip_weights_changed`)."""
        _lib.require_gpu(example)
        static_in = _lib.f32c(example).clone()
        eng = self._engine()
        with torch.no_grad():
            warm = torch.cuda.Stream(device=example.device)
            warm.wait_stream(torch.cuda.current_stream(example.device))
            with torch.cuda.stream(warm):
                for _ in range(2):                   # lazy buffers, packed weights, kernel attributes
                    eng.forward(static_in, save=False)
            torch.cuda.current_stream(example.device).wait_stream(warm)
            torch.cuda.synchronize(example.device)
            eng._readers.clear()                     # events of the eager warm-up must not be waited on inside the capture
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = eng.forward(static_in, save=False)
            eng._readers.clear()

        def run(x: torch.Tensor) -> torch.Tensor:
            if tuple(x.shape) != tuple(static_in.shape):
                raise ValueError(f'graphed_forward was captured for {tuple(static_in.shape)}, got {tuple(x.shape)}')
            static_in.copy_(x)
            graph.replay()
            return static_out

        run.graph = graph
        return run

    def finetune_backbone(self, enable: bool = True):
        """Extension: un-freeze the VGG16-bn backbone.  The reference never does (ssd.py:31-32; 72.2 -> 187.9
        GFLOP per image); train_params() then yields every feature parameter, in module order, before the detectors."""
        self._tune_all = bool(enable)
        frozen = True
        for name, layer in self.features.items():
            if name == 'conv_6_1':
                frozen = False
            if frozen:
                for p in layer.parameters():
                    p.requires_grad = self._tune_all
        self._eng = None            # new launch plan: frozen flags, gradient layout, backward depth
        return self

    def set_compute_dtype(self, dtype: str):
        """'fp32' (default; exact-fp32 MFMA, Winograd for the deep 3x3 layers) or 'bf16' (BASELINE configs[3] / [4]: the frozen
        backbone keeps bf16 activations in HBM and runs on the bf16 patch convolution; the trainable tail's forward convolutions,
        data and weight gradients round their operands to bf16 while staging; every product accumulates in fp32; parameters,
        gradients, BN statistics, loss and Adam stay fp32 = the "fp32 master" side)."""
        if dtype not in ('fp32', 'bf16'):
            raise ValueError(f"compute dtype must be 'fp32' or 'bf16', got {dtype!r}")
        self.compute_dtype = dtype
        return self

    def named_train_params(self):
        is_vgg = not self._tune_all
        for name, layer in self.features.items():
            if name == 'conv_6_1':
                is_vgg = False
            if is_vgg:
                continue
            for pn, param in layer.named_parameters():
                yield f'features.{name}.{pn}', param
        for name, layer in self.detectors.items():
            for pn, param in layer.named_parameters():
                yield f'detectors.{name}.{pn}', param

    def train_params(self):
        """ssd.py:160-179: features from conv_6_1 on, then the detectors."""
        for _, p in self.named_train_params():
            yield p

    def loss(self, outputs: torch.Tensor, targets: torch.Tensor, default_bboxes: torch.Tensor, a: int = 1) -> torch.Tensor:
        """MultiBox loss (ssd.py:181-229): IoU>0.25 multi-match, SmoothL1 + softmax CE, 3:1 hard-negative mining by the
        (k+1)-th largest value, per-image 1/pos normalisation, batch mean.  0-dim tensor with grad."""
        _lib.require_gpu(outputs, targets, default_bboxes)
        if outputs.dim() != 3 or targets.dim() != 3 or outputs.shape[0] != targets.shape[0] or outputs.shape[2] != targets.shape[2]:
            raise ValueError(f'loss: outputs {tuple(outputs.shape)} and targets {tuple(targets.shape)} do not match (N, *, 4 + C)')
        if tuple(default_bboxes.shape) != (outputs.shape[1], 4):
            raise ValueError(f'loss: default_bboxes must be ({outputs.shape[1]}, 4), got {tuple(default_bboxes.shape)}')
        out = outputs if (outputs.is_contiguous() and outputs.dtype == torch.float32) else _lib.f32c(outputs)
        return _MultiBoxLoss.apply(out, _lib.f32c(targets), _lib.f32c(default_bboxes), a, T.MATCH_THRESHOLD)

    def _match(self, gt: torch.Tensor, df: torch.Tensor, threshold: float = 0.25) -> torch.Tensor:
        """(N,G,>=4),(P,4) -> (N,P,G) bool match mask (ssd.py:231-250), bit-exact."""
        _lib.require_gpu(gt, df)
        gt, df = _lib.f32c(gt), _lib.f32c(df)
        N, G, row = gt.shape
        P = df.shape[0]
        mask = torch.empty(N, P, G, device=gt.device, dtype=torch.uint8)
        if G > 0:
            _lib.check(_lib.load().ssdhip_match_mask(_lib.ptr(gt), _lib.ptr(df), N, P, G, row, float(threshold), _lib.ptr(mask),
                                                     _lib.stream()), '_match')
        return mask.bool()
