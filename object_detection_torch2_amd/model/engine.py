"""Launch plan of the SSD300 forward / tail-backward on libssdhip.so.

Replaces the ATen dispatch behind SSD.forward (model/ssd.py:86-106) and the autograd backward of the trainable tail
(train.py:121).  Design (MI355X-first):
  * NHWC fp32 activations, raw conv outputs `z` are the only materialised tensors; BatchNorm(train)+ReLU of layer L is
    applied while layer L+1 (and its detector head) stage their input tiles, or fused into the max-pool pass;
  * BN batch statistics come from per-block partials written by the conv epilogue (no extra pass over `z`);
  * detector heads write straight into the (N, P, 4+C) output (zero-copy concat, ssd.py:103-104);
  * the VGG part is frozen (ssd.py:31-32): no graph, no saved activations, no dgrad/wgrad for conv_1_1..conv_5_3
    (SSD.finetune_backbone() un-freezes it: the same backward loop then runs down to conv_1_1, through the max-pools);
  * gradients land in one flat fp32 buffer (views per parameter) so data-parallel all-reduce is a few large RCCL calls;
  * three HIP streams: the detector heads (forward) and every weight gradient (backward) run on a side stream next to the
    layer chain, the frozen backbone on its own stream across steps (SSD_SINGLE_STREAM=1 disables both);
  * the deep 3x3 layers (conv_3_2 .. conv_6_1, heads on >= 16-px maps) run through Winograd F(4x4,3x3) (csrc/winograd4.hip),
    conv_1_2 .. conv_3_1 through the one-kernel F(2x2) form (SSD_WINOGRAD=f2: F(2x2) everywhere).
"""
import contextlib
import ctypes
import os

import torch

from .. import _lib
from . import topology as T


def _rup(v, m):
    return (v + m - 1) // m * m


class _ConvSpec:
    def __init__(self, name, cin, cout, k, stride, pad, frozen, is_head=False):
        self.name, self.cin, self.cout, self.k, self.stride, self.pad = name, cin, cout, k, stride, pad
        self.frozen, self.is_head = frozen, is_head
        self.cin_pad = 4 if cin == 3 else cin
        self.KP = _rup(k * k * self.cin_pad, 32)
        self.rows_pad = _rup(cout, 128)
        # dgrad operand: rows = ci, k = tap * co_pad + co
        self.co_pad = _rup(cout, 32)
        self.KPd = k * k * self.co_pad
        self.rows_pad_d = _rup(cin, 128)
        self.packed = None
        self.packed_key = None
        self.packed_d = None
        self.packed_d_key = None
        # Winograd F(2x2,3x3) forward for the deep 3x3/s1/p1 layers (conv_3_1 .. conv_6_1): measured x1.1 .. x1.8 over the
        # direct kernel (scripts/wino_bench.py); the wide, shallow layers (conv_1_x, conv_2_x) lose to the transform traffic
        self.wino = (not is_head and k == 3 and stride == 1 and pad == 1 and cin % 32 == 0 and cin >= 128
                     and cout % 4 == 0 and 256 % (cout // 4) == 0)
        # detector heads on >= 10x10 maps with wide inputs also run forward through Winograd; their input transform V is kept
        # and reused by the Winograd weight gradient in backward
        self.wino_head = is_head and k == 3 and stride == 1 and pad == 1 and cin >= 256
        self.wino_u = None
        self.wino_key = None
        self.wino4_u = None      # F(4x4,3x3) weight planes (frozen backbone layers, forward only)
        self.wino4_key = None


class _Stage:
    """conv + bn (+ pool) (+ detector head) of the feature chain."""

    def __init__(self, conv, bn_name, pool_pad, head, head_index):
        self.conv, self.bn_name, self.pool_pad, self.head, self.head_index = conv, bn_name, pool_pad, head, head_index


class Engine:
    def __init__(self, module):
        self.m = module                      # the SSD nn.Module (parameters / buffers live there)
        self.lib = _lib.load()
        self.nc = module.num_classes
        self.row = self.nc + 4
        # name -> layer module: SSD keeps the reference's ModuleDicts; the VGG16 classifier keeps the reference's nn.Sequential
        # (state_dict keys features.<index>.*) and hands out a name lookup
        self._feat = module.feature_lookup() if hasattr(module, 'feature_lookup') else module.features
        plan = T.feature_plan(module.topo)
        heads = {h['tag']: (i, h) for i, h in enumerate(T.head_plan(self.nc, module.topo))}
        self.stages = []
        i = 0
        while i < len(plan):
            l = plan[i]
            assert l['kind'] == 'conv'
            conv = _ConvSpec(l['name'], l['cin'], l['cout'], l['k'], l['stride'], l['pad'], l['frozen'] and not module._tune_all)
            bn = plan[i + 1]['name']
            i += 3
            pool_pad = None
            if i < len(plan) and plan[i]['kind'] == 'pool':
                pool_pad = plan[i]['pad']
                i += 1
            head, hidx = None, None
            if l['tag'] in heads:
                hidx, h = heads[l['tag']]
                head = _ConvSpec(h['name'], h['cin'], h['cout'], 3, 1, 1, False, is_head=True)
                head.anchors = h['anchors']
            self.stages.append(_Stage(conv, bn, pool_pad, head, hidx))
        self.tail_start = next((i for i, s in enumerate(self.stages) if not s.conv.frozen), len(self.stages))
        self.headless = not any(s.head is not None for s in self.stages)      # VGG16 classifier: forward returns the last activation
        self._bufs = {}
        self._bn = {}          # bn name -> dict(scale, shift, mean, invstd)
        self._saved = None     # forward context consumed by backward
        self._flat = None
        self.dist_hook = None  # callable(flat_grad) for data-parallel all-reduce (parallel.py)
        self._side = {}        # device -> side stream (heads / weight gradients)
        self.single_stream = os.environ.get('SSD_SINGLE_STREAM', '0') == '1'
        self._plane_limit = 1 << 30      # elements one plane-GEMM launch may address (tests lower it to exercise the grouping)
        self._readers = {}     # backbone stage -> events after which its output / BN state may be overwritten by the next
                               # backbone pass: the forward consumers (det_4_3 head, conv_6_1) and their weight gradients
        self.bf16 = False      # set per forward from module.compute_dtype; backward follows the saved forward
        self.prof = None       # list: when set, forward conv launches are bracketed by events (bench.py roofline leg)
        # grid of the persistent fused Winograd kernel: 0 = one block per CU; a block fills its CU (LDS, registers) for the whole
        # launch, so a smaller grid leaves CUs to the tail / weight-gradient streams that run beside the backbone
        # SSD_WF_BLOCKS unset: 224 of the 256 CUs while the backbone runs beside the tail's streams (the small dependent launches of the
        # tail then never queue behind a whole persistent launch: 12.35-12.46 vs 12.56-12.63 ms per step; 240 or 192 are worse), all
        # 256 when it runs alone
        # A/B switches of DESIGN.md section 8, read once (a buffer set caches decisions taken from them)
        self._no_splitk = os.environ.get('SSD_NO_SPLITK') == '1'
        self._no_fused = os.environ.get('SSD_NO_FUSED_WINOGRAD') == '1'
        self._no_patch = os.environ.get('SSD_NO_BF16_PATCH') == '1'
        self._no_plane_queue = os.environ.get('SSD_NO_PLANE_QUEUE') == '1'      # A/B: every plane GEMM on the one-slot-per-block kernel
        # opt-in, never the default: 128 x 128 plane GEMMs as three bf16 terms per fp32 operand and six products on the bf16 MFMA (csrc/gemm_planes.hip:
        # error vs fp64 no larger than the fp32 MFMA's, 1.45x its speed); the default keeps exact fp32 products, the reference's arithmetic
        self._plane_split = os.environ.get('SSD_PLANE_GEMM', 'fp32') == 'bf16x3'
        if os.environ.get('SSD_PLANE_GEMM', 'fp32') not in ('fp32', 'bf16x3'):
            raise ValueError("SSD_PLANE_GEMM must be 'fp32' or 'bf16x3'")
        self._z_bf16 = os.environ.get('SSD_BF16_Z', 'bf16') == 'bf16'
        # frozen 3x3 backbone layers on maps up to SSD_W4_MAX_MAP = 160 px (SSD300: conv_3_2 .. conv_5_3 at 75 .. 19 px; SSD512: from conv_3_2 at 128 px): Winograd F(4x4,3x3) (csrc/winograd4.hip: 1.78x fewer MFMA flops
        # than F(2x2), 2.7x its rounding error, still 8x inside the per-layer bound of the tests); SSD_WINOGRAD=f2 keeps F(2x2) everywhere
        self._wino_mode = os.environ.get('SSD_WINOGRAD', 'f4')
        if self._wino_mode not in ('f2', 'f4'):
            raise ValueError(f"SSD_WINOGRAD must be 'f2' or 'f4', got {self._wino_mode!r}")
        self._pool_fuse = os.environ.get('SSD_NO_POOL_FUSE') != '1'      # A/B: conv + 2x2 max-pool without materialising z (conv_1_2, conv_2_2)
        self._pool_xform = os.environ.get('SSD_POOL_XFORM') == '1'       # A/B: the layer behind a fused pool reads the stored extreme and applies BatchNorm.ReLU on load
        self._pool_pair = os.environ.get('SSD_POOL_PAIR') == '1'         # A/B: the pooled epilogue stores window maxima AND minima (round-3 form) instead of the one the pool uses
        self._w4_max_map = int(os.environ.get('SSD_W4_MAX_MAP', '160'))     # largest map side on the F(4x4) passes (SSD512: conv_3_2 / conv_3_3 at 128 px)
        self._early_f4 = os.environ.get('SSD_EARLY_F4', '0') == '1'       # opt-in: conv_2_2 / conv_3_1 on the two-pass F(4x4) form (csrc/wino4_fused.hip)
        self._w4_min_cin = int(os.environ.get('SSD_W4_MIN_CIN', '256'))      # A/B: 128 also routes conv_3_1 (K = 128) to the F(4x4) passes
        self._cus = None       # compute units of the device (first forward): the persistent backbone grids take 7/8 of them beside the tail
        self._wf_blocks_env = os.environ.get('SSD_WF_BLOCKS')
        self._wf_blocks = int(self._wf_blocks_env) if self._wf_blocks_env is not None else 0
        # opt-in (SSD_PQ_CUS=208): CUs the persistent work-queue plane GEMM fills (ssdhip_conv_params.max_cus); 208 of 256 beside the tail: 9.53 vs 9.57 ms per
        # step -- not the default: the launch the bench line's `roofline` prices would then run on 13/16 of the chip by design
        self._pq_cus = int(os.environ.get('SSD_PQ_CUS', '0'))
        self._pack_table = None     # (signature, device table, largest job) of the batched weight re-layout (_pack_many)
        self._fused_min = int(os.environ.get('SSD_FUSED_MIN', '70'))      # smallest map side that runs on the fused Winograd kernel
        # phased execution (graph_step.py): the frozen backbone and the trainable tail as two separately enqueued (captured) pieces.
        # What the tail reads of the backbone -- z and BN state of the stages with a detector head and of the last frozen stage --
        # exists twice (`parity`), so backbone(i+1) never overwrites what tail(i) still reads and no event crosses between the pieces.
        self._bb_ctx = {}      # parity -> what forward(phase='backbone') left for forward(phase='tail')
        self._phased = False   # True while a phased step runs: no cross-step reader events (the double buffers replace them)

    # ------------------------------------------------------------------------------------------------ helpers
    def _weight(self, spec):
        mod = (self.m.detectors if spec.is_head else self._feat)[spec.name]
        return mod.weight, mod.bias

    def _pack(self, spec, dgrad=False):
        w, _ = self._weight(spec)
        _lib.require_gpu(w)
        key = (w.data_ptr(), w._version, self.m._hip_epoch if not spec.frozen else 0)
        if not dgrad:
            if spec.packed is None or spec.packed.device != w.device:
                spec.packed = torch.empty(spec.rows_pad * spec.KP, device=w.device, dtype=torch.float32)
                spec.packed_key = None
            if spec.packed_key != key:
                _lib.check(self.lib.ssdhip_pack_conv_weight(_lib.ptr(w), _lib.ptr(spec.packed), spec.cout, spec.cin, spec.k, spec.k,
                                                            spec.rows_pad, spec.cin_pad, spec.KP, 0, _lib.stream()), 'pack_conv_weight')
                spec.packed_key = key
            return spec.packed
        if spec.packed_d is None or spec.packed_d.device != w.device:
            spec.packed_d = torch.empty(spec.rows_pad_d * spec.KPd, device=w.device, dtype=torch.float32)
            spec.packed_d_key = None
        if spec.packed_d_key != key:
            _lib.check(self.lib.ssdhip_pack_conv_weight(_lib.ptr(w), _lib.ptr(spec.packed_d), spec.cout, spec.cin, spec.k, spec.k,
                                                        spec.rows_pad_d, spec.co_pad, spec.KPd, 1, _lib.stream()), 'pack_conv_weight')
            spec.packed_d_key = key
        return spec.packed_d

    def _pack_many(self, items):
        """The (spec, dgrad) layouts of _pack whose keys are stale, in ONE launch (ssdhip_pack_conv_weights): an optimizer step changes every
        trainable weight, i.e. ~26 five-microsecond launches per step otherwise.  The job table lives on the device and is rebuilt only when a
        pointer in it changes."""
        stale = []
        for spec, dgrad in items:
            w, _ = self._weight(spec)
            _lib.require_gpu(w)
            key = (w.data_ptr(), w._version, self.m._hip_epoch if not spec.frozen else 0)
            if not dgrad:
                if spec.packed is None or spec.packed.device != w.device:
                    spec.packed = torch.empty(spec.rows_pad * spec.KP, device=w.device, dtype=torch.float32)
                    spec.packed_key = None
                if spec.packed_key != key:
                    stale.append((spec, False, w, spec.packed, (spec.cout, spec.cin, spec.k, spec.k, spec.rows_pad, spec.cin_pad, spec.KP, 0), key))
            else:
                if spec.packed_d is None or spec.packed_d.device != w.device:
                    spec.packed_d = torch.empty(spec.rows_pad_d * spec.KPd, device=w.device, dtype=torch.float32)
                    spec.packed_d_key = None
                if spec.packed_d_key != key:
                    stale.append((spec, True, w, spec.packed_d, (spec.cout, spec.cin, spec.k, spec.k, spec.rows_pad_d, spec.co_pad, spec.KPd, 1), key))
        if len(stale) < 2:
            for spec, dgrad, *_ in stale:
                self._pack(spec, dgrad=dgrad)
            return
        sig = tuple((w.data_ptr(), dst.data_ptr()) + prm for _, _, w, dst, prm, _ in stale)
        if self._pack_table is None or self._pack_table[0] != sig:
            import numpy as np
            rec = np.zeros(len(stale), dtype=np.dtype([('w', '<u8'), ('packed', '<u8'), ('p', '<i4', (8,))]))      # ssdhip_pack_job: 48 bytes
            for r, (_, _, w, dst, prm, _) in zip(rec, stale):
                r['w'], r['packed'], r['p'] = w.data_ptr(), dst.data_ptr(), prm
            dev = stale[0][2].device
            self._pack_table = (sig, torch.from_numpy(rec.view(np.uint8).copy()).to(dev), max(prm[4] * prm[6] for *_, prm, _ in stale))
        _, table, biggest = self._pack_table
        _lib.check(self.lib.ssdhip_pack_conv_weights(_lib.ptr(table), len(stale), biggest, _lib.stream()), 'pack_conv_weights')
        for spec, dgrad, _, _, _, key in stale:
            if dgrad:
                spec.packed_d_key = key
            else:
                spec.packed_key = key

    def _bn_state(self, name, c, dev, slot=0):
        key = name if slot == 0 else (name, slot)
        st = self._bn.get(key)
        if st is None or st['scale'].device != dev:
            st = {k: torch.empty(c, device=dev, dtype=torch.float32) for k in ('scale', 'shift', 'mean', 'invstd')}
            self._bn[key] = st
        return st

    def shared_stages(self):
        """Backbone stages whose output the trainable tail reads: those with a detector head (det_4_3 on conv_4_3) and the last frozen
        stage (input of conv_6_1)."""
        sh = {i for i in range(self.tail_start) if self.stages[i].head is not None}
        if 0 < self.tail_start:
            sh.add(self.tail_start - 1)
        return sh

    def _alt(self, b, kind, i, like):
        """Second copy (parity 1) of a hand-over buffer of stage i, allocated on first use."""
        tab = b.setdefault('alt', {})
        t = tab.get((kind, i))
        if t is None:
            t = torch.zeros_like(like) if kind == 'abf' else torch.empty_like(like)
            tab[(kind, i)] = t
        return t

    def _geometry(self, H, W):
        """Per-stage spatial sizes for an HxW input."""
        geo, h, w = [], H, W
        for s in self.stages:
            c = s.conv
            ho = (h + 2 * c.pad - c.k) // c.stride + 1
            wo = (w + 2 * c.pad - c.k) // c.stride + 1
            g = dict(hi=h, wi=w, ho=ho, wo=wo)
            h, w = ho, wo
            if s.pool_pad is not None:
                h = (h + 2 * s.pool_pad - 2) // 2 + 1
                w = (w + 2 * s.pool_pad - 2) // 2 + 1
            g['ha'], g['wa'] = h, w      # activation size after the optional pool
            geo.append(g)
        return geo

    def _buffers(self, N, H, W, dev):
        key = (N, H, W, str(dev))
        b = self._bufs.get(key)
        if b is not None:
            return b
        geo = self._geometry(H, W)
        f = dict(device=dev, dtype=torch.float32)
        b = dict(geo=geo, x4=torch.empty(N, H, W, 4, **f), z=[], pooled=[])
        offs, p = [], 0
        for s, g in zip(self.stages, geo):
            b['z'].append(torch.empty(N, g['ho'], g['wo'], s.conv.cout, **f))
            b['pooled'].append(torch.empty(N, g['ha'], g['wa'], s.conv.cout, **f) if s.pool_pad is not None else None)
            if s.head is not None:           # heads read the activation BEFORE the pool (ssd.py:100-104 order)
                offs.append(p)
                p += g['ho'] * g['wo'] * s.head.anchors
        b['head_off'], b['P'] = offs, p
        b['head_v'], b['head_f4'] = {}, set()
        head_m = 0
        for i, (s, g) in enumerate(zip(self.stages, geo)):
            if s.head is not None and s.head.wino_head and min(g['ho'], g['wo']) >= 10:
                if self._wgrad_wino4(s.head, g['ho'], g['wo']):
                    # heads on >= 16-px maps: forward AND weight gradient through F(4x4), sharing one input transform V (planes of 128-row multiples)
                    tp = _rup(N * ((g['ho'] + 3) // 4) * ((g['wo'] + 3) // 4), 128)
                    b['head_v'][i] = torch.empty(36 * tp * s.head.cin, **f)
                    b['head_f4'].add(i)
                    head_m = max(head_m, 36 * tp * s.head.co_pad)
                    continue
                tp = (N * ((g['ho'] + 1) // 2) * ((g['wo'] + 1) // 2) + 127) // 128 * 128
                b['head_v'][i] = torch.empty(16 * tp * s.head.cin, **f)
                head_m = max(head_m, 16 * tp * s.head.co_pad)     # own workspace: the heads run on the side stream
        b['head_m'] = torch.empty(head_m, **f) if head_m else None

        def workspace(stage_ids):
            """BN partial table, fp64 scratch and Winograd V / M planes for one set of stages (one set per stream that runs them)."""
            max_partial, cmax, wino_v, wino_m = 0, 1, 0, 0
            for i in stage_ids:
                s, g = self.stages[i], geo[i]
                M = N * g['ho'] * g['wo']
                # BN partial rows: one per 64-row m-tile (direct kernels, BM >= 64) or per 8 Winograd tiles; the 2x2 tiles of an
                # odd map cover more than its pixels (19x19 -> 10x10 tiles), so count tiles, not pixels / 4
                tiles = N * ((g['ho'] + 1) // 2) * ((g['wo'] + 1) // 2)
                max_partial = max(max_partial, (max((M + 63) // 64, (tiles + 7) // 8, 4096 if s.conv.cin_pad == 4 else 0) + 8) * s.conv.cout * 2)
                cmax = max(cmax, s.conv.cout)
                if self._wino4_ok(s.conv, g, sizing=True):
                    _, tp4, _ = self._wino4_plan(s.conv, g, N)
                    wino_v, wino_m = max(wino_v, 36 * tp4 * s.conv.cin), max(wino_m, 36 * tp4 * s.conv.cout)
                    max_partial = max(max_partial, (self.lib.ssdhip_winograd4_stat_rows(N, g['hi'], g['wi'], s.conv.cout) + 8) * s.conv.cout * 2)
                if s.conv.wino and self._wino_ok(g):
                    tp = (N * ((g['hi'] + 1) // 2) * ((g['wi'] + 1) // 2) + 127) // 128 * 128
                    wino_v, wino_m = max(wino_v, 16 * tp * s.conv.cin), max(wino_m, 16 * tp * s.conv.cout)
            return dict(partial=torch.empty(max(max_partial, 1), **f), ws64=torch.empty(64 * cmax * 2, device=dev, dtype=torch.float64),
                        wino_v=torch.empty(wino_v, **f) if wino_v else None, wino_m=torch.empty(wino_m, **f) if wino_m else None)

        n_st = len(self.stages)
        tail_ws = workspace(range(self.tail_start, n_st))
        b['ws'] = [workspace(range(self.tail_start)) if self.tail_start > 0 else tail_ws, tail_ws]    # [backbone, tail]
        if len(self._bufs) >= 3:                 # keep the three most recent batch geometries (≈3.4 GB each at N=32)
            self._bufs.pop(next(iter(self._bufs)))
        self._bufs[key] = b
        return b

    def _wf_geo(self, b, N, h, w, dev):
        """Tile-block geometry table of an (N, h, w) map for the fused Winograd kernel: filled once per buffer set and shape."""
        tab = b.setdefault('wf_geo', {})
        t = tab.get((h, w))
        if t is None:
            t = torch.empty(_lib.check(self.lib.ssdhip_wino_fused_geo_bytes(N, h, w), 'wino_fused_geo_bytes'), device=dev, dtype=torch.uint8)
            _lib.check(self.lib.ssdhip_wino_fused_geo(N, h, w, _lib.ptr(t), _lib.stream()), 'wino_fused_geo')
            tab[(h, w)] = t
        return t

    def _fused_ok(self, c, g, N):
        """Layers that run on the fused Winograd kernel: 3x3/s1/p1, wide maps (>= 70 px: from the 38x38 maps down the three-pass form
        is faster), channel counts the kernel tiles (Cin % 16, Cout % 64), tensors within its 32-bit offsets; fp32 mode only."""
        if self.bf16 or c.is_head or self._no_fused:
            return False
        return (c.k == 3 and c.stride == 1 and c.pad == 1 and c.cin % 16 == 0 and 32 <= c.cin <= 512 and c.cout % 64 == 0
                and min(g['hi'], g['wi']) >= self._fused_min and N * g['hi'] * g['wi'] * max(c.cin * 4, c.cout) < (1 << 31)
                and N * ((g['hi'] + 1) // 2) < 65536)

    def _patch_ok(self, i, N, geo):
        """bf16 mode (BASELINE configs[3] / [4]): frozen 3x3/s1/p1 backbone layers run on the bf16 patch convolution (conv_bf16.hip),
        reading an ACTIVATED bf16 tensor with a zero halo that the producer's apply pass wrote -- the activations of the backbone are
        bf16 in HBM.  The trainable tail keeps the implicit-GEMM bf16 template (fp32 tensors, operands rounded at staging)."""
        if not self.bf16 or i <= 0 or i >= len(self.stages) or i > self.tail_start or self._no_patch:
            return False
        c, g = self.stages[i].conv, geo[i]
        if i == self.tail_start and (self.stages[i - 1].pool_pad is not None or self.stages[i - 1].conv.is_head):
            return False               # its weight gradient reads the fp32 (z, scale, shift) of the stage below: no pool in between
        # i == tail_start: the first trainable layer (conv_6_1) reads the backbone's last activation; its forward fits the same kernel
        # (weights repacked when they change), its backward keeps reading the fp32 z + scale / shift of that activation
        return ((c.frozen or i == self.tail_start) and not c.is_head and c.k == 3 and c.stride == 1 and c.pad == 1 and c.cin % 32 == 0 and c.cin >= 32
                and c.cout % 64 == 0 and g['wi'] >= 14 and self.lib.ssdhip_conv_bf16_act_bytes(N, g['hi'], g['wi'], c.cin) < (1 << 32)
                and N * g['hi'] * g['wi'] * c.cout < (1 << 31))

    @staticmethod
    def _wino_ok(g):
        return g['hi'] <= 160 and g['wi'] <= 160        # from the 150x150 maps down (x1.1 there, x1.8 at 38x38); 300x300 is transform-traffic-bound

    def _wino_weights(self, spec):
        w, _ = self._weight(spec)
        key = (w.data_ptr(), w._version, self.m._hip_epoch if not spec.frozen else 0)
        rows = spec.rows_pad if not spec.is_head else _rup(spec.co_pad, 128)
        if spec.wino_u is None or spec.wino_u.device != w.device:
            spec.wino_u = torch.empty(16 * rows * spec.cin, device=w.device, dtype=torch.float32)
            spec.wino_key = None
        if spec.wino_key != key:
            _lib.check(self.lib.ssdhip_winograd_weight(_lib.ptr(w), _lib.ptr(spec.wino_u), spec.cout, spec.cin, rows, spec.cin,
                                                       _lib.stream()), 'winograd_weight')
            spec.wino_key = key
        return spec.wino_u

    def _plane_gemm(self, p, tp, cin, cout, w_plane_stride, what, planes=16, spec=None):
        """ssdhip_conv2d_nhwc as 16 (F(2x2)) or 36 (F(4x4)) plane GEMMs; the kernel addresses its input with 32-bit byte offsets (2^30
        elements), so big batches run the planes in groups (same kernel, pointers advanced)."""
        groups = 1
        while (planes // groups) * tp * max(cin, cout) >= self._plane_limit and groups < planes:
            groups = next(d for d in range(groups + 1, planes + 1) if planes % d == 0)
        per = planes // groups
        x0, w0, y0 = p.x, p.w, p.y
        p.N = per
        # 128 x 128 plane GEMMs whose tile count fits the resident slots badly (conv_3_2 .. conv_4_3) run on the work-queue kernel when handed
        # its workspace (gemm_planes.hip); its control words start at zero and every launch leaves them at zero
        if self._plane_split and p.tile == 128128 and cin % 32 == 0:
            p.compute = 2
        need = self.lib.ssdhip_conv2d_ws_bytes(ctypes.byref(p)) if p.compute == 0 else 0
        if need > 0 and spec is not None and not self._no_plane_queue:
            bufs = spec.__dict__.setdefault('planeq_ws', {})
            key = (what.split(' ')[0], torch.cuda.current_device())
            ws = bufs.get(key)
            if ws is None or ws.numel() * 4 < need:
                ws = torch.zeros((need + 3) // 4, device=torch.device('cuda', torch.cuda.current_device()), dtype=torch.float32)
                bufs[key] = ws
            p.ws, p.ws_bytes = ws.data_ptr(), ws.numel() * 4
            p.max_cus = self._pq_cus
        for gi in range(groups):
            p.x = x0 + gi * per * tp * cin * 4
            p.w = w0 + gi * per * w_plane_stride * 4
            p.y = y0 + gi * per * tp * cout * 4
            _lib.check(self.lib.ssdhip_conv2d_nhwc(ctypes.byref(p), _lib.stream()), what)
        queued = bool(p.ws)
        p.x, p.w, p.y, p.N = x0, w0, y0, planes
        p.ws, p.ws_bytes, p.compute, p.max_cus = None, 0, 0, 0
        return queued

    def _wino4_ok(self, c, g, sizing=False):
        """Layers on the Winograd F(4x4,3x3) passes: forward of the frozen backbone's 3x3/s1/p1 layers and of the first trainable layer
        on top of it (conv_6_1: its weight gradient reads the backbone's activation and dz, not the forward's products; no data gradient
        is taken through it), K = Cin >= 256 (the transforms move 2.25x the activation bytes each way: they pay for themselves from
        K = 256 on), maps of 8 .. SSD_W4_MAX_MAP (160) px (conv_3_2 .. conv_6_1 of SSD300: 75, 38, 19 px; 128, 64 and 32 px of SSD512); fp32 mode only."""
        first_tail = 0 < self.tail_start < len(self.stages) and c is self.stages[self.tail_start].conv
        return (self._wino_mode == 'f4' and (sizing or not self.bf16) and (c.frozen or first_tail) and not c.is_head
                and c.k == 3 and c.stride == 1 and c.pad == 1
                and c.cin >= self._w4_min_cin and c.cin % 32 == 0 and c.cout % 4 == 0 and 256 % (c.cout // 4) == 0
                and 8 <= min(g['hi'], g['wi']) and max(g['hi'], g['wi']) <= self._w4_max_map)

    def _wino4_plan(self, c, g, N):
        """(tiles, padded tiles per plane, GEMM tile code) of an F(4x4) layer: 128x128 GEMM tiles once the grid has > 2000 of them, else
        64x128 (and planes padded to 64 rows only: 800 tiles of a 19x19 map at batch 32 -> 832, not 896)."""
        T = N * ((g['hi'] + 3) // 4) * ((g['wi'] + 3) // 4)
        tp128 = _rup(T, 128)
        if (36 * tp128 // 128) * ((c.cout + 127) // 128) > 2000:
            return T, tp128, 128128
        return T, _rup(T, 64), 64128

    def _wino4_weights(self, spec):
        w, _ = self._weight(spec)
        key = (w.data_ptr(), w._version, self.m._hip_epoch if not spec.frozen else 0)
        rows = spec.rows_pad if not spec.is_head else _rup(spec.co_pad, 128)
        if spec.wino4_u is None or spec.wino4_u.device != w.device:
            spec.wino4_u = torch.empty(36 * rows * spec.cin, device=w.device, dtype=torch.float32)
            spec.wino4_key = None
        if spec.wino4_key != key:
            _lib.check(self.lib.ssdhip_winograd4_weight(_lib.ptr(w), _lib.ptr(spec.wino4_u), spec.cout, spec.cin, rows, spec.cin,
                                                        _lib.stream()), 'winograd4_weight')
            spec.wino4_key = key
        return spec.wino4_u

    def _conv_winograd4(self, spec, x, xs, z, ws, N, h, w, g, stats):
        """3x3/s1/p1 conv as F(4x4,3x3): input transform -> 36 plane GEMMs (one launch) -> output transform.  Returns the number of BN
        partial rows written to `stats`."""
        L, st = self.lib, _lib.stream()
        U = self._wino4_weights(spec)
        _, bias = self._weight(spec)
        V, Mo = ws['wino_v'], ws['wino_m']
        T, tp, tile = self._wino4_plan(spec, g, N)
        _lib.check(L.ssdhip_winograd4_input(_lib.ptr(x), _lib.ptr(xs[0]) if xs is not None else None,
                                            _lib.ptr(xs[1]) if xs is not None else None, _lib.ptr(V), N, h, w, spec.cin, tp, st), f'winograd4_input {spec.name}')
        p = _lib.ConvParams()
        p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), U.data_ptr(), Mo.data_ptr(), tp * spec.cout
        p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 36, 1, tp, spec.cin, 1, tp, spec.cout
        p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate = 1, 1, 1, 0, spec.cin, 0, 0
        p.tile = tile
        p.rows_per_img, p.w_img_stride = tp, spec.rows_pad * spec.cin
        if self.prof is None:
            self._plane_gemm(p, tp, spec.cin, spec.cout, spec.rows_pad * spec.cin, f'winograd4-gemm {spec.name}', planes=36, spec=spec)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            queued = self._plane_gemm(p, tp, spec.cin, spec.cout, spec.rows_pad * spec.cin, f'winograd4-gemm {spec.name}', planes=36, spec=spec)
            e1.record()
            self.prof.append(dict(layer=spec.name + ' (F(4x4) winograd-domain GEMM)',
                                  kernel='gemm_planes_queue_kernel' if queued else 'gemm_planes_dma_kernel<%d, 128, 0, 8, 0>' % (tile // 1000),
                                  direct_flops=2.0 * N * h * w * spec.cin * spec.cout * 9,
                                  flops=2.0 * 36 * tp * spec.cin * spec.cout, flops_unpadded=2.0 * 36 * T * spec.cin * spec.cout, e0=e0, e1=e1))
        return _lib.check(L.ssdhip_winograd4_output(_lib.ptr(Mo), _lib.ptr(bias), _lib.ptr(z), _lib.ptr(stats), 0, N, h, w, spec.cout, tp, st),
                          f'winograd4_output {spec.name}')

    def _conv_winograd(self, spec, x, xs, z, ws, N, h, w, stats):
        """3x3/s1/p1 conv as input transform -> 16 plane GEMMs (one launch of the implicit-GEMM kernel) -> output transform.
        Returns the number of BN partial rows written to `stats`."""
        L, st = self.lib, _lib.stream()
        U = self._wino_weights(spec)
        _, bias = self._weight(spec)
        V, Mo = ws['wino_v'], ws['wino_m']
        tp = (N * ((h + 1) // 2) * ((w + 1) // 2) + 127) // 128 * 128
        _lib.check(L.ssdhip_winograd_input(_lib.ptr(x), _lib.ptr(xs[0]) if xs is not None else None,
                                           _lib.ptr(xs[1]) if xs is not None else None, _lib.ptr(V), N, h, w, spec.cin, st), f'winograd_input {spec.name}')
        p = _lib.ConvParams()
        p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), U.data_ptr(), Mo.data_ptr(), tp * spec.cout
        p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 16, 1, tp, spec.cin, 1, tp, spec.cout
        p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate = 1, 1, 1, 0, spec.cin, 0, 0
        # measured (scripts/wino_bench.py): 128x128 tiles once the grid has > 2000 of them, 64x128 below
        p.tile = 128128 if (16 * tp // 128) * ((spec.cout + 127) // 128) > 2000 else 64128
        p.rows_per_img, p.w_img_stride = tp, spec.rows_pad * spec.cin
        if self.prof is None:
            self._plane_gemm(p, tp, spec.cin, spec.cout, spec.rows_pad * spec.cin, f'winograd-gemm {spec.name}', spec=spec)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._plane_gemm(p, tp, spec.cin, spec.cout, spec.rows_pad * spec.cin, f'winograd-gemm {spec.name}', spec=spec)
            e1.record()
            self.prof.append(dict(layer=spec.name + ' (winograd-domain GEMM)',
                                  # the symbol rocprofv3 shows: K >= 256 planes and the K = 128 specialisation run on the LDS-DMA kernel
                                  # (gemm_planes.hip: BM, BN, ablation, waves, compile-time K-steps), other K on the register-staged one
                                  kernel=('gemm_planes_dma_kernel<128, 128, 0, 8, 4>' if (spec.cin == 128 and p.tile == 128128) else
                                          'gemm_planes_dma_kernel<128, 128, 0, 8, 0>' if (spec.cin >= 256 and p.tile == 128128) else
                                          'gemm_planes_dma_kernel<%d, 128, 0, 8, 0>' % (p.tile // 1000) if spec.cin >= 256 else
                                          'conv_igemm_kernel<%d, 128, 1, 0, false, false, 0, false>' % (p.tile // 1000)),
                                  direct_flops=2.0 * N * h * w * spec.cin * spec.cout * 9,
                                  flops=2.0 * 16 * tp * spec.cin * spec.cout,
                                  flops_unpadded=2.0 * 16 * N * ((h + 1) // 2) * ((w + 1) // 2) * spec.cin * spec.cout, e0=e0, e1=e1))
        return _lib.check(L.ssdhip_winograd_output(_lib.ptr(Mo), _lib.ptr(bias), _lib.ptr(z), _lib.ptr(stats), N, h, w, spec.cout, st),
                          f'winograd_output {spec.name}')

    def _head_winograd4(self, spec, x, xs, V, b, out_ptr, out_batch_stride, N, h, w):
        """Detector head forward through Winograd F(4x4,3x3); V stays valid for the F(4x4) weight gradient in backward."""
        L, st = self.lib, _lib.stream()
        U = self._wino4_weights(spec)
        _, bias = self._weight(spec)
        Mo = b['head_m']
        tp = _rup(N * ((h + 3) // 4) * ((w + 3) // 4), 128)
        _lib.check(L.ssdhip_winograd4_input(_lib.ptr(x), _lib.ptr(xs[0]) if xs is not None else None,
                                            _lib.ptr(xs[1]) if xs is not None else None, _lib.ptr(V), N, h, w, spec.cin, tp, st), f'winograd4_input {spec.name}')
        cp = spec.co_pad
        rows = _rup(cp, 128)
        p = _lib.ConvParams()
        p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), U.data_ptr(), Mo.data_ptr(), tp * cp
        p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 36, 1, tp, spec.cin, 1, tp, cp
        p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate = 1, 1, 1, 0, spec.cin, 0, 0
        p.tile = 128128 if cp <= 128 else 128064
        p.rows_per_img, p.w_img_stride = tp, rows * spec.cin
        if self.prof is None:
            self._plane_gemm(p, tp, spec.cin, cp, rows * spec.cin, f'winograd4-gemm-head {spec.name}', planes=36, spec=spec)
        else:               # roofline leg of bench.py: every launch of the plane GEMM is bracketed, so its average matches rocprofv3's
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            queued = self._plane_gemm(p, tp, spec.cin, cp, rows * spec.cin, f'winograd4-gemm-head {spec.name}', planes=36, spec=spec)
            e1.record()
            self.prof.append(dict(layer=spec.name + ' (F(4x4) winograd-domain GEMM, head)',
                                  kernel=('gemm_planes_queue_kernel' if queued else 'gemm_planes_dma_kernel<128, 128, 0, 8, 0>' if p.tile == 128128 else 'gemm_planes_dma_kernel<128, 64, 0, 4, 0>'),
                                  direct_flops=2.0 * N * h * w * spec.cin * spec.cout * 9,
                                  flops=2.0 * 36 * tp * spec.cin * cp,
                                  flops_unpadded=2.0 * 36 * N * ((h + 3) // 4) * ((w + 3) // 4) * spec.cin * spec.cout, e0=e0, e1=e1))
        _lib.check(L.ssdhip_winograd4_output_head(_lib.ptr(Mo), _lib.ptr(bias), out_ptr, out_batch_stride, N, h, w, spec.cout, cp, tp, st),
                   f'winograd4_output_head {spec.name}')

    def _head_winograd(self, spec, x, xs, V, b, out_ptr, out_batch_stride, N, h, w):
        """Detector head forward through Winograd; V (input transform of the head's activation) stays valid for backward."""
        L, st = self.lib, _lib.stream()
        U = self._wino_weights(spec)
        _, bias = self._weight(spec)
        Mo = b['head_m']
        tp = (N * ((h + 1) // 2) * ((w + 1) // 2) + 127) // 128 * 128
        _lib.check(L.ssdhip_winograd_input(_lib.ptr(x), _lib.ptr(xs[0]) if xs is not None else None,
                                           _lib.ptr(xs[1]) if xs is not None else None, _lib.ptr(V), N, h, w, spec.cin, st), f'winograd_input {spec.name}')
        cp = spec.co_pad
        p = _lib.ConvParams()
        p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), U.data_ptr(), Mo.data_ptr(), tp * cp
        p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 16, 1, tp, spec.cin, 1, tp, cp
        p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate = 1, 1, 1, 0, spec.cin, 0, 0
        p.tile = 128128 if cp <= 128 else 128064
        p.rows_per_img, p.w_img_stride = tp, _rup(cp, 128) * spec.cin
        self._plane_gemm(p, tp, spec.cin, cp, _rup(cp, 128) * spec.cin, f'winograd-gemm-head {spec.name}', spec=spec)
        _lib.check(L.ssdhip_winograd_output_head(_lib.ptr(Mo), _lib.ptr(bias), out_ptr, out_batch_stride, N, h, w, spec.cout, cp, st),
                   f'winograd_output_head {spec.name}')

    def _conv(self, spec, x, xs, y, y_batch_stride, N, hi, wi, ho, wo, stats, tile=0):
        w = self._pack(spec)
        _, bias = self._weight(spec)
        p = _lib.ConvParams()
        p.x, p.w, p.bias = x.data_ptr(), w.data_ptr(), (bias.data_ptr() if bias is not None else None)
        p.in_scale = xs[0].data_ptr() if xs is not None else None
        p.in_shift = xs[1].data_ptr() if xs is not None else None
        p.y = y
        p.stats = stats.data_ptr() if stats is not None else None
        p.y_batch_stride = y_batch_stride
        p.N, p.Hi, p.Wi, p.Cin = N, hi, wi, spec.cin_pad
        p.Ho, p.Wo, p.Cout = ho, wo, spec.cout
        p.R, p.S, p.stride, p.pad = spec.k, spec.k, spec.stride, spec.pad
        p.KP, p.mode, p.accumulate, p.tile = spec.KP, 0, 0, tile
        p.compute = int(self.bf16 and spec.cin_pad != 4)      # the padded-RGB layer (K = 27) stays on the fp32 kernel
        self._splitk_ws(spec, p, ('fwd', N, hi, wi))
        bm = self.lib.ssdhip_conv2d_block_m(ctypes.byref(p))
        # roofline leg of bench.py: only launches of >= 10 GFLOP are bracketed (the dominant kernel is among them; an event pair
        # costs a few microseconds of stream time, which the ~40 small launches of the tail would pay for nothing)
        if self.prof is None or 2.0 * N * ho * wo * spec.cout * spec.cin * spec.k * spec.k < 1e10:
            _lib.check(self.lib.ssdhip_conv2d_nhwc(ctypes.byref(p), _lib.stream()), f'conv2d {spec.name}')
            return bm
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(self.lib.ssdhip_conv2d_nhwc(ctypes.byref(p), _lib.stream()), f'conv2d {spec.name}')
        e1.record()
        tile = self.lib.ssdhip_conv2d_tile(ctypes.byref(p))
        # the symbol rocprofv3 reports for this launch (template arguments BM, BN, KS, MODE, XFORM, SMALLC, ABL)
        kernel = 'conv_igemm_kernel<%d, %d, %d, 0, %s, %s, 0%s>' % (tile // 1000, tile % 1000, spec.k,
                                                                    'true' if xs is not None else 'false',
                                                                    'true' if spec.cin_pad == 4 else 'false',
                                                                    ', true' if p.compute else ', false')
        self.prof.append(dict(layer=spec.name, kernel=kernel, flops=2.0 * N * ho * wo * spec.cout * spec.cin * spec.k * spec.k,
                              e0=e0, e1=e1))
        return bm

    def _splitk_ws(self, spec, p, tag):
        """Split-K workspace of one (layer, direction): launches on the tiny maps split K over blocks (conv_igemm.hip).  One buffer
        per layer and direction, because heads, tail convolutions and data gradients of different layers run on different streams."""
        if self._no_splitk:
            return
        need = self.lib.ssdhip_conv2d_ws_bytes(ctypes.byref(p))
        if need <= 0:
            return
        # keyed by direction only and grown to the largest need: ragged last batches, evaluation and inference batch sizes must not
        # each leave their own slab set behind (launches of one layer and direction are ordered on one stream, whatever the batch)
        bufs = spec.__dict__.setdefault('splitk_ws', {})
        key = tag[0]
        buf = bufs.get(key)
        if buf is None or buf.numel() * 4 < need or buf.device.index != torch.cuda.current_device():
            if buf is not None and buf.device.index == torch.cuda.current_device():
                buf.record_stream(torch.cuda.current_stream())      # its last reader is on this stream: recycle only behind it
            buf = torch.empty((need + 3) // 4, device=torch.device('cuda', torch.cuda.current_device()), dtype=torch.float32)
            bufs[key] = buf
        p.ws, p.ws_bytes = buf.data_ptr(), buf.numel() * 4

    # ------------------------------------------------------------------------------------------------ streams
    def _note_reader(self, stage, event):
        if event is not None and not self.single_stream and self.tail_start > 0 and not self._phased:
            self._readers.setdefault(stage, []).append(event)

    def _side_stream(self, dev, which=0):
        st = self._side.get((dev, which))
        if st is None:
            # the frozen backbone (which = 2) fills the chip with long launches; the side stream's short, dependent launches (heads,
            # weight gradients, repacking) go first whenever both have work ready (SSD_STREAM_PRIO=0 disables)
            mode = os.environ.get('SSD_STREAM_PRIO', '1')       # 1 (default): side streams high; 0: all normal; bb (A/B): the backbone stream high instead
            prio = -1 if ((which != 2 and mode == '1') or (which == 2 and mode == 'bb')) else 0
            st = torch.cuda.Stream(device=dev, priority=prio)
            self._side[(dev, which)] = st
        return st

    class _Fork:
        """`with fork.side():` enqueues on the side stream after everything the main stream has enqueued so far;
        `fork.join()` makes the main stream wait for the side stream.  No-op object when single_stream is set."""

        def __init__(self, engine, dev, which=0, enabled=True):
            self.main = torch.cuda.current_stream(dev)
            self.side_stream = None if (engine.single_stream or not enabled) else engine._side_stream(dev, which)

        def side(self, after=None):
            """after: the stream whose queue the side work must follow (default: the main stream)."""
            if self.side_stream is None:
                return contextlib.nullcontext()
            self.side_stream.wait_event((after or self.main).record_event())
            return torch.cuda.stream(self.side_stream)

        def join(self):
            if self.side_stream is not None:
                self.main.wait_event(self.side_stream.record_event())

        def mark(self):
            """Event at the side stream's current tail (None when single-stream)."""
            return self.side_stream.record_event() if self.side_stream is not None else None

        def wait(self, event):
            if event is not None:
                self.main.wait_event(event)

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, x, save, ready_event=None, phase='all', parity=0):
        """phase 'all' (default): the whole forward.  'backbone' / 'tail' (graph_step.py): the frozen stages alone, everything on the
        current stream, leaving their hand-over buffers of `parity` for a later 'tail' call (x is None there), which runs the detector
        heads, the trainable stages and returns the outputs.  Same launches in the same per-stream order as 'all'."""
        if phase not in ('all', 'backbone', 'tail'):
            raise ValueError(f'unknown phase {phase!r}')
        if phase != 'all' and not (0 < self.tail_start < len(self.stages)) or (phase != 'all' and self.headless):
            raise RuntimeError('phased execution needs a frozen backbone under a trainable tail with detector heads')
        if phase == 'tail':
            ctx = self._bb_ctx.get(parity)
            if ctx is None:
                raise RuntimeError(f"forward(phase='tail', parity={parity}) without a backbone pass of that parity")
            b, N, H, W, dev = ctx['b'], ctx['N'], ctx['H'], ctx['W'], ctx['dev']
            training, self.bf16 = ctx['training'], ctx['bf16']
        else:
            _lib.require_gpu(x)
            if x.dim() != 4 or x.shape[1] != 3:
                raise ValueError(f'expected (N, 3, H, W) input, got {tuple(x.shape)}')
            x_in = x
            x = _lib.f32c(x)
            if x is not x_in:
                ready_event = None      # a contiguous fp32 copy was just made on the caller's stream: its completion is what counts
            N, _, H, W = x.shape
            dev = x.device
            b = self._buffers(N, H, W, dev)
            training = self.m.training
            self.bf16 = self.m.compute_dtype == 'bf16'
        geo = b['geo']
        shared = self.shared_stages() if parity else ()
        out = torch.empty(N, b['P'], self.row, device=dev, dtype=torch.float32) if phase != 'backbone' else None
        acts = []            # per stage: (z, xform) read by its head, (tensor, xform) read by the next conv
        nbt = []
        fork = self._Fork(self, dev, enabled=phase != 'backbone')
        main = fork.main
        # The frozen backbone depends on the images only, so it gets its own stream: the backbone of step i+1 starts as soon as
        # its images are there, i.e. next to the tail forward, the tail backward and the optimizer of step i (small, latency-bound
        # launches that cannot fill the chip).  Only two backbone outputs have consumers outside the backbone stream -- the inputs
        # of det_4_3 and of conv_6_1 (forward and weight gradient): those stages wait for the events their readers leave in _readers.
        bb = None if (self.single_stream or self.tail_start == 0 or phase != 'all') else self._side_stream(dev, which=2)
        if self._cus is None:
            self._cus = int(self.lib.ssdhip_cu_count())
        if self._wf_blocks_env is None:
            # 7/8 of the CUs (224 of 256 on an MI355X), a multiple of 32 so that every XCD gets whole channel-tile groups
            beside_tail = (bb is not None or (phase != 'all' and not self.single_stream)) and self.tail_start < len(self.stages)
            self._wf_blocks = max(32, self._cus * 7 // 8 // 32 * 32) if beside_tail else 0
        entry = main.record_event() if fork.side_stream is not None else None
        if fork.side_stream is not None:
            fork.side_stream.wait_event(entry)               # `out` may reuse memory that step i's loss kernels still read
        if bb is not None:
            # ready_event (optional, from the caller): "x is complete".  With it the backbone does not queue behind the caller's
            # stream, i.e. not behind the previous step's backward and optimizer; without it, it waits for everything enqueued.
            bb.wait_event(ready_event if ready_event is not None else entry)
            x.record_stream(bb)                              # the allocator must not recycle x while the backbone stream reads it
        if save and self.tail_start > 0 and phase != 'backbone':
            # the optimizer just changed every trainable weight: repack them (forward + data-gradient layouts, Winograd U) on the
            # side stream while the frozen backbone runs; the later _pack / _wino_weights calls find their keys current
            with fork.side():
                plain = []               # layouts of the implicit-GEMM kernels: one batched launch (_pack_many)
                for i, (s, g) in enumerate(zip(self.stages, geo)):       # includes det_4_3 on the frozen conv_4_3
                    for spec in (s.conv, s.head):
                        if spec is None or spec.frozen:
                            continue
                        if self._wino4_ok(spec, g) or (spec.is_head and i in b['head_f4'] and not self.bf16):
                            self._wino4_weights(spec)
                        elif not self.bf16 and ((spec.wino and self._wino_ok(g)) or (spec.is_head and i in b['head_v'])):
                            self._wino_weights(spec)
                        else:
                            plain.append((spec, False))
                        if i > self.tail_start or spec.is_head and i >= self.tail_start:
                            plain.append((spec, True))
                self._pack_many(plain)
            prepacked = fork.mark()
        else:
            prepacked = None
        if phase != 'tail':
            with (torch.cuda.stream(bb) if bb is not None else contextlib.nullcontext()):
                _lib.check(self.lib.ssdhip_normalize_nchw_to_nhwc4(_lib.ptr(x), _lib.ptr(b['x4']), N, H, W, _lib.stream()), 'normalize')
        cur, cur_xs = b['x4'], None
        for i, (s, g) in enumerate(zip(self.stages, geo)):
            c = s.conv
            bnm = self._feat[s.bn_name]
            on_bb = bb is not None and i < self.tail_start
            if phase == 'backbone' and i >= self.tail_start:
                break
            if phase == 'tail' and i < self.tail_start:
                # the backbone pass of this parity left z / BN state / the next layer's input: only the detector head runs here
                acts.append(ctx['acts'][i])
                cur, cur_xs = ctx['acts'][i][2], ctx['acts'][i][3]
                if i == self.tail_start - 1:
                    cur, cur_xs = ctx['cur'], ctx['cur_xs']
                if s.head is not None:
                    self._launch_head(i, s, g, b, acts[i][0], acts[i][1], out, N, fork, None)
                continue
            if i == self.tail_start:
                if bb is not None:
                    main.wait_event(bb.record_event())       # backbone done: the tail continues on the caller's stream
                fork.wait(prepacked)
            ws = b['ws'][0 if i < self.tail_start else 1]
            alt = i in shared              # parity 1: this stage's hand-over buffers are the second copies
            z = b['z'][i] if not alt else self._alt(b, 'z', i, b['z'][i])
            M = N * g['ho'] * g['wo']
            bst = self._bn_state(s.bn_name, c.cout, dev, 1 if alt else 0)
            with (torch.cuda.stream(bb) if on_bb else contextlib.nullcontext()):
                if on_bb:
                    for ev in self._readers.pop(i, ()):      # last step's readers of this stage's output / BN state
                        bb.wait_event(ev)
                st = _lib.stream()
                pk = b.setdefault('patch', {})               # per buffer set: (stage, mode switches) -> runs on the bf16 patch kernel
                for j in (i, i + 1):
                    if (j, self.bf16) not in pk:
                        pk[(j, self.bf16)] = self._patch_ok(j, N, geo)
                patch, nxt_patch = pk[(i, self.bf16)], pk[(i + 1, self.bf16)]
                # z of a patch layer is bf16 when only the next layer's apply pass reads it (no head, no fp32 consumer)
                z_bf16 = (patch and nxt_patch and s.head is None and i + 1 < self.tail_start
                          and self._z_bf16)
                pool_fused = False
                if patch:
                    w1, b1 = self._weight(c)
                    key = (w1.data_ptr(), w1._version, self.m._hip_epoch if not c.frozen else 0)
                    if getattr(c, 'cb_img', None) is None or c.cb_img.device != w1.device:
                        c.cb_img = torch.empty(self.lib.ssdhip_conv_bf16_weight_bytes(c.cout, c.cin), device=w1.device, dtype=torch.uint8)
                        c.cb_key = None
                    if c.cb_key != key:
                        _lib.check(self.lib.ssdhip_conv_bf16_pack(_lib.ptr(w1), _lib.ptr(c.cb_img), c.cout, c.cin, st), f'conv_bf16_pack {c.name}')
                        c.cb_key = key
                    if self.prof is not None:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    rows = _lib.check(self.lib.ssdhip_conv_bf16(_lib.ptr(cur), _lib.ptr(c.cb_img), _lib.ptr(b1) if b1 is not None else None, _lib.ptr(z),
                                                                int(z_bf16), _lib.ptr(ws['partial']) if training else None, N, g['hi'], g['wi'],
                                                                c.cin, c.cout, self._wf_blocks, st), f'conv2d {c.name}')
                    if self.prof is not None:
                        e1.record()
                        fl = 2.0 * N * g['hi'] * g['wi'] * c.cin * c.cout * 9
                        self.prof.append(dict(layer=c.name + ' (bf16 patch)', e0=e0, e1=e1, flops=fl, peak_tflops=2500.0,
                                              kernel='conv_bf16_kernel<%s, %s>' % ('4, 2, 2' if c.cout % 128 == 0 else '2, 2, 1', 'true' if z_bf16 else 'false')))
                elif self._wino4_ok(c, g):
                    rows = self._conv_winograd4(c, cur, cur_xs, z, ws, N, g['hi'], g['wi'], g, ws['partial'] if training else None)
                elif self._fused_ok(c, g, N):
                    # conv_1_2 .. conv_3_1 (300x300 .. 75x75 maps, K <= 128; conv_3_2 / conv_3_3 too under SSD_WINOGRAD=f2): the one-kernel Winograd form (wino_fused.hip); nothing of the
                    # Winograd domain in HBM.  Measured against the paths below (scripts/wf_bench.py, batch 32): conv_1_2 1.57 vs 1.80 ms,
                    # conv_2_1 0.71 vs 0.86, conv_2_2 1.17 vs 1.6-1.9, conv_3_1 0.71 vs 0.79, conv_3_2/3 1.18 vs 1.23; from the 38x38
                    # maps down the three-pass form wins (its 128x128 GEMM tiles need half the L2 -> LDS bytes per flop)
                    w1, b1 = self._weight(c)
                    key = (w1.data_ptr(), w1._version, self.m._hip_epoch if not c.frozen else 0)
                    if getattr(c, 'wf_img', None) is None or c.wf_img.device != w1.device:
                        c.wf_img = torch.empty(self.lib.ssdhip_wino_fused_weight_floats(c.cout, c.cin), device=w1.device, dtype=torch.float32)
                        c.wf_key = None
                    if c.wf_key != key:
                        _lib.check(self.lib.ssdhip_wino_fused_pack(_lib.ptr(w1), _lib.ptr(c.wf_img), c.cout, c.cin, st), f'wino_fused_pack {c.name}')
                        c.wf_key = key
                    if self.prof is not None:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    # frozen layer in front of a 2x2 / pad 0 max-pool on an even map (conv_1_2, conv_2_2): the kernel stores the window
                    # maxima and minima of z instead of z, the pool pass becomes a select over a quarter-size pair (bit-identical result)
                    pool_fused = (self._pool_fuse and c.frozen and s.pool_pad == 0 and s.head is None and not nxt_patch
                                  and g['ho'] % 2 == 0 and g['wo'] % 2 == 0)
                    # opt-in (SSD_EARLY_F4=1): the K = 128 layers among these (conv_2_2, conv_3_1) through the two-pass F(4x4) form
                    # (csrc/wino4_fused.hip): 0.90 vs 0.99 and 0.42 vs 0.49 ms per layer (scripts/w4f_bench.py), at F(4x4)'s 2.7x rounding error
                    early_f4 = self._early_f4 and c.frozen and c.cin >= 128 and c.cin % 32 == 0 and c.cout % 64 == 0 and not self.bf16
                    if pool_fused:
                        zmm = b.setdefault('zmm', {}).get(i)
                        if zmm is None:
                            # the window maxima and minima, or (default) only the extreme the pool will use: the sign of a channel's BatchNorm
                            # scale is the sign of its weight, known before the batch statistics (SSD_POOL_PAIR=1: the pair form, for A/B)
                            zmm = torch.empty(2 if (self._pool_pair or early_f4) else 1, N, g['ha'], g['wa'], c.cout, device=dev, dtype=torch.float32)
                            b['zmm'][i] = zmm
                    if early_f4:
                        if getattr(c, 'w4f_img', None) is None or c.w4f_img.device != w1.device or c.w4f_key != key:
                            from .. import ops
                            c.w4f_img, c.w4f_key = ops.wino4f_pack(w1), key
                        need = _lib.check(self.lib.ssdhip_wino4f_v_floats(N, g['hi'], g['wi'], c.cin), 'wino4f_v_floats')
                        vb = b.get('w4f_vb')
                        if vb is None or vb.numel() < need:
                            vb = b['w4f_vb'] = torch.empty(need, device=dev, dtype=torch.float32)
                        _lib.check(self.lib.ssdhip_wino4f_input(_lib.ptr(cur), _lib.ptr(cur_xs[0]) if cur_xs is not None else None,
                                                                _lib.ptr(cur_xs[1]) if cur_xs is not None else None, _lib.ptr(vb), N, g['hi'], g['wi'],
                                                                c.cin, st), f'wino4f_input {c.name}')
                        if pool_fused:
                            rows = _lib.check(self.lib.ssdhip_wino4f_gemm_pool(_lib.ptr(vb), _lib.ptr(c.w4f_img), _lib.ptr(b1) if b1 is not None else None,
                                                                               _lib.ptr(zmm[0]), _lib.ptr(zmm[1]), _lib.ptr(ws['partial']) if training else None,
                                                                               N, g['hi'], g['wi'], c.cin, c.cout, st), f'conv2d {c.name}')
                        else:
                            rows = _lib.check(self.lib.ssdhip_wino4f_gemm(_lib.ptr(vb), _lib.ptr(c.w4f_img), _lib.ptr(b1) if b1 is not None else None,
                                                                          _lib.ptr(z), _lib.ptr(ws['partial']) if training else None,
                                                                          N, g['hi'], g['wi'], c.cin, c.cout, st), f'conv2d {c.name}')
                    elif pool_fused and zmm.shape[0] == 1:
                        rows = _lib.check(self.lib.ssdhip_wino_fused_conv_pool_sel(_lib.ptr(cur), _lib.ptr(c.wf_img), _lib.ptr(b1) if b1 is not None else None,
                                                                                   _lib.ptr(cur_xs[0]) if cur_xs is not None else None,
                                                                                   _lib.ptr(cur_xs[1]) if cur_xs is not None else None,
                                                                                   _lib.ptr(bnm.weight), _lib.ptr(zmm[0]), _lib.ptr(ws['partial']) if training else None,
                                                                                   _lib.ptr(self._wf_geo(b, N, g['hi'], g['wi'], dev)), N, g['hi'], g['wi'],
                                                                                   c.cin, c.cout, self._wf_blocks, st), f'conv2d {c.name}')
                    elif pool_fused:
                        rows = _lib.check(self.lib.ssdhip_wino_fused_conv_pool(_lib.ptr(cur), _lib.ptr(c.wf_img), _lib.ptr(b1) if b1 is not None else None,
                                                                               _lib.ptr(cur_xs[0]) if cur_xs is not None else None,
                                                                               _lib.ptr(cur_xs[1]) if cur_xs is not None else None,
                                                                               _lib.ptr(zmm[0]), _lib.ptr(zmm[1]), _lib.ptr(ws['partial']) if training else None,
                                                                               _lib.ptr(self._wf_geo(b, N, g['hi'], g['wi'], dev)), N, g['hi'], g['wi'],
                                                                               c.cin, c.cout, self._wf_blocks, st), f'conv2d {c.name}')
                    else:
                        rows = _lib.check(self.lib.ssdhip_wino_fused_conv(_lib.ptr(cur), _lib.ptr(c.wf_img), _lib.ptr(b1) if b1 is not None else None,
                                                                          _lib.ptr(cur_xs[0]) if cur_xs is not None else None,
                                                                          _lib.ptr(cur_xs[1]) if cur_xs is not None else None, _lib.ptr(z),
                                                                          _lib.ptr(ws['partial']) if training else None,
                                                                          _lib.ptr(self._wf_geo(b, N, g['hi'], g['wi'], dev)), N, g['hi'], g['wi'],
                                                                          c.cin, c.cout, self._wf_blocks, st), f'conv2d {c.name}')
                    if self.prof is not None and early_f4:
                        e1.record()
                        tiles = N * ((g['hi'] + 3) // 4) * ((g['wi'] + 3) // 4)
                        self.prof.append(dict(layer=c.name + ' (two-pass F(4x4))', e0=e0, e1=e1, kernel='wino4f_input_kernel + wino4f_gemm_kernel',
                                              flops=2.0 * 36 * tiles * c.cin * c.cout, direct_flops=2.0 * N * g['hi'] * g['wi'] * c.cin * c.cout * 9))
                    elif self.prof is not None:
                        e1.record()
                        tiles = N * ((g['hi'] + 1) // 2) * ((g['wi'] + 1) // 2)
                        self.prof.append(dict(layer=c.name + ' (fused Winograd)', e0=e0, e1=e1,
                                              kernel='wino_fused8_kernel<%s, 0, %d, %s>' % ('true' if cur_xs is not None else 'false', (2 if zmm.shape[0] == 1 else 1) if pool_fused else 0,
                                                                                        'true' if (cur_xs is not None and os.environ.get('SSD_WF_PRE') != '0') else 'false'),
                                              flops=2.0 * 16 * tiles * c.cin * c.cout,            # MFMA flops of the 16 Winograd-domain products
                                              direct_flops=2.0 * N * g['hi'] * g['wi'] * c.cin * c.cout * 9))
                elif c.wino and self._wino_ok(g) and not self.bf16:     # bf16 MFMA outruns the fp32 Winograd pipeline: direct kernel
                    rows = self._conv_winograd(c, cur, cur_xs, z, ws, N, g['hi'], g['wi'], ws['partial'] if training else None)
                elif (c.cin == 64 and c.cout in (64, 128) and c.k == 3 and c.stride == 1 and c.pad == 1 and min(g['hi'], g['wi']) >= 100
                      and not self.bf16):
                    # conv_1_2 / conv_2_1: K = 576 is too short for the staged kernel (103-108 TF); conv_c64.hip feeds the MFMAs from
                    # global memory with the weights resident in LDS (112-127 TF, scripts/c64_bench.py)
                    w1, b1 = self._weight(c)
                    key = (w1.data_ptr(), w1._version, self.m._hip_epoch if not c.frozen else 0)
                    if getattr(c, 'c64_img', None) is None or c.c64_img.device != w1.device:
                        c.c64_img = torch.empty(self.lib.ssdhip_conv_c64_weight_floats(c.cout), device=w1.device, dtype=torch.float32)
                        c.c64_key = None
                    if c.c64_key != key:
                        _lib.check(self.lib.ssdhip_conv_c64_pack(_lib.ptr(w1), _lib.ptr(c.c64_img), c.cout, st), f'conv_c64_pack {c.name}')
                        c.c64_key = key
                    rows = self.lib.ssdhip_conv_c64_stat_rows(c.cout)
                    _lib.check(self.lib.ssdhip_conv_c64(_lib.ptr(cur), _lib.ptr(c.c64_img), _lib.ptr(b1) if b1 is not None else None,
                                                        _lib.ptr(cur_xs[0]) if cur_xs is not None else None,
                                                        _lib.ptr(cur_xs[1]) if cur_xs is not None else None, _lib.ptr(z),
                                                        _lib.ptr(ws['partial']) if training else None, N, g['hi'], g['wi'], c.cout, st),
                               f'conv2d {c.name}')
                elif c.cin_pad == 4 and c.cout == 64 and c.k == 3 and c.stride == 1 and c.pad == 1:
                    # conv_1_1: a 737 MB store with 0.3 GFLOP per image -- its own HBM-bound kernel (conv_rgb.hip), unpacked weights
                    w1, b1 = self._weight(c)
                    rows = self.lib.ssdhip_conv_rgb3x3_stat_rows(N, g['hi'], g['wi'])
                    _lib.check(self.lib.ssdhip_conv_rgb3x3(_lib.ptr(cur), _lib.ptr(w1), _lib.ptr(b1) if b1 is not None else None, _lib.ptr(z),
                                                           _lib.ptr(ws['partial']) if training else None, N, g['hi'], g['wi'], c.cout, st),
                               f'conv2d {c.name}')
                else:
                    bm = self._conv(c, cur, cur_xs, z.data_ptr(), g['ho'] * g['wo'] * c.cout, N, g['hi'], g['wi'], g['ho'], g['wo'],
                                    ws['partial'] if training else None)
                    rows = (M + bm - 1) // bm
                if training and rows * c.cout * 2 > ws['partial'].numel():
                    # the kernels cannot check this themselves; an undersized table once corrupted its neighbours silently
                    raise RuntimeError(f'{c.name}: {rows} BatchNorm partial rows do not fit the workspace ({ws["partial"].numel()} floats)')
                if bb is not None and i == self.tail_start and not self._phased:
                    self._note_reader(i - 1, main.record_event())     # the backbone's last activation is consumed
                if training:
                    track = bnm.track_running_stats and bnm.running_mean is not None
                    mom = 0.1 if bnm.momentum is None else bnm.momentum
                    _lib.check(self.lib.ssdhip_bn_finalize(_lib.ptr(ws['partial']), rows, c.cout, M, _lib.ptr(bnm.weight),
                                                           _lib.ptr(bnm.bias), _lib.ptr(bnm.running_mean) if track else None,
                                                           _lib.ptr(bnm.running_var) if track else None, mom, bnm.eps,
                                                           _lib.ptr(bst['scale']), _lib.ptr(bst['shift']), _lib.ptr(bst['mean']),
                                                           _lib.ptr(bst['invstd']), _lib.ptr(ws['ws64']), st), f'batch_norm {s.bn_name}')
                    if track and bnm.num_batches_tracked is not None:
                        nbt.append(bnm.num_batches_tracked)
                else:
                    _lib.check(self.lib.ssdhip_bn_fold_eval(_lib.ptr(bnm.weight), _lib.ptr(bnm.bias), _lib.ptr(bnm.running_mean),
                                                            _lib.ptr(bnm.running_var), bnm.eps, c.cout, _lib.ptr(bst['scale']),
                                                            _lib.ptr(bst['shift']), st), f'batch_norm(eval) {s.bn_name}')
                act_xs = (bst['scale'], bst['shift'])       # act_L = relu(z * scale + shift), never materialised
                if nxt_patch:
                    # the next layer reads bf16: BN apply + ReLU (+ pool) into the interior of its haloed input, in one pass
                    abf = b.setdefault('abf', {}).get(i + 1)
                    if abf is None:
                        cn, gn = self.stages[i + 1].conv, geo[i + 1]
                        abf = torch.zeros(self.lib.ssdhip_conv_bf16_act_bytes(N, gn['hi'], gn['wi'], cn.cin) // 2, device=dev, dtype=torch.bfloat16)
                        b['abf'][i + 1] = abf
                    if alt and i + 1 == self.tail_start:
                        abf = self._alt(b, 'abf', i + 1, abf)        # zero halo like the first copy; the interior is rewritten every pass
                    _lib.check(self.lib.ssdhip_bn_relu_to_bf16(_lib.ptr(z), int(z_bf16), _lib.ptr(bst['scale']), _lib.ptr(bst['shift']), _lib.ptr(abf),
                                                               N, g['ho'], g['wo'], c.cout, int(s.pool_pad is not None), s.pool_pad or 0, st),
                               f'bn_relu_to_bf16 after {c.name}')
                    cur, cur_xs = abf, None
                elif pool_fused:
                    pooled = b['pooled'][i] if not alt else self._alt(b, 'pooled', i, b['pooled'][i])
                    zmm = b['zmm'][i]
                    if zmm.shape[0] == 1 and self._pool_xform and self._fused_ok(self.stages[i + 1].conv, geo[i + 1], N) and not self._early_f4:
                        # A/B (SSD_POOL_XFORM=1): no pooled activation in memory at all -- the next layer's fused kernel applies this
                        # BatchNorm.ReLU on load, like the layers without a pool in between (bit-identical: the same fma and max)
                        cur, cur_xs = zmm[0], act_xs
                    elif zmm.shape[0] == 1:     # the producer stored the extreme the pool uses: BatchNorm.ReLU of it IS the pooled activation
                        _lib.check(self.lib.ssdhip_bn_relu_pool(_lib.ptr(zmm[0]), _lib.ptr(bst['scale']), _lib.ptr(bst['shift']), _lib.ptr(pooled),
                                                                N, g['ha'], g['wa'], c.cout, 0, 0, st), f'pool (selected extreme) after {c.name}')
                        cur, cur_xs = pooled, None
                    else:
                        _lib.check(self.lib.ssdhip_bn_relu_select(_lib.ptr(zmm[0]), _lib.ptr(zmm[1]), _lib.ptr(bst['scale']), _lib.ptr(bst['shift']),
                                                                  _lib.ptr(pooled), N * g['ha'] * g['wa'], c.cout, st), f'pool (select) after {c.name}')
                        cur, cur_xs = pooled, None
                elif s.pool_pad is not None:
                    pooled = b['pooled'][i] if not alt else self._alt(b, 'pooled', i, b['pooled'][i])
                    _lib.check(self.lib.ssdhip_bn_relu_pool(_lib.ptr(z), _lib.ptr(bst['scale']), _lib.ptr(bst['shift']), _lib.ptr(pooled),
                                                            N, g['ho'], g['wo'], c.cout, 1, s.pool_pad, st), f'pool after {c.name}')
                    cur, cur_xs = pooled, None
                else:
                    cur, cur_xs = z, act_xs
                # what the backward reads as the next stage's input is always an fp32 tensor (+ scale / shift): where the next stage's
                # FORWARD took the haloed bf16 copy and is trainable (conv_6_1), its weight gradient still reads z and this BN state
                # (_patch_ok admits that case only without a pool in between)
                acts.append((z, act_xs, cur, cur_xs) if cur_xs is not None or cur.dtype == torch.float32 else (z, act_xs, z, act_xs))
                if s.head is not None and phase != 'backbone':
                    self._launch_head(i, s, g, b, z, act_xs, out, N, fork, bb if on_bb else None)
                    if on_bb:
                        self._note_reader(i, fork.mark())
        if phase == 'backbone':
            if nbt:
                torch._foreach_add_(nbt, 1)
            self._bb_ctx[parity] = dict(b=b, N=N, H=H, W=W, dev=dev, acts=acts, cur=cur, cur_xs=cur_xs, training=training, bf16=self.bf16)
            return None
        if bb is not None and self.tail_start >= len(self.stages):
            main.wait_event(bb.record_event())       # everything frozen (VGG16 transfer learning): the whole stack ran on the backbone stream
        fork.join()
        if nbt:
            torch._foreach_add_(nbt, 1)
        self._saved = dict(b=b, N=N, acts=acts, training=training, bf16=self.bf16) if save else None
        if self.headless:
            if cur_xs is not None:
                raise RuntimeError('a head-less feature stack must end in a pool (its BN.ReLU is applied there)')
            return cur          # (N, h, w, C) NHWC: the pooled last activation (a workspace tensor: consume it before the next forward)
        return out

    def _launch_head(self, i, s, g, b, z, act_xs, out, N, fork, after):
        """Detector head of stage i on the side stream: needs z and this stage's BN scale / shift only; writes its own rows of `out`."""
        off = b['head_off'][s.head_index]
        with fork.side(after=after):
            if i in b['head_f4'] and not self.bf16:
                self._head_winograd4(s.head, z, act_xs, b['head_v'][i], b, ctypes.c_void_p(out.data_ptr() + off * self.row * 4),
                                     b['P'] * self.row, N, g['ho'], g['wo'])
            elif i in b['head_v'] and not self.bf16:
                self._head_winograd(s.head, z, act_xs, b['head_v'][i], b, ctypes.c_void_p(out.data_ptr() + off * self.row * 4),
                                    b['P'] * self.row, N, g['ho'], g['wo'])
            else:
                self._conv(s.head, z, act_xs, out.data_ptr() + off * self.row * 4, b['P'] * self.row, N, g['ho'], g['wo'],
                           g['ho'], g['wo'], None)

    # ------------------------------------------------------------------------------------------------ backward
    def grad_layout(self):
        """(names, params, offsets, total): flat gradient buffer layout in SSD.train_params() order, 16-B aligned."""
        names, params, offs, p = [], [], [], 0
        for n, q in self.m.named_train_params():
            names.append(n); params.append(q); offs.append(p)
            p += _rup(q.numel(), 4)
        return names, params, offs, p

    def grad_cut(self):
        """Offset in the flat gradient buffer behind the first tail stage's parameters: everything from here on is final before
        that stage's weight gradient (the largest, last) runs, so the all-reduce of [cut, end) overlaps it."""
        names, _, offs, _ = self.grad_layout()
        return offs[names.index(f'features.{self.stages[self.tail_start + 1].conv.name}.weight')]

    def _flat_grad(self, dev):
        names, params, offs, total = self.grad_layout()
        if self._flat is None or self._flat['buf'].device != dev or self._flat['total'] != total:
            self._flat = dict(buf=torch.zeros(total, device=dev, dtype=torch.float32), total=total, names=names, offs=offs)
        views = {n: self._flat['buf'][o:o + q.numel()].view(q.shape) for n, q, o in zip(names, params, offs)}
        return self._flat['buf'], views, params, names

    def _bwd_buffers(self, sv):
        b, N = sv['b'], sv['N']
        if 'bwd' in b:
            return b['bwd']
        dev = b['x4'].device
        f = dict(device=dev, dtype=torch.float32)
        geo = b['geo']
        bw = dict(da={}, dz={}, dyh={}, dpool={})
        max_c, ws_bytes = 0, 0
        for i in range(self.tail_start, len(self.stages)):
            s, g = self.stages[i], geo[i]
            bw['da'][i] = torch.empty(N, g['ho'], g['wo'], s.conv.cout, **f)
            bw['dz'][i] = torch.empty(N, g['ho'], g['wo'], s.conv.cout, **f)
            if s.pool_pad is not None:
                bw['dpool'][i] = torch.empty(N, g['ha'], g['wa'], s.conv.cout, **f)
            max_c = max(max_c, s.conv.cout)
        for i, (s, g) in enumerate(zip(self.stages, geo)):
            if s.head is not None:
                bw['dyh'][i] = torch.zeros(N, g['ho'] * g['wo'], s.head.co_pad, **f)
        bw['partial'] = torch.empty(512 * max_c * 2, **f)
        bw['coef'] = torch.empty(max_c * 2, **f)
        # wgrad split-K workspace: query every layer once
        for i, (s, g) in enumerate(zip(self.stages, geo)):
            for spec, (hi, wi, ho, wo) in ((s.conv, (g['hi'], g['wi'], g['ho'], g['wo'])), (s.head, (g['ho'], g['wo'], g['ho'], g['wo']))):
                if spec is None or spec.frozen:
                    continue
                p = self._wgrad_params(spec, N, hi, wi, ho, wo)
                ws_bytes = max(ws_bytes, _lib.check(self.lib.ssdhip_conv2d_wgrad_ws_bytes(ctypes.byref(p)), 'wgrad_ws_bytes'))    # direct (bf16 mode: always)
                if self._wgrad_wino(spec, hi, wi):
                    ws_bytes = max(ws_bytes, _lib.check(self.lib.ssdhip_conv2d_wgrad_winograd_ws_bytes(ctypes.byref(p)), 'wgrad_ws_bytes'))
                if self._wgrad_wino4(spec, hi, wi):
                    ws_bytes = max(ws_bytes, _lib.check(self.lib.ssdhip_conv2d_wgrad_winograd4_ws_bytes(ctypes.byref(p)), 'wgrad_ws_bytes'))
        bw['wgrad_ws'] = torch.empty((ws_bytes + 3) // 4, **f)
        dg_v = dg_m = 0             # Winograd data gradients (un-frozen backbone): dy planes V and dx planes M
        for i in range(self.tail_start + 1, len(self.stages)):
            s, g = self.stages[i], geo[i]
            if self._dgrad_wino(s.conv, g):
                tp = (N * ((g['ho'] + 1) // 2) * ((g['wo'] + 1) // 2) + 127) // 128 * 128
                dg_v, dg_m = max(dg_v, 16 * tp * s.conv.cout), max(dg_m, 16 * tp * s.conv.cin)
        bw['dg_v'] = torch.empty(dg_v, **f) if dg_v else None
        bw['dg_m'] = torch.empty(dg_m, **f) if dg_m else None
        d4_v = d4_m = 0             # F(4x4) data gradients of detector heads (det_7_1): dy planes V and dx planes M
        for i in range(self.tail_start, len(self.stages)):
            s, g = self.stages[i], geo[i]
            if s.head is not None and self._head_dgrad_wino4(s.head, g['ho'], g['wo']):
                tp = _rup(N * ((g['ho'] + 3) // 4) * ((g['wo'] + 3) // 4), 128)
                d4_v, d4_m = max(d4_v, 36 * tp * s.head.co_pad), max(d4_m, 36 * tp * s.head.cin)
        bw['d4_v'] = torch.empty(d4_v, **f) if d4_v else None
        bw['d4_m'] = torch.empty(d4_m, **f) if d4_m else None
        b['bwd'] = bw
        return bw

    def _wgrad_wino4(self, spec, hi, wi):
        """Winograd F(4x4,3x3) weight gradient (36 planes of a quarter as many tiles, dW = G^T dU G in fp64: error against fp64 3.6e-7 ..
        8.5e-7 of max |dW|, the class of the F(2x2) form and of torch's fp32): conv_6_1 and the heads on >= 16-px maps."""
        return self._wino_mode == 'f4' and self._wgrad_wino(spec, hi, wi) and min(hi, wi) >= 16

    @staticmethod
    def _wgrad_wino(spec, hi, wi):
        """Winograd F(2x2,3x3) weight gradient: measured x1.2 .. x1.8 for conv_6_1 and the heads on >= 10x10 maps
        (scripts/conv_bench.py with BWD=1); tiny maps and narrow inputs stay on the direct kernel."""
        return spec.k == 3 and spec.stride == 1 and spec.pad == 1 and spec.cin >= 256 and min(hi, wi) >= 10

    def _wgrad_params(self, spec, N, hi, wi, ho, wo):
        p = _lib.WgradParams()
        p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = N, hi, wi, spec.cin_pad, ho, wo, spec.cout
        p.dw_cin = spec.cin if spec.cin_pad != spec.cin else 0          # RGB input is padded to 4 channels, the parameter is not
        p.R, p.S, p.stride, p.pad = spec.k, spec.k, spec.stride, spec.pad
        p.dy_row = spec.co_pad if spec.is_head else spec.cout
        p.dy_batch_stride = ho * wo * p.dy_row
        return p

    def _wgrad(self, spec, x, xs, dy, views, bw, N, hi, wi, ho, wo, wino_v=None):
        p = self._wgrad_params(spec, N, hi, wi, ho, wo)
        p.wino_v = wino_v.data_ptr() if wino_v is not None else None
        prefix = ('detectors.' if spec.is_head else 'features.') + spec.name
        p.x, p.dy = x.data_ptr(), dy.data_ptr()
        p.in_scale = xs[0].data_ptr() if xs is not None else None
        p.in_shift = xs[1].data_ptr() if xs is not None else None
        p.dw_oihw = views[prefix + '.weight'].data_ptr()
        # conv bias in front of a train-mode BatchNorm: d bias = sum(dz) = 0 exactly (BN removes the batch mean), the
        # reference's autograd holds rounding noise there; the flat buffer slot stays 0 (zeroed at allocation)
        p.dbias = views[prefix + '.bias'].data_ptr() if (spec.is_head and (prefix + '.bias') in views) else None
        p.ws = bw['wgrad_ws'].data_ptr()
        # bf16 mode: operands rounded to bf16 at staging, 16x the matrix rate -- the direct kernel outruns the fp32 Winograd form
        p.compute = int(self.bf16)
        if self._wgrad_wino4(spec, hi, wi) and not self.bf16:
            fn = self.lib.ssdhip_conv2d_wgrad_winograd4      # heads: wino_v is the forward's F(4x4) input transform (_buffers: head_f4)
        elif self._wgrad_wino(spec, hi, wi) and not self.bf16:
            fn = self.lib.ssdhip_conv2d_wgrad_winograd
        else:
            fn = self.lib.ssdhip_conv2d_wgrad
        _lib.check(fn(ctypes.byref(p), _lib.stream()), f'wgrad {spec.name}')

    def _head_dgrad_wino4(self, spec, h, w):
        """Data gradient of a detector head on a >= 16-px map through the F(4x4) passes (the adjoint convolution of dy): det_7_1 of SSD300
        (19x19, 1024 <- 150 channels: 322 us on the implicit-GEMM gather kernel, 36 of every 144 of its MFMA flops needed here)."""
        return (self._wino_mode == 'f4' and not self.bf16 and spec.is_head and spec.k == 3 and spec.stride == 1 and spec.pad == 1
                and spec.cin >= 256 and spec.cin % 4 == 0 and 256 % (spec.cin // 4) == 0 and spec.co_pad % 32 == 0 and min(h, w) >= 16)

    def _head_dgrad_winograd4(self, spec, dy, dx, bw, N, h, w, accumulate):
        L, st = self.lib, _lib.stream()
        wt, _ = self._weight(spec)
        key = (wt.data_ptr(), wt._version, self.m._hip_epoch)
        rows = _rup(spec.cin, 128)
        if getattr(spec, 'd4_u', None) is None or spec.d4_u.device != wt.device:
            spec.d4_u = torch.empty(36 * rows * spec.co_pad, device=wt.device, dtype=torch.float32)
            spec.d4_key = None
        if spec.d4_key != key:
            _lib.check(L.ssdhip_winograd4_weight_adjoint(_lib.ptr(wt), _lib.ptr(spec.d4_u), spec.cout, spec.cin, rows, spec.co_pad, st),
                       f'winograd4_weight_adjoint {spec.name}')
            spec.d4_key = key
        V, Mo = bw['d4_v'], bw['d4_m']
        tp = _rup(N * ((h + 3) // 4) * ((w + 3) // 4), 128)
        _lib.check(L.ssdhip_winograd4_input(_lib.ptr(dy), None, None, _lib.ptr(V), N, h, w, spec.co_pad, tp, st), f'winograd4_input (dgrad {spec.name})')
        p = _lib.ConvParams()
        p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), spec.d4_u.data_ptr(), Mo.data_ptr(), tp * spec.cin
        p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 36, 1, tp, spec.co_pad, 1, tp, spec.cin
        p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate = 1, 1, 1, 0, spec.co_pad, 0, 0
        p.tile = 128128
        p.rows_per_img, p.w_img_stride = tp, rows * spec.co_pad
        self._plane_gemm(p, tp, spec.co_pad, spec.cin, rows * spec.co_pad, f'winograd4-gemm-dgrad {spec.name}', planes=36, spec=spec)
        _lib.check(L.ssdhip_winograd4_output(_lib.ptr(Mo), None, _lib.ptr(dx), None, int(accumulate), N, h, w, spec.cin, tp, st),
                   f'winograd4_output (dgrad {spec.name})')

    def _dgrad_wino(self, spec, g):
        """Data gradient of a 3x3/s1/p1 layer = the same convolution of dy with the flipped, transposed filter: it runs through the
        forward's Winograd pipeline where that pays (the un-frozen backbone's 75x75 .. 19x19 layers; the tail's maps are too small)."""
        return (self.m._tune_all and not self.bf16 and not spec.is_head and spec.k == 3 and spec.stride == 1 and spec.pad == 1
                and spec.cout % 32 == 0 and spec.cout >= 128 and spec.cin % 4 == 0 and 256 % (spec.cin // 4) == 0 and spec.cin >= 64
                and 19 <= g['hi'] <= 160 and 19 <= g['wi'] <= 160)

    def _dgrad_winograd(self, spec, dy, dx, bw, N, h, w):
        L, st = self.lib, _lib.stream()
        wt, _ = self._weight(spec)
        key = (wt.data_ptr(), wt._version, self.m._hip_epoch)
        rows = _rup(spec.cin, 128)
        if getattr(spec, 'dg_u', None) is None or spec.dg_u.device != wt.device:
            spec.dg_u = torch.empty(16 * rows * spec.cout, device=wt.device, dtype=torch.float32)
            spec.dg_key = None
        if spec.dg_key != key:
            flipped = wt.detach().flip(2, 3).transpose(0, 1).contiguous()          # (Cin, Cout, 3, 3): filter of the adjoint convolution
            _lib.check(L.ssdhip_winograd_weight(_lib.ptr(flipped), _lib.ptr(spec.dg_u), spec.cin, spec.cout, rows, spec.cout, st), 'winograd_weight (dgrad)')
            spec.dg_key = key
        V, Mo = bw['dg_v'], bw['dg_m']
        tp = (N * ((h + 1) // 2) * ((w + 1) // 2) + 127) // 128 * 128
        _lib.check(L.ssdhip_winograd_input(_lib.ptr(dy), None, None, _lib.ptr(V), N, h, w, spec.cout, st), f'winograd_input (dgrad {spec.name})')
        p = _lib.ConvParams()
        p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), spec.dg_u.data_ptr(), Mo.data_ptr(), tp * spec.cin
        p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 16, 1, tp, spec.cout, 1, tp, spec.cin
        p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate = 1, 1, 1, 0, spec.cout, 0, 0
        p.tile = 128128 if (16 * tp // 128) * ((spec.cin + 127) // 128) > 2000 else 64128
        if spec.cin < 128:
            p.tile = 128064
        p.rows_per_img, p.w_img_stride = tp, rows * spec.cout
        self._plane_gemm(p, tp, spec.cout, spec.cin, rows * spec.cout, f'winograd-gemm-dgrad {spec.name}', spec=spec)
        _lib.check(L.ssdhip_winograd_output(_lib.ptr(Mo), None, _lib.ptr(dx), None, N, h, w, spec.cin, st), f'winograd_output (dgrad {spec.name})')

    def _dgrad(self, spec, dy, dx, N, hdy, wdy, hin, win, accumulate):
        w = self._pack(spec, dgrad=True)
        p = _lib.ConvParams()
        p.x, p.w, p.bias, p.in_scale, p.in_shift = dy.data_ptr(), w.data_ptr(), None, None, None
        p.y, p.stats, p.y_batch_stride = dx.data_ptr(), None, hin * win * spec.cin
        p.N, p.Hi, p.Wi, p.Cin = N, hdy, wdy, spec.co_pad
        p.Ho, p.Wo, p.Cout = hin, win, spec.cin
        p.R, p.S, p.stride, p.pad = spec.k, spec.k, spec.stride, spec.pad
        p.KP, p.mode, p.accumulate, p.tile = spec.KPd, 1, int(accumulate), 0
        p.compute = int(self.bf16)
        self._splitk_ws(spec, p, ('dgrad', N, hdy, wdy))
        _lib.check(self.lib.ssdhip_conv2d_nhwc(ctypes.byref(p), _lib.stream()), f'dgrad {spec.name}')

    def backward(self, grad_out):
        sv = self._saved
        if sv is None:
            raise RuntimeError('backward called without a saved forward (forward ran under no_grad or was already consumed)')
        if not sv['training']:
            raise RuntimeError('backward through eval-mode BatchNorm is not implemented (the reference never calls eval())')
        self._saved = None
        self.bf16 = sv['bf16']
        b, N, acts = sv['b'], sv['N'], sv['acts']
        geo = b['geo']
        _lib.require_gpu(grad_out)
        grad_out = _lib.f32c(grad_out)
        dev = grad_out.device
        bw = self._bwd_buffers(sv)
        flat, views, params, names = self._flat_grad(dev)
        st = _lib.stream()
        L = self.lib
        # dense (N,P,row) gradient -> channel-padded per-head NHWC gradients
        heads = [(i, s, g) for i, (s, g) in enumerate(zip(self.stages, geo)) if s.head is not None]
        sig = tuple(bw['dyh'][i].data_ptr() for i, _, _ in heads)
        if len(heads) > 1:            # one launch for all heads (device job table, rebuilt only when a gradient buffer moves)
            tbl = bw.get('scatter_table')
            if tbl is None or tbl[0] != sig:
                import numpy as np
                rec = np.zeros(len(heads), dtype=np.dtype([('out', '<u8'), ('p', '<i4', (4,))]))          # ssdhip_scatter_job: 24 bytes
                for r, (i, s, g) in zip(rec, heads):
                    r['out'], r['p'] = bw['dyh'][i].data_ptr(), (b['head_off'][s.head_index], g['ho'] * g['wo'], s.head.cout, s.head.co_pad)
                tbl = bw['scatter_table'] = (sig, torch.from_numpy(rec.view(np.uint8).copy()).to(grad_out.device),
                                            max(N * g['ho'] * g['wo'] * s.head.co_pad for _, s, g in heads))
            _lib.check(L.ssdhip_scatter_head_grads(_lib.ptr(grad_out), _lib.ptr(tbl[1]), len(heads), N, b['P'], self.row, tbl[2], st), 'scatter_head_grads')
        else:
            for i, s, g in heads:
                _lib.check(L.ssdhip_scatter_head_grad(_lib.ptr(grad_out), _lib.ptr(bw['dyh'][i]), N, b['P'], self.row,
                                                      b['head_off'][s.head_index], g['ho'] * g['wo'], s.head.cout, s.head.co_pad, st),
                           'scatter_head_grad')
        if self.headless:
            # grad_out = gradient of the pooled last activation (N, ha, wa, C) NHWC: through BN.ReLU.MaxPool into da[last]
            last_i = len(self.stages) - 1
            ls, lg = self.stages[last_i], geo[last_i]
            lst = self._bn[ls.bn_name]
            if tuple(grad_out.shape) != (N, lg['ha'], lg['wa'], ls.conv.cout):
                raise ValueError(f'backward: expected a gradient of shape {(N, lg["ha"], lg["wa"], ls.conv.cout)}, got {tuple(grad_out.shape)}')
            _lib.check(L.ssdhip_bn_relu_pool_bwd(_lib.ptr(b['z'][last_i]), _lib.ptr(lst['scale']), _lib.ptr(lst['shift']), _lib.ptr(grad_out),
                                                 _lib.ptr(bw['da'][last_i]), N, lg['ho'], lg['wo'], ls.conv.cout, ls.pool_pad, st),
                       f'pool backward after {ls.conv.name}')
        # Weight gradients are leaves of the dependency graph (nothing downstream reads them before the optimizer): they all go
        # to the side stream, each after the main stream has produced its dy, and run next to the BN / data-gradient chain.
        fork = self._Fork(self, dev)
        # heads on frozen activations (det_4_3): weight gradient only
        for i in range(self.tail_start):
            s, g = self.stages[i], geo[i]
            if s.head is not None:
                x, xs = acts[i][0], acts[i][1]
                with fork.side():
                    self._wgrad(s.head, x, xs, bw['dyh'][i], views, bw, N, g['ho'], g['wo'], g['ho'], g['wo'], None if self.bf16 else b['head_v'].get(i))
                if not self._phased:
                    self._note_reader(i, fork.mark())    # the next backbone pass may overwrite z[i] / its BN state after this
        last = len(self.stages) - 1
        for i in range(last, self.tail_start - 1, -1):
            s, g = self.stages[i], geo[i]
            c = s.conv
            da, dz = bw['da'][i], bw['dz'][i]
            x_act, x_xs = acts[i][0], acts[i][1]        # this stage's activation (input of its head)
            if s.head is not None:
                with fork.side():
                    self._wgrad(s.head, x_act, x_xs, bw['dyh'][i], views, bw, N, g['ho'], g['wo'], g['ho'], g['wo'], None if self.bf16 else b['head_v'].get(i))
                if self._head_dgrad_wino4(s.head, g['ho'], g['wo']) and bw['d4_v'] is not None:
                    self._head_dgrad_winograd4(s.head, bw['dyh'][i], da, bw, N, g['ho'], g['wo'], accumulate=(i != last))
                else:
                    self._dgrad(s.head, bw['dyh'][i], da, N, g['ho'], g['wo'], g['ho'], g['wo'], accumulate=(i != last))
            # BatchNorm(train) + ReLU backward: da -> dz, d gamma, d beta
            bst = self._bn[s.bn_name]
            M = N * g['ho'] * g['wo']
            mb = max(1, min(512, (M + 127) // 128))
            z = b['z'][i]
            _lib.check(L.ssdhip_bn_relu_bwd_reduce(_lib.ptr(z), _lib.ptr(da), _lib.ptr(bst['scale']), _lib.ptr(bst['shift']),
                                                   _lib.ptr(bst['mean']), _lib.ptr(bst['invstd']), M, c.cout, _lib.ptr(bw['partial']),
                                                   mb, st), 'bn_relu_bwd_reduce')
            _lib.check(L.ssdhip_bn_relu_bwd_finalize(_lib.ptr(bw['partial']), mb, c.cout, M,
                                                     _lib.ptr(views[f'features.{s.bn_name}.weight']),
                                                     _lib.ptr(views[f'features.{s.bn_name}.bias']), _lib.ptr(bw['coef']),
                                                     _lib.ptr(b['ws'][1]['ws64']), st), 'bn_relu_bwd_finalize')
            _lib.check(L.ssdhip_bn_relu_bwd_apply(_lib.ptr(z), _lib.ptr(da), _lib.ptr(bst['scale']), _lib.ptr(bst['shift']),
                                                  _lib.ptr(bst['mean']), _lib.ptr(bst['invstd']), _lib.ptr(bw['coef']), M, c.cout,
                                                  _lib.ptr(dz), st), 'bn_relu_bwd_apply')
            xin, xin_xs = (acts[i - 1][2], acts[i - 1][3]) if i > 0 else (b['x4'], None)
            if i == self.tail_start and self.dist_hook is not None:
                # every gradient behind the first stage's parameters is final: reduce it while the largest wgrad runs
                cut = self.grad_cut()
                assert self._flat['offs'][names.index(f'features.{c.name}.weight')] == 0
                fork.join()
                self.dist_hook.start(flat, cut, flat.numel())
            with fork.side():
                self._wgrad(c, xin, xin_xs, dz, views, bw, N, g['hi'], g['wi'], g['ho'], g['wo'])
            if i == self.tail_start and i > 0 and not self._phased:
                self._note_reader(i - 1, fork.mark())   # conv_6_1's weight gradient reads the backbone's last activation
            if i > self.tail_start:
                prev, gp = self.stages[i - 1], geo[i - 1]
                target = bw['da'][i - 1] if prev.pool_pad is None else bw['dpool'][i - 1]
                if self._dgrad_wino(c, g) and bw['dg_v'] is not None:
                    self._dgrad_winograd(c, dz, target, bw, N, g['hi'], g['wi'])
                else:
                    self._dgrad(c, dz, target, N, g['ho'], g['wo'], g['hi'], g['wi'], accumulate=False)
                if prev.pool_pad is not None:       # un-frozen backbone only: gradient of the pooled map, then through BN.ReLU.MaxPool
                    pst = self._bn[prev.bn_name]
                    _lib.check(L.ssdhip_bn_relu_pool_bwd(_lib.ptr(b['z'][i - 1]), _lib.ptr(pst['scale']), _lib.ptr(pst['shift']),
                                                         _lib.ptr(bw['dpool'][i - 1]), _lib.ptr(bw['da'][i - 1]), N, gp['ho'], gp['wo'],
                                                         prev.conv.cout, prev.pool_pad, st), f'pool backward after {prev.conv.name}')
        fork.join()
        if self.dist_hook is not None:
            self.dist_hook.finish(flat)
        return [views[n] for n in names], params
