"""Functional (allocate-and-call) wrappers over single libssdhip.so entry points.  The engine drives the same entry
points with pre-allocated buffers; these exist for per-kernel parity tests, profiling and ad-hoc use."""
import ctypes

import torch

from . import _lib


def _rup(v, m):
    return (v + m - 1) // m * m


def pack_weight(w_oihw, mode=0):
    """OIHW -> packed implicit-GEMM operand.  Returns (packed, KP, inner_pad)."""
    _lib.require_gpu(w_oihw)
    w = _lib.f32c(w_oihw)
    cout, cin, k, _ = w.shape
    if mode == 0:
        inner = 4 if cin == 3 else cin
        rows = _rup(cout, 128)
    else:
        inner = _rup(cout, 32)
        rows = _rup(cin, 128)
    KP = _rup(k * k * inner, 32)
    packed = torch.empty(rows * KP, device=w.device, dtype=torch.float32)
    _lib.check(_lib.load().ssdhip_pack_conv_weight(_lib.ptr(w), _lib.ptr(packed), cout, cin, k, k, rows, inner, KP, mode, _lib.stream()),
               'pack_conv_weight')
    return packed, KP, inner


def _splitk_ws(lib, p, split_k):
    """Workspace for the split-K form of small launches (tiny maps); the returned tensor must outlive the launch."""
    need = lib.ssdhip_conv2d_ws_bytes(ctypes.byref(p)) if split_k else 0
    if need <= 0:
        return None
    ws = torch.empty((need + 3) // 4, device=torch.device('cuda', torch.cuda.current_device()), dtype=torch.float32)
    p.ws, p.ws_bytes = ws.data_ptr(), ws.numel() * 4
    return ws


def conv2d_nhwc(x, w_oihw, bias=None, stride=1, pad=0, in_scale=None, in_shift=None, want_stats=False, tile=0, bf16=False, split_k=True):
    """x (N,H,W,Cin_pad) fp32 NHWC -> (z (N,Ho,Wo,Cout), stats or None).  stats: (MB, Cout, 2) partial sums.
    bf16: operands rounded to bf16 at staging, bf16 MFMA with fp32 accumulate (tensors stay fp32).  split_k: let launches whose
    tile grid cannot fill the chip split K over blocks (fixed-order second pass; same result up to the summation order)."""
    _lib.require_gpu(x, w_oihw)
    lib = _lib.load()
    N, H, W, C = x.shape
    cout, cin, k, _ = w_oihw.shape
    packed, KP, inner = pack_weight(w_oihw, 0)
    if inner != C:
        raise ValueError(f'input has {C} channels, weights expect {inner} (pad the 3-channel image to 4)')
    ho, wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    z = torch.empty(N, ho, wo, cout, device=x.device, dtype=torch.float32)
    p = _lib.ConvParams()
    p.x, p.w, p.bias = x.data_ptr(), packed.data_ptr(), (bias.data_ptr() if bias is not None else None)
    p.in_scale = in_scale.data_ptr() if in_scale is not None else None
    p.in_shift = in_shift.data_ptr() if in_shift is not None else None
    p.y, p.y_batch_stride = z.data_ptr(), ho * wo * cout
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = N, H, W, C, ho, wo, cout
    p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = k, k, stride, pad, KP, 0, 0, tile
    p.compute = int(bool(bf16))
    stats = None
    if want_stats:
        bm = lib.ssdhip_conv2d_block_m(ctypes.byref(p))
        stats = torch.empty((N * ho * wo + bm - 1) // bm, cout, 2, device=x.device, dtype=torch.float32)
        p.stats = stats.data_ptr()
    ws = _splitk_ws(lib, p, split_k)
    _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), _lib.stream()), 'conv2d_nhwc')
    return z, stats


def conv_rgb3x3(x4, w_oihw, bias=None, want_stats=False):
    """conv_1_1: x4 (N,H,W,4) fp32 (RGB padded to 4), w (64,3,3,3) -> (z (N,H,W,64), per-wave partial stats or None)."""
    _lib.require_gpu(x4, w_oihw)
    lib = _lib.load()
    N, H, W, C = x4.shape
    if C != 4 or tuple(w_oihw.shape[1:]) != (3, 3, 3):
        raise ValueError(f'conv_rgb3x3: expected (N,H,W,4) input and (Cout,3,3,3) weights, got {tuple(x4.shape)} {tuple(w_oihw.shape)}')
    cout = w_oihw.shape[0]
    z = torch.empty(N, H, W, cout, device=x4.device, dtype=torch.float32)
    stats = torch.empty(lib.ssdhip_conv_rgb3x3_stat_rows(N, H, W), cout, 2, device=x4.device, dtype=torch.float32) if want_stats else None
    _lib.check(lib.ssdhip_conv_rgb3x3(_lib.ptr(x4), _lib.ptr(_lib.f32c(w_oihw)), _lib.ptr(bias) if bias is not None else None, _lib.ptr(z),
                                      _lib.ptr(stats) if stats is not None else None, N, H, W, cout, _lib.stream()), 'conv_rgb3x3')
    return z, stats


def conv_c64(x, w_oihw, bias=None, in_scale=None, in_shift=None, want_stats=False):
    """3x3/s1/p1 conv with Cin = 64 (conv_1_2 / conv_2_1 kernel): x (N,H,W,64) -> (z (N,H,W,Cout), per-wave partial stats or None)."""
    _lib.require_gpu(x, w_oihw)
    lib = _lib.load()
    N, H, W, C = x.shape
    cout = w_oihw.shape[0]
    if C != 64 or tuple(w_oihw.shape[1:]) != (64, 3, 3):
        raise ValueError(f'conv_c64: expected (N,H,W,64) input and (Cout,64,3,3) weights, got {tuple(x.shape)} {tuple(w_oihw.shape)}')
    wimg = torch.empty(_lib.check(lib.ssdhip_conv_c64_weight_floats(cout), 'conv_c64_weight_floats'), device=x.device, dtype=torch.float32)
    _lib.check(lib.ssdhip_conv_c64_pack(_lib.ptr(_lib.f32c(w_oihw)), _lib.ptr(wimg), cout, _lib.stream()), 'conv_c64_pack')
    z = torch.empty(N, H, W, cout, device=x.device, dtype=torch.float32)
    stats = torch.empty(lib.ssdhip_conv_c64_stat_rows(cout), cout, 2, device=x.device, dtype=torch.float32) if want_stats else None
    _lib.check(lib.ssdhip_conv_c64(_lib.ptr(x), _lib.ptr(wimg), _lib.ptr(bias) if bias is not None else None,
                                   _lib.ptr(in_scale) if in_scale is not None else None, _lib.ptr(in_shift) if in_shift is not None else None,
                                   _lib.ptr(z), _lib.ptr(stats) if stats is not None else None, N, H, W, cout, _lib.stream()), 'conv_c64')
    return z, stats


def wino_fused_pack(w_oihw):
    """OIHW (Cout % 64 == 0, Cin % 8 == 0, 3, 3) -> the fused Winograd kernel's weight image (one LDS-ready 32 KiB block per
    64-channel tile and 8-channel K-step)."""
    _lib.require_gpu(w_oihw)
    lib = _lib.load()
    w = _lib.f32c(w_oihw)
    cout, cin = w.shape[0], w.shape[1]
    if tuple(w.shape[2:]) != (3, 3):
        raise ValueError(f'wino_fused_pack: 3x3 filters only, got {tuple(w.shape)}')
    upk = torch.empty(_lib.check(lib.ssdhip_wino_fused_weight_floats(cout, cin), 'wino_fused_weight_floats'), device=w.device, dtype=torch.float32)
    _lib.check(lib.ssdhip_wino_fused_pack(_lib.ptr(w), _lib.ptr(upk), cout, cin, _lib.stream()), 'wino_fused_pack')
    return upk


_wf_geo = {}


def wino_fused_geo(N, H, W, device):
    """Tile-block geometry table of an (N, H, W) map for the fused Winograd kernel (cached per shape and device)."""
    key = (N, H, W, str(device))
    g = _wf_geo.get(key)
    if g is None:
        lib = _lib.load()
        g = torch.empty(_lib.check(lib.ssdhip_wino_fused_geo_bytes(N, H, W), 'wino_fused_geo_bytes'), device=device, dtype=torch.uint8)
        _lib.check(lib.ssdhip_wino_fused_geo(N, H, W, _lib.ptr(g), _lib.stream()), 'wino_fused_geo')
        if len(_wf_geo) >= 8:
            _wf_geo.pop(next(iter(_wf_geo)))
        _wf_geo[key] = g
    return g


def wino_fused_conv(x, w_oihw, bias=None, in_scale=None, in_shift=None, want_stats=False, upk=None, max_blocks=0):
    """3x3/s1/p1 conv as ONE fused Winograd F(2x2,3x3) kernel: x (N,H,W,Cin) NHWC -> (z (N,H,W,Cout), stats (MB,Cout,2) or None)."""
    _lib.require_gpu(x, w_oihw)
    lib = _lib.load()
    N, H, W, C = x.shape
    cout, cin = w_oihw.shape[0], w_oihw.shape[1]
    if cin != C:
        raise ValueError(f'wino_fused_conv: input has {C} channels, weights expect {cin}')
    if upk is None:
        upk = wino_fused_pack(w_oihw)
    z = torch.empty(N, H, W, cout, device=x.device, dtype=torch.float32)
    stats = torch.empty(_lib.check(lib.ssdhip_wino_fused_stat_rows(N, H, W), 'wino_fused_stat_rows'), cout, 2, device=x.device,
                        dtype=torch.float32) if want_stats else None
    _lib.check(lib.ssdhip_wino_fused_conv(_lib.ptr(x), _lib.ptr(upk), _lib.ptr(bias) if bias is not None else None,
                                          _lib.ptr(in_scale) if in_scale is not None else None,
                                          _lib.ptr(in_shift) if in_shift is not None else None, _lib.ptr(z),
                                          _lib.ptr(stats) if stats is not None else None, _lib.ptr(wino_fused_geo(N, H, W, x.device)),
                                          N, H, W, C, cout, max_blocks, _lib.stream()), 'wino_fused_conv')
    return z, stats


def wino_fused_conv_pool(x, w_oihw, bias=None, in_scale=None, in_shift=None, want_stats=False, upk=None, max_blocks=0):
    """The fused Winograd convolution when a 2x2 / stride 2 max-pool follows: returns (zmax, zmin (N,H/2,W/2,Cout), stats or None)."""
    _lib.require_gpu(x, w_oihw)
    lib = _lib.load()
    N, H, W, C = x.shape
    cout, cin = w_oihw.shape[0], w_oihw.shape[1]
    if cin != C or H % 2 or W % 2:
        raise ValueError('wino_fused_conv_pool: channel mismatch or odd map size')
    if upk is None:
        upk = wino_fused_pack(w_oihw)
    zmax = torch.empty(N, H // 2, W // 2, cout, device=x.device, dtype=torch.float32)
    zmin = torch.empty_like(zmax)
    stats = torch.empty(_lib.check(lib.ssdhip_wino_fused_stat_rows(N, H, W), 'wino_fused_stat_rows'), cout, 2, device=x.device,
                        dtype=torch.float32) if want_stats else None
    _lib.check(lib.ssdhip_wino_fused_conv_pool(_lib.ptr(x), _lib.ptr(upk), _lib.ptr(bias) if bias is not None else None,
                                               _lib.ptr(in_scale) if in_scale is not None else None,
                                               _lib.ptr(in_shift) if in_shift is not None else None, _lib.ptr(zmax), _lib.ptr(zmin),
                                               _lib.ptr(stats) if stats is not None else None, _lib.ptr(wino_fused_geo(N, H, W, x.device)),
                                               N, H, W, C, cout, max_blocks, _lib.stream()), 'wino_fused_conv_pool')
    return zmax, zmin, stats


def wino_fused_conv_pool_sel(x, w_oihw, bn_weight, bias=None, in_scale=None, in_shift=None, want_stats=False, upk=None, max_blocks=0):
    """As wino_fused_conv_pool, but only the extreme the pool will use is stored: per window the maximum of z where bn_weight[co] >= 0, else
    the minimum.  Returns (zsel (N,H/2,W/2,Cout), stats or None); bn_relu_pool(zsel, scale, shift, pool=False) is then the pooled activation."""
    _lib.require_gpu(x, w_oihw, bn_weight)
    lib = _lib.load()
    N, H, W, C = x.shape
    cout, cin = w_oihw.shape[0], w_oihw.shape[1]
    if cin != C or H % 2 or W % 2 or bn_weight.numel() != cout:
        raise ValueError('wino_fused_conv_pool_sel: channel mismatch or odd map size')
    if upk is None:
        upk = wino_fused_pack(w_oihw)
    zsel = torch.empty(N, H // 2, W // 2, cout, device=x.device, dtype=torch.float32)
    stats = torch.empty(_lib.check(lib.ssdhip_wino_fused_stat_rows(N, H, W), 'wino_fused_stat_rows'), cout, 2, device=x.device,
                        dtype=torch.float32) if want_stats else None
    _lib.check(lib.ssdhip_wino_fused_conv_pool_sel(_lib.ptr(x), _lib.ptr(upk), _lib.ptr(bias) if bias is not None else None,
                                                   _lib.ptr(in_scale) if in_scale is not None else None,
                                                   _lib.ptr(in_shift) if in_shift is not None else None, _lib.ptr(_lib.f32c(bn_weight)), _lib.ptr(zsel),
                                                   _lib.ptr(stats) if stats is not None else None, _lib.ptr(wino_fused_geo(N, H, W, x.device)),
                                                   N, H, W, C, cout, max_blocks, _lib.stream()), 'wino_fused_conv_pool_sel')
    return zsel, stats


def bn_relu_select(zmax, zmin, scale, shift):
    """max-pool(relu(bn(z))) from the window maxima / minima of z (see wino_fused_conv_pool)."""
    _lib.require_gpu(zmax, zmin, scale, shift)
    y = torch.empty_like(zmax)
    C = zmax.shape[-1]
    _lib.check(_lib.load().ssdhip_bn_relu_select(_lib.ptr(zmax), _lib.ptr(zmin), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y), zmax.numel() // C, C,
                                                 _lib.stream()), 'bn_relu_select')
    return y


def conv_bf16_pack(w_oihw):
    """OIHW fp32 (Cout % 64 == 0, Cin % 32 == 0, 3, 3) -> the bf16 patch convolution's weight images (uint8 tensor)."""
    _lib.require_gpu(w_oihw)
    lib = _lib.load()
    w = _lib.f32c(w_oihw)
    cout, cin = w.shape[0], w.shape[1]
    if tuple(w.shape[2:]) != (3, 3):
        raise ValueError(f'conv_bf16_pack: 3x3 filters only, got {tuple(w.shape)}')
    wpk = torch.empty(_lib.check(lib.ssdhip_conv_bf16_weight_bytes(cout, cin), 'conv_bf16_weight_bytes'), device=w.device, dtype=torch.uint8)
    _lib.check(lib.ssdhip_conv_bf16_pack(_lib.ptr(w), _lib.ptr(wpk), cout, cin, _lib.stream()), 'conv_bf16_pack')
    return wpk


def haloed_bf16(N, H, W, C, device):
    """Zero-initialised allocation for an activated bf16 NHWC tensor with a one-pixel zero halo, (buffer, view (N, H+2, W+2, C))."""
    nbytes = _lib.check(_lib.load().ssdhip_conv_bf16_act_bytes(N, H, W, C), 'conv_bf16_act_bytes')
    buf = torch.zeros(nbytes // 2, device=device, dtype=torch.bfloat16)
    return buf, buf[: N * (H + 2) * (W + 2) * C].view(N, H + 2, W + 2, C)


def conv_bf16(a_buf, shape, w_oihw, bias=None, want_stats=False, wpk=None, z_bf16=False, max_blocks=0):
    """3x3/s1/p1 conv on the bf16 MFMA from a haloed activated bf16 input (haloed_bf16 buffer, shape = (N, H, W, Cin)):
    -> (z (N,H,W,Cout) fp32 or bf16, stats (MB,Cout,2) fp32 or None)."""
    _lib.require_gpu(a_buf, w_oihw)
    lib = _lib.load()
    N, H, W, C = shape
    cout, cin = w_oihw.shape[0], w_oihw.shape[1]
    if cin != C:
        raise ValueError(f'conv_bf16: input has {C} channels, weights expect {cin}')
    if a_buf.dtype != torch.bfloat16 or a_buf.numel() * 2 < _lib.check(lib.ssdhip_conv_bf16_act_bytes(N, H, W, C), 'conv_bf16_act_bytes'):
        raise ValueError('conv_bf16: input must be the bf16 buffer of haloed_bf16(N, H, W, C)')
    if wpk is None:
        wpk = conv_bf16_pack(w_oihw)
    z = torch.empty(N, H, W, cout, device=a_buf.device, dtype=torch.bfloat16 if z_bf16 else torch.float32)
    stats = torch.empty(_lib.check(lib.ssdhip_conv_bf16_stat_rows(N, H, W), 'conv_bf16_stat_rows'), cout, 2, device=a_buf.device,
                        dtype=torch.float32) if want_stats else None
    _lib.check(lib.ssdhip_conv_bf16(_lib.ptr(a_buf), _lib.ptr(wpk), _lib.ptr(bias) if bias is not None else None, _lib.ptr(z), int(z_bf16),
                                    _lib.ptr(stats) if stats is not None else None, N, H, W, C, cout, max_blocks, _lib.stream()), 'conv_bf16')
    return z, stats


def bn_relu_to_bf16(z, scale, shift, pool_pad=None):
    """relu(z * scale + shift) (then MaxPool2d(2, 2, pool_pad) unless pool_pad is None) of z (N,H,W,C) fp32 or bf16 -> the haloed
    activated bf16 input of conv_bf16: (buffer, view (N, Ho+2, Wo+2, C))."""
    _lib.require_gpu(z, scale, shift)
    if z.dtype not in (torch.float32, torch.bfloat16) or not z.is_contiguous():
        raise ValueError('bn_relu_to_bf16: z must be a contiguous fp32 or bf16 tensor')
    N, H, W, C = z.shape
    pool = pool_pad is not None
    pad = int(pool_pad or 0)
    Ho, Wo = ((H + 2 * pad - 2) // 2 + 1, (W + 2 * pad - 2) // 2 + 1) if pool else (H, W)
    buf, view = haloed_bf16(N, Ho, Wo, C, z.device)
    _lib.check(_lib.load().ssdhip_bn_relu_to_bf16(_lib.ptr(z), int(z.dtype == torch.bfloat16), _lib.ptr(_lib.f32c(scale)), _lib.ptr(_lib.f32c(shift)),
                                                  _lib.ptr(buf), N, H, W, C, int(pool), pad, _lib.stream()), 'bn_relu_to_bf16')
    return buf, view


def conv2d_dgrad(dy, w_oihw, in_hw, stride=1, pad=0, accumulate_into=None, tile=0, bf16=False, split_k=True):
    """dy (N,Ho,Wo,Cout_pad) -> dx (N,Hi,Wi,Cin): data gradient of conv2d(x, w, stride, pad)."""
    _lib.require_gpu(dy, w_oihw)
    N, ho, wo, cdy = dy.shape
    cout, cin, k, _ = w_oihw.shape
    packed, KP, inner = pack_weight(w_oihw, 1)
    if inner != cdy:
        raise ValueError(f'dy has {cdy} channels, expected {inner} (Cout padded to a multiple of 32)')
    hi, wi = in_hw
    dx = accumulate_into if accumulate_into is not None else torch.empty(N, hi, wi, cin, device=dy.device, dtype=torch.float32)
    p = _lib.ConvParams()
    p.x, p.w, p.y, p.y_batch_stride = dy.data_ptr(), packed.data_ptr(), dx.data_ptr(), hi * wi * cin
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = N, ho, wo, cdy, hi, wi, cin
    p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = k, k, stride, pad, KP, 1, int(accumulate_into is not None), tile
    p.compute = int(bool(bf16))
    ws = _splitk_ws(_lib.load(), p, split_k)
    _lib.check(_lib.load().ssdhip_conv2d_nhwc(ctypes.byref(p), _lib.stream()), 'conv2d_dgrad')
    return dx


def conv2d_wgrad(x, dy, w_shape, stride=1, pad=0, in_scale=None, in_shift=None, want_bias=True, winograd=False, bf16=False, wino_v=None):
    """x (N,Hi,Wi,Cin), dy (N,Ho,Wo,dy_row>=Cout) -> (dW OIHW, db or None).  winograd=True: F(2x2,3x3) path (3x3/s1/p1); winograd=4: F(4x4,3x3).
    wino_v (Winograd paths): the forward's input transform of x (planes padded to 128 rows), reused instead of recomputed.
    bf16 (direct path only): operands rounded to bf16 at staging, bf16 MFMA with fp32 accumulate, fp32 result."""
    _lib.require_gpu(x, dy)
    lib = _lib.load()
    cout, cin, k, _ = w_shape
    N, hi, wi, _ = x.shape
    _, ho, wo, row = dy.shape
    dw = torch.empty(w_shape, device=x.device, dtype=torch.float32)
    db = torch.empty(cout, device=x.device, dtype=torch.float32) if want_bias else None
    p = _lib.WgradParams()
    p.x, p.dy = x.data_ptr(), dy.data_ptr()
    p.in_scale = in_scale.data_ptr() if in_scale is not None else None
    p.in_shift = in_shift.data_ptr() if in_shift is not None else None
    p.dw_oihw, p.dbias = dw.data_ptr(), (db.data_ptr() if db is not None else None)
    p.dy_batch_stride = ho * wo * row
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = N, hi, wi, cin, ho, wo, cout
    p.R, p.S, p.stride, p.pad, p.dy_row = k, k, stride, pad, row
    if bf16 and winograd:
        raise ValueError('conv2d_wgrad: the bf16 weight gradient is the direct kernel (no Winograd form)')
    if wino_v is not None and not winograd:
        raise ValueError('conv2d_wgrad: wino_v only applies to the Winograd forms')
    p.wino_v = wino_v.data_ptr() if wino_v is not None else None
    p.compute = int(bf16)
    if winograd == 4 and not isinstance(winograd, bool):      # F(4x4,3x3)
        fn_ws, fn = lib.ssdhip_conv2d_wgrad_winograd4_ws_bytes, lib.ssdhip_conv2d_wgrad_winograd4
    else:
        fn_ws, fn = (lib.ssdhip_conv2d_wgrad_winograd_ws_bytes, lib.ssdhip_conv2d_wgrad_winograd) if winograd else \
            (lib.ssdhip_conv2d_wgrad_ws_bytes, lib.ssdhip_conv2d_wgrad)
    ws = torch.empty((_lib.check(fn_ws(ctypes.byref(p)), 'wgrad_ws_bytes') + 3) // 4, device=x.device, dtype=torch.float32)
    p.ws = ws.data_ptr()
    _lib.check(fn(ctypes.byref(p), _lib.stream()), 'conv2d_wgrad')
    return dw, db


def bn_finalize(stats, count, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """(MB,C,2) partials -> dict(scale, shift, mean, invstd); updates running stats in place like nn.BatchNorm2d."""
    _lib.require_gpu(stats, gamma, beta)
    MB, C, _ = stats.shape
    dev = stats.device
    out = {k: torch.empty(C, device=dev, dtype=torch.float32) for k in ('scale', 'shift', 'mean', 'invstd')}
    ws = torch.empty(64 * C * 2, device=dev, dtype=torch.float64)
    _lib.check(_lib.load().ssdhip_bn_finalize(_lib.ptr(stats), MB, C, int(count), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                                              _lib.ptr(running_var), momentum, eps, _lib.ptr(out['scale']), _lib.ptr(out['shift']),
                                              _lib.ptr(out['mean']), _lib.ptr(out['invstd']), _lib.ptr(ws), _lib.stream()), 'batch_norm')
    return out


def bn_relu_pool(z, scale, shift, pool=False, pool_pad=0):
    _lib.require_gpu(z, scale, shift)
    N, H, W, C = z.shape
    if pool:
        ho, wo = (H + 2 * pool_pad - 2) // 2 + 1, (W + 2 * pool_pad - 2) // 2 + 1
    else:
        ho, wo = H, W
    y = torch.empty(N, ho, wo, C, device=z.device, dtype=torch.float32)
    _lib.check(_lib.load().ssdhip_bn_relu_pool(_lib.ptr(z), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y), N, H, W, C, int(pool), pool_pad,
                                               _lib.stream()), 'bn_relu_pool')
    return y


def bn_relu_bwd(z, da, st):
    """Backward of relu(bn_train(z)): returns (dz, dgamma, dbeta).  st: dict from bn_finalize."""
    lib = _lib.load()
    C = z.shape[-1]
    M = z.numel() // C
    dev = z.device
    mb = max(1, min(512, (M + 127) // 128))
    partial = torch.empty(mb, C, 2, device=dev, dtype=torch.float32)
    coef = torch.empty(C, 2, device=dev, dtype=torch.float32)
    ws = torch.empty(64 * C * 2, device=dev, dtype=torch.float64)
    dgamma = torch.empty(C, device=dev, dtype=torch.float32)
    dbeta = torch.empty(C, device=dev, dtype=torch.float32)
    dz = torch.empty_like(z)
    s = _lib.stream()
    args = (_lib.ptr(z), _lib.ptr(da), _lib.ptr(st['scale']), _lib.ptr(st['shift']), _lib.ptr(st['mean']), _lib.ptr(st['invstd']))
    _lib.check(lib.ssdhip_bn_relu_bwd_reduce(*args, M, C, _lib.ptr(partial), mb, s), 'bn_relu_bwd_reduce')
    _lib.check(lib.ssdhip_bn_relu_bwd_finalize(_lib.ptr(partial), mb, C, M, _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(coef), _lib.ptr(ws), s),
               'bn_relu_bwd_finalize')
    _lib.check(lib.ssdhip_bn_relu_bwd_apply(*args, _lib.ptr(coef), M, C, _lib.ptr(dz), s), 'bn_relu_bwd_apply')
    return dz, dgamma, dbeta


def to_nhwc(x_nchw, pad_to=None):
    x = x_nchw.permute(0, 2, 3, 1).contiguous()
    if pad_to and x.shape[-1] < pad_to:
        x = torch.cat([x, x.new_zeros(*x.shape[:-1], pad_to - x.shape[-1])], dim=-1).contiguous()
    return x


def conv3x3_winograd(x, w_oihw, bias=None, in_scale=None, in_shift=None, want_stats=False, tile=0):
    """3x3 / stride 1 / pad 1 convolution through the Winograd F(2x2,3x3) passes.  x (N,H,W,Cin) NHWC, Cin % 32 == 0.
    Returns (z (N,H,W,Cout), stats (rows,Cout,2) or None)."""
    _lib.require_gpu(x, w_oihw)
    lib = _lib.load()
    N, H, W, C = x.shape
    cout, cin, k, _ = w_oihw.shape
    if k != 3 or cin != C or C % 32:
        raise ValueError('conv3x3_winograd: 3x3 kernels, Cin % 32 == 0')
    T, Tp = ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.ssdhip_winograd_tiles(N, H, W, ctypes.byref(T), ctypes.byref(Tp)), 'winograd_tiles')
    T, Tp = T.value, Tp.value
    rows_pad = _rup(cout, 128)
    dev = x.device
    U = torch.empty(16 * rows_pad * C, device=dev, dtype=torch.float32)
    V = torch.empty(16 * Tp * C, device=dev, dtype=torch.float32)
    Mo = torch.empty(16 * Tp * cout, device=dev, dtype=torch.float32)
    z = torch.empty(N, H, W, cout, device=dev, dtype=torch.float32)
    st = _lib.stream()
    _lib.check(lib.ssdhip_winograd_weight(_lib.ptr(_lib.f32c(w_oihw)), _lib.ptr(U), cout, cin, rows_pad, C, st), 'winograd_weight')
    _lib.check(lib.ssdhip_winograd_input(_lib.ptr(x), _lib.ptr(in_scale), _lib.ptr(in_shift), _lib.ptr(V), N, H, W, C, st), 'winograd_input')
    p = _lib.ConvParams()
    p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), U.data_ptr(), Mo.data_ptr(), Tp * cout
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 16, 1, Tp, C, 1, Tp, cout
    p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = 1, 1, 1, 0, C, 0, 0, tile
    p.rows_per_img, p.w_img_stride = Tp, rows_pad * C
    _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st), 'winograd gemm')
    stats = torch.empty((T + 7) // 8, cout, 2, device=dev, dtype=torch.float32) if want_stats else None
    _lib.check(lib.ssdhip_winograd_output(_lib.ptr(Mo), _lib.ptr(bias), _lib.ptr(z), _lib.ptr(stats), N, H, W, cout, st), 'winograd_output')
    return z, stats


# Winograd F(4x4,3x3) matrices for the points (0, 3/2, -3/2, 2/3, -2/3, inf) as exact rationals (csrc/winograd4.hip holds the same
# numbers as float / double constants; tests/test_host_cpu.py checks the convolution identity in exact arithmetic)
WINOGRAD4_BT = [['1', '0', '-97/36', '0', '1', '0'], ['0', '-2/3', '-4/9', '3/2', '1', '0'], ['0', '2/3', '-4/9', '-3/2', '1', '0'],
                ['0', '-3/2', '-9/4', '2/3', '1', '0'], ['0', '3/2', '-9/4', '-2/3', '1', '0'], ['0', '1', '0', '-97/36', '0', '1']]
WINOGRAD4_G = [['1', '0', '0'], ['8/65', '12/65', '18/65'], ['8/65', '-12/65', '18/65'], ['-81/130', '-27/65', '-18/65'],
               ['-81/130', '27/65', '-18/65'], ['0', '0', '1']]
WINOGRAD4_AT = [['1', '1', '1', '1', '1', '0'], ['0', '3/2', '-3/2', '2/3', '-2/3', '0'], ['0', '9/4', '9/4', '4/9', '4/9', '0'],
                ['0', '27/8', '-27/8', '8/27', '-8/27', '1']]


def conv3x3_winograd4(x, w_oihw, bias=None, in_scale=None, in_shift=None, want_stats=False, tile=0, return_v=False):
    """3x3 / stride 1 / pad 1 convolution through the Winograd F(4x4,3x3) passes (csrc/winograd4.hip).  x (N,H,W,Cin) NHWC,
    Cin % 32 == 0.  Returns (z (N,H,W,Cout), stats (rows,Cout,2) or None) (+ the input transform V with return_v: what the engine
    hands to the heads' weight gradient).  The padding rows T..Tp of every plane are filled with NaN here: the engine feeds the GEMM
    uninitialised memory there (torch.empty workspaces), which is only safe because no pass reads a product of a padding row -- a
    result that stays finite under this poison proves it."""
    _lib.require_gpu(x, w_oihw)
    lib = _lib.load()
    N, H, W, C = x.shape
    cout, cin, k, _ = w_oihw.shape
    if k != 3 or cin != C or C % 32:
        raise ValueError('conv3x3_winograd4: 3x3 kernels, Cin % 32 == 0')
    row_pad = 64 if tile and tile // 1000 == 64 else 128
    T, Tp = ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.ssdhip_winograd4_tiles(N, H, W, row_pad, ctypes.byref(T), ctypes.byref(Tp)), 'winograd4_tiles')
    T, Tp = T.value, Tp.value
    rows_pad = _rup(cout, 128)
    dev = x.device
    U = torch.empty(36 * rows_pad * C, device=dev, dtype=torch.float32)
    V = torch.empty(36 * Tp * C, device=dev, dtype=torch.float32)
    Mo = torch.empty(36 * Tp * cout, device=dev, dtype=torch.float32)
    z = torch.empty(N, H, W, cout, device=dev, dtype=torch.float32)
    st = _lib.stream()
    _lib.check(lib.ssdhip_winograd4_weight(_lib.ptr(_lib.f32c(w_oihw)), _lib.ptr(U), cout, cin, rows_pad, C, st), 'winograd4_weight')
    if Tp > T:
        V.view(36, Tp, C)[:, T:].fill_(float('nan'))     # padding rows of every plane feed the GEMM; their products must never be read back
    _lib.check(lib.ssdhip_winograd4_input(_lib.ptr(x), _lib.ptr(in_scale), _lib.ptr(in_shift), _lib.ptr(V), N, H, W, C, Tp, st), 'winograd4_input')
    p = _lib.ConvParams()
    p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), U.data_ptr(), Mo.data_ptr(), Tp * cout
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 36, 1, Tp, C, 1, Tp, cout
    p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = 1, 1, 1, 0, C, 0, 0, tile or 128128
    p.rows_per_img, p.w_img_stride = Tp, rows_pad * C
    _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st), 'winograd4 gemm')
    rows = _lib.check(lib.ssdhip_winograd4_stat_rows(N, H, W, cout), 'winograd4_stat_rows')
    stats = torch.empty(rows, cout, 2, device=dev, dtype=torch.float32) if want_stats else None
    got = _lib.check(lib.ssdhip_winograd4_output(_lib.ptr(Mo), _lib.ptr(bias), _lib.ptr(z), _lib.ptr(stats), 0, N, H, W, cout, Tp, st), 'winograd4_output')
    assert got == rows
    return (z, stats, V) if return_v else (z, stats)


def wino4f_pack(w_oihw):
    """OIHW 3x3 weights -> the per-K-step LDS images of the two-pass F(4x4) form (csrc/wino4_fused.hip): U = G g G^T in fp64 (the
    three-pass form's weight transform), then the blocked layout."""
    _lib.require_gpu(w_oihw)
    lib = _lib.load()
    cout, cin, k, _ = w_oihw.shape
    if k != 3 or cin % 32 or cout % 64:
        raise ValueError('wino4f_pack: 3x3 kernels, Cin % 32 == 0, Cout % 64 == 0')
    rows_pad = _rup(cout, 128)
    U = torch.empty(36 * rows_pad * cin, device=w_oihw.device, dtype=torch.float32)
    upk = torch.empty(36 * cout * cin, device=w_oihw.device, dtype=torch.float32)
    st = _lib.stream()
    _lib.check(lib.ssdhip_winograd4_weight(_lib.ptr(_lib.f32c(w_oihw)), _lib.ptr(U), cout, cin, rows_pad, cin, st), 'winograd4_weight')
    _lib.check(lib.ssdhip_wino4f_pack(_lib.ptr(U), _lib.ptr(upk), cout, cin, rows_pad, st), 'wino4f_pack')
    return upk


def conv3x3_wino4f(x, w_oihw, bias=None, in_scale=None, in_shift=None, want_stats=False, upk=None, pool=False, vb=None):
    """3x3 / stride 1 / pad 1 convolution through the two-pass Winograd F(4x4,3x3) form (csrc/wino4_fused.hip).  x (N,H,W,Cin) NHWC,
    Cin % 32 == 0, Cout % 64 == 0.  Returns (z (N,H,W,Cout), stats (rows,Cout,2) or None); with pool=True (even H, W) the per-window
    maxima and minima (zmax, zmin, stats) instead of z."""
    _lib.require_gpu(x, w_oihw)
    lib = _lib.load()
    N, H, W, C = x.shape
    cout, cin, k, _ = w_oihw.shape
    if k != 3 or cin != C or C % 32 or cout % 64:
        raise ValueError('conv3x3_wino4f: 3x3 kernels, Cin % 32 == 0, Cout % 64 == 0')
    dev = x.device
    if upk is None:
        upk = wino4f_pack(w_oihw)
    need = _lib.check(lib.ssdhip_wino4f_v_floats(N, H, W, C), 'wino4f_v_floats')
    if vb is None:
        vb = torch.empty(need, device=dev, dtype=torch.float32)
    assert vb.numel() >= need
    st = _lib.stream()
    _lib.check(lib.ssdhip_wino4f_input(_lib.ptr(x), _lib.ptr(in_scale), _lib.ptr(in_shift), _lib.ptr(vb), N, H, W, C, st), 'wino4f_input')
    rows = _lib.check(lib.ssdhip_wino4f_tile_blocks(N, H, W), 'wino4f_tile_blocks')
    stats = torch.empty(rows, cout, 2, device=dev, dtype=torch.float32) if want_stats else None
    if pool:
        zmax = torch.empty(N, H // 2, W // 2, cout, device=dev, dtype=torch.float32)
        zmin = torch.empty_like(zmax)
        got = _lib.check(lib.ssdhip_wino4f_gemm_pool(_lib.ptr(vb), _lib.ptr(upk), _lib.ptr(bias), _lib.ptr(zmax), _lib.ptr(zmin), _lib.ptr(stats),
                                                      N, H, W, C, cout, st), 'wino4f_gemm_pool')
        assert got == rows
        return zmax, zmin, stats
    z = torch.empty(N, H, W, cout, device=dev, dtype=torch.float32)
    got = _lib.check(lib.ssdhip_wino4f_gemm(_lib.ptr(vb), _lib.ptr(upk), _lib.ptr(bias), _lib.ptr(z), _lib.ptr(stats), N, H, W, C, cout, st), 'wino4f_gemm')
    assert got == rows
    return z, stats


def conv3x3_dgrad_winograd4(dy, w_oihw, accumulate_into=None):
    """Data gradient of a 3x3 / stride 1 / pad 1 convolution through the F(4x4,3x3) passes: the adjoint convolution of dy.
    dy (N,H,W,KP) with KP = Cout padded to a multiple of 32 (zero channels beyond Cout) -> dx (N,H,W,Cin)."""
    _lib.require_gpu(dy, w_oihw)
    lib = _lib.load()
    N, H, W, KP = dy.shape
    cout, cin, k, _ = w_oihw.shape
    if k != 3 or KP < cout or KP % 32 or cin % 4 or 256 % (cin // 4):
        raise ValueError('conv3x3_dgrad_winograd4: 3x3 kernels, dy rows padded to 32 channels, Cin = 4 * a divisor of 256')
    T, Tp = ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.ssdhip_winograd4_tiles(N, H, W, 128, ctypes.byref(T), ctypes.byref(Tp)), 'winograd4_tiles')
    T, Tp = T.value, Tp.value
    rows = _rup(cin, 128)
    dev = dy.device
    U = torch.empty(36 * rows * KP, device=dev, dtype=torch.float32)
    V = torch.full((36 * Tp * KP,), float('nan'), device=dev, dtype=torch.float32)     # rows t < T are overwritten, the padding rows stay poisoned
    Mo = torch.empty(36 * Tp * cin, device=dev, dtype=torch.float32)
    dx = accumulate_into if accumulate_into is not None else torch.empty(N, H, W, cin, device=dev, dtype=torch.float32)
    st = _lib.stream()
    _lib.check(lib.ssdhip_winograd4_weight_adjoint(_lib.ptr(_lib.f32c(w_oihw)), _lib.ptr(U), cout, cin, rows, KP, st), 'winograd4_weight_adjoint')
    _lib.check(lib.ssdhip_winograd4_input(_lib.ptr(dy), None, None, _lib.ptr(V), N, H, W, KP, Tp, st), 'winograd4_input')
    p = _lib.ConvParams()
    p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), U.data_ptr(), Mo.data_ptr(), Tp * cin
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 36, 1, Tp, KP, 1, Tp, cin
    p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = 1, 1, 1, 0, KP, 0, 0, 128128
    p.rows_per_img, p.w_img_stride = Tp, rows * KP
    _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st), 'winograd4 gemm (dgrad)')
    _lib.check(lib.ssdhip_winograd4_output(_lib.ptr(Mo), None, _lib.ptr(dx), None, int(accumulate_into is not None), N, H, W, cin, Tp, st),
               'winograd4_output (dgrad)')
    return dx
