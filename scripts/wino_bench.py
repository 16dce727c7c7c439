"""Winograd F(2x2,3x3) passes vs the direct implicit-GEMM kernel at the SSD300 batch-32 shapes."""
import os; os.environ.setdefault('SSDHIP_PROBES', '1')   # probe build of the library (make -C object_detection_torch2_amd/csrc probes)
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import ops, _lib
DEV = 'cuda:0'
L = [('conv_3_1', 75, 128, 256), ('conv_3_2', 75, 256, 256), ('conv_4_1', 38, 256, 512), ('conv_4_2', 38, 512, 512), ('conv_5_2', 19, 512, 512), ('conv_6_1', 19, 512, 1024),
     ('conv_2_2', 150, 128, 128)]
lib = _lib.load()
def T(fn, it=5):
    fn(); torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); b = 1e9
    for _ in range(it):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); b = min(b, e0.elapsed_time(e1))
    return b
for name, H, cin, cout in L:
    N = 32
    x = torch.randn(N, H, H, cin, device=DEV); w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05; b = torch.randn(cout, device=DEV)
    sc = torch.rand(cin, device=DEV) + 0.5; sh = torch.randn(cin, device=DEV)
    if cout % 4 or 256 % (cout // 4):
        continue
    td = T(lambda: ops.conv2d_nhwc(x, w, b, pad=1, in_scale=sc, in_shift=sh, want_stats=True))
    t_, tp_ = ctypes.c_int(), ctypes.c_int(); lib.ssdhip_winograd_tiles(N, H, H, ctypes.byref(t_), ctypes.byref(tp_)); Tn, Tp = t_.value, tp_.value
    rows_pad = (cout + 127) // 128 * 128
    U = torch.empty(16 * rows_pad * cin, device=DEV); V = torch.empty(16 * Tp * cin, device=DEV); Mo = torch.empty(16 * Tp * cout, device=DEV)
    z = torch.empty(N, H, H, cout, device=DEV); stats = torch.empty((Tn + 7) // 8, cout, 2, device=DEV); st = _lib.stream()
    lib.ssdhip_winograd_weight(_lib.ptr(w), _lib.ptr(U), cout, cin, rows_pad, cin, st)
    ti = T(lambda: lib.ssdhip_winograd_input(_lib.ptr(x), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(V), N, H, H, cin, st))
    res = {}
    for tile in (128128, 64128, 128064, 9128128, 9064128):
        p = _lib.ConvParams(); p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), U.data_ptr(), Mo.data_ptr(), Tp * cout
        p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 16, 1, Tp, cin, 1, Tp, cout
        p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = 1, 1, 1, 0, cin, 0, 0, tile
        p.rows_per_img, p.w_img_stride = Tp, rows_pad * cin
        res[tile] = T(lambda: lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st))
    to = T(lambda: lib.ssdhip_winograd_output(_lib.ptr(Mo), _lib.ptr(b), _lib.ptr(z), _lib.ptr(stats), N, H, H, cout, st))
    tg = min(res.values())
    print(f'{name:9s} direct {td:.3f} ms | winograd in {ti:.3f} + gemm {tg:.3f} ({min(res, key=res.get)}; ' + ' '.join(f'{k}:{v:.3f}' for k, v in res.items()) + f') + out {to:.3f} = {ti+tg+to:.3f} ms  (x{td/(ti+tg+to):.2f})', flush=True)
