"""Plane GEMM (36 planes, 128x128 tiles) at fixed M and N over K: fits t(tile) = c + KT * s, i.e. the per-tile fixed cost against the
per-K-step cost.  python scripts/plane_k_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import _lib
lib = _lib.load(); dev = 'cuda:0'
REPS = 20
def timed(fn):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3
for tp, cout in ((3200, 512), (11648, 256), (896, 1024)):
    for cin in (128, 256, 384, 512, 768, 1024):
        rows_pad = (cout + 127) // 128 * 128
        U = torch.randn(36 * rows_pad * cin, device=dev) * 0.02
        V = torch.randn(36 * tp * cin, device=dev)
        M = torch.empty(36 * tp * cout, device=dev)
        p = _lib.ConvParams()
        p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), U.data_ptr(), M.data_ptr(), tp * cout
        p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 36, 1, tp, cin, 1, tp, cout
        p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = 1, 1, 1, 0, cin, 0, 0, int(os.environ.get('TILE', '128128'))
        p.rows_per_img, p.w_img_stride = tp, rows_pad * cin
        st = _lib.stream()
        t = timed(lambda: _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st), 'gemm'))
        tiles = 36 * tp // 128 * ((cout + 127) // 128)
        per_tile = t * 512 / tiles        # us of one resident block slot per tile (2 blocks per CU x 256 CUs)
        print(f'tp {tp} cout {cout} K {cin:4d}: {t:7.1f} us  {2.0 * 36 * tp * cin * cout / t / 1e6:6.1f} TF  tiles {tiles}  slot-time per tile {per_tile:6.2f} us  per K-step {per_tile / (cin // 32):5.2f}', flush=True)
