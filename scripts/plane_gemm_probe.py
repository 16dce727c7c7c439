"""Plane GEMM (gemm_planes.hip) at the Winograd-domain shapes of SSD300 batch 32: tile / tiles-per-block sweep and timing-only
ablations (tile code 7 = no stores, 6 = no DMA after the first step, 5 = no barrier; results of 5/6/7 are wrong by design)."""
import os; os.environ.setdefault('SSDHIP_PROBES', '1')   # probe build of the library (make -C object_detection_torch2_amd/csrc probes)
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import _lib
DEV = 'cuda:0'
lib = _lib.load()
SHAPES = [('conv_2_2', 180224, 128, 128), ('conv_3_1', 45056, 128, 256), ('conv_4_2', 11648, 512, 512), ('conv_3_2', 46208, 256, 256), ('conv_4_1', 11648, 256, 512), ('conv_5_x', 3200, 512, 512), ('conv_6_1', 3200, 512, 1024)]
codes = [int(c) for c in sys.argv[1:]] or [128128, 7128128, 6128128, 5128128, 40128128, 41128128, 43128128, 45128128, 47128128]
def run(p, st):
    return lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st)
for name, Tp, cin, cout in SHAPES:
    rows_pad = (cout + 127) // 128 * 128
    U = torch.randn(16 * rows_pad * cin, device=DEV) * 0.05; V = torch.randn(16 * Tp * cin, device=DEV); Mo = torch.empty(16 * Tp * cout, device=DEV)
    st = _lib.stream(); out = []
    ps = {}
    for tile in codes:
        p = _lib.ConvParams(); p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), U.data_ptr(), Mo.data_ptr(), Tp * cout
        p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 16, 1, Tp, cin, 1, Tp, cout
        p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = 1, 1, 1, 0, cin, 0, 0, tile
        p.rows_per_img, p.w_img_stride = Tp, rows_pad * cin
        if run(p, st) == 0:
            ps[tile] = p
    torch.cuda.synchronize()
    best = {t: 1e9 for t in ps}
    for _ in range(6):                      # interleaved rounds: no first-measured bias
        for t in ps:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(ps[t], st); e1.record(); torch.cuda.synchronize(); best[t] = min(best[t], e0.elapsed_time(e1))
    fl = 2.0 * 16 * Tp * cin * cout
    print(f'{name:9s} ' + ' '.join(f'{t}:{best[t]*1e3:.0f}us/{fl/best[t]/1e9:.0f}TF' for t in ps), flush=True)
