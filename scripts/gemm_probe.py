"""How fast is the implicit-GEMM kernel as a plain 1x1 GEMM at Winograd-domain shapes (short K loops)?"""
import os; os.environ.setdefault('SSDHIP_PROBES', '1')   # probe build of the library (make -C object_detection_torch2_amd/csrc probes)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import ops
DEV = 'cuda:0'
for name, rows, cin, cout in (('conv_4_2 wino', 16 * 11648, 512, 512), ('conv_4_1 wino', 16 * 11648, 256, 512), ('conv_5 wino', 16 * 3200, 512, 512),
                               ('conv_6_1 wino', 16 * 3200, 512, 1024), ('conv_3_2 wino', 16 * 46208, 256, 256)):
    x = torch.randn(1, rows // 64, 64, cin, device=DEV)
    w = torch.randn(cout, cin, 1, 1, device=DEV) * 0.05
    for tile in (0, 128128, 64128):
        ops.conv2d_nhwc(x, w, None, tile=tile); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(); ops.conv2d_nhwc(x, w, None, tile=tile); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
        fl = 2.0 * rows * cin * cout
        print(f'{name:14s} tile {tile:6d} rows {rows} K {cin} N {cout}: {best:.3f} ms {fl/best/1e9:.1f} TF/s', flush=True)
