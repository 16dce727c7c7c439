"""conv_1_2 / conv_2_1 shapes at batch 32: dedicated Cin = 64 kernel vs the staged implicit-GEMM kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import ops
DEV = 'cuda:0'
for name, H, cout, xf in (('conv_1_2', 300, 64, True), ('conv_2_1', 150, 128, False)):
    x = torch.randn(32, H, H, 64, device=DEV); w = torch.randn(cout, 64, 3, 3, device=DEV) * 0.05; b = torch.zeros(cout, device=DEV)
    sc, sh = (torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)) if xf else (None, None)
    fl = 2.0 * 32 * H * H * cout * 576
    for tag, fn in (('c64', lambda: ops.conv_c64(x, w, b, sc, sh, want_stats=True)), ('igemm', lambda: ops.conv2d_nhwc(x, w, b, pad=1, in_scale=sc, in_shift=sh, want_stats=True))):
        fn(); torch.cuda.synchronize(); best = 1e9
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
        print(f'{name} {tag}: {best*1e3:.0f} us {fl/best/1e9:.1f} TF (incl. weight pack for c64)', flush=True)
