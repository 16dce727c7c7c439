"""Weight gradient of the 3x3 / s1 / p1 layers of the trainable tail at batch 32: direct, Winograd F(2x2) and F(4x4) forms (HIP events)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import _lib
lib = _lib.load()
dev = 'cuda:0'
REPS = 20
SHAPES = [('conv_6_1', 32, 19, 19, 512, 1024, 1024), ('det_4_3', 32, 38, 38, 512, 100, 128), ('det_7_1', 32, 19, 19, 1024, 150, 160), ('det_8_2', 32, 10, 10, 512, 150, 160)]


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


for name, N, H, W, cin, cout, row in SHAPES:
    x = torch.randn(N, H, W, cin, device=dev)
    dy = torch.randn(N, H, W, row, device=dev)
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    dw = torch.empty(cout, cin, 3, 3, device=dev)
    p = _lib.WgradParams()
    p.x, p.dy, p.in_scale, p.in_shift, p.dw_oihw, p.dbias = x.data_ptr(), dy.data_ptr(), sc.data_ptr(), sh.data_ptr(), dw.data_ptr(), None
    p.dy_batch_stride = H * W * row
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = N, H, W, cin, H, W, cout
    p.R, p.S, p.stride, p.pad, p.dy_row = 3, 3, 1, 1, row
    out = []
    for tag, fws, f in (('direct', lib.ssdhip_conv2d_wgrad_ws_bytes, lib.ssdhip_conv2d_wgrad), ('F(2x2)', lib.ssdhip_conv2d_wgrad_winograd_ws_bytes, lib.ssdhip_conv2d_wgrad_winograd),
                        ('F(4x4)', lib.ssdhip_conv2d_wgrad_winograd4_ws_bytes, lib.ssdhip_conv2d_wgrad_winograd4)):
        ws = torch.empty((_lib.check(fws(ctypes.byref(p)), 'ws') + 3) // 4, device=dev)
        p.ws = ws.data_ptr()
        st = _lib.stream()
        out.append(f'{tag} {timed(lambda: _lib.check(f(ctypes.byref(p), st), tag)):7.1f} us')
    print(name, ' | '.join(out), flush=True)
