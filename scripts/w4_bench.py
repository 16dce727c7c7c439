"""Per-pass timing of the Winograd F(4x4,3x3) path (csrc/winograd4.hip + the plane GEMM) on the backbone shapes at batch 32:
input transform, 36-plane GEMM, output transform; HIP events over REPS launches after a warm-up.
    python scripts/w4_bench.py            (SSDHIP_PROBES=1 SSDHIP_W4_VEC=4 for the float4 input transform of the probe build)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import _lib

lib = _lib.load()
dev = 'cuda:0'
REPS = int(os.environ.get('REPS', '20'))
SHAPES = [('conv_3_2', 32, 75, 75, 256, 256), ('conv_4_1', 32, 38, 38, 256, 512), ('conv_4_2', 32, 38, 38, 512, 512), ('conv_5_2', 32, 19, 19, 512, 512),
          ('conv_6_1', 32, 19, 19, 512, 1024)]


def timed(fn):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


for name, N, H, W, cin, cout in SHAPES:
    T = N * ((H + 3) // 4) * ((W + 3) // 4)
    tp128 = (T + 127) // 128 * 128
    tile = 128128 if (36 * tp128 // 128) * ((cout + 127) // 128) > 2000 else 64128
    if os.environ.get('TILE'):
        tile = int(os.environ['TILE'])
    pad = tile % 1000000 // 1000          # (probe builds: tile codes >= 1,000,000 carry an A/B selector in front)
    tp = (T + pad - 1) // pad * pad
    rows_pad = (cout + 127) // 128 * 128
    x = torch.randn(N, H, W, cin, device=dev)
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    U = torch.empty(36 * rows_pad * cin, device=dev)
    V = torch.zeros(36 * tp * cin, device=dev)
    M = torch.empty(36 * tp * cout, device=dev)
    z = torch.empty(N, H, W, cout, device=dev)
    rows = lib.ssdhip_winograd4_stat_rows(N, H, W, cout)
    stats = torch.empty(rows * cout * 2, device=dev)
    st = _lib.stream()
    _lib.check(lib.ssdhip_winograd4_weight(_lib.ptr(w), _lib.ptr(U), cout, cin, rows_pad, cin, st), 'w')
    p = _lib.ConvParams()
    p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), U.data_ptr(), M.data_ptr(), tp * cout
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 36, 1, tp, cin, 1, tp, cout
    p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = 1, 1, 1, 0, cin, 0, 0, tile
    p.rows_per_img, p.w_img_stride = tp, rows_pad * cin
    t_in = timed(lambda: _lib.check(lib.ssdhip_winograd4_input(_lib.ptr(x), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(V), N, H, W, cin, tp, st), 'in'))
    t_g = timed(lambda: _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st), 'gemm'))
    t_out = timed(lambda: _lib.check(lib.ssdhip_winograd4_output(_lib.ptr(M), _lib.ptr(b), _lib.ptr(z), _lib.ptr(stats), 0, N, H, W, cout, tp, st), 'out'))
    gf = 2.0 * 36 * tp * cin * cout / 1e9
    mb_in = (x.numel() + V.numel()) * 4 / 1e6
    mb_out = (M.numel() + z.numel()) * 4 / 1e6
    print(f'{name}: T={T} tp={tp} tile={tile}  input {t_in:7.1f} us ({mb_in / t_in:5.2f} TB/s)  gemm {t_g:7.1f} us ({gf / t_g * 1e3:6.1f} TF)  '
          f'output {t_out:7.1f} us ({mb_out / t_out:5.2f} TB/s)  total {t_in + t_g + t_out:7.1f} us = {2.0 * N * H * W * cin * cout * 9 / 1e6 / (t_in + t_g + t_out):6.1f} TF direct-equivalent', flush=True)
