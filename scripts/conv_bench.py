"""Per-layer timing of the HIP conv kernels at the SSD300 batch-32 shapes (forward; --bwd adds dgrad/wgrad of the tail)."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import ops
ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, default=32); ap.add_argument('--tiles', default='0'); ap.add_argument('--layers', default='')
ap.add_argument('--iters', type=int, default=5)
args = ap.parse_args()
DEV = 'cuda:0'
# name, H, Cin, Cout, k, stride, pad, xform
L = [('conv_1_1', 300, 3, 64, 3, 1, 1, 0), ('conv_1_2', 300, 64, 64, 3, 1, 1, 1), ('conv_2_1', 150, 64, 128, 3, 1, 1, 0), ('conv_2_2', 150, 128, 128, 3, 1, 1, 1),
     ('conv_3_1', 75, 128, 256, 3, 1, 1, 0), ('conv_3_2', 75, 256, 256, 3, 1, 1, 1), ('conv_4_1', 38, 256, 512, 3, 1, 1, 0), ('conv_4_2', 38, 512, 512, 3, 1, 1, 1),
     ('conv_5_1', 19, 512, 512, 3, 1, 1, 0), ('conv_5_2', 19, 512, 512, 3, 1, 1, 1), ('conv_6_1', 19, 512, 1024, 3, 1, 1, 1), ('conv_7_1', 19, 1024, 1024, 1, 1, 0, 1),
     ('conv_8_1', 19, 1024, 256, 1, 1, 0, 1), ('conv_8_2', 19, 256, 512, 3, 2, 1, 1), ('det_4_3', 38, 512, 100, 3, 1, 1, 1), ('det_7_1', 19, 1024, 150, 3, 1, 1, 1)]
sel = set(args.layers.split(',')) if args.layers else None
tiles = [int(t) for t in args.tiles.split(',')]
for name, H, cin, cout, k, stride, pad, xf in L:
    if sel and name not in sel: continue
    N = args.n
    x = torch.randn(N, H, H, 4 if cin == 3 else cin, device=DEV)
    w = torch.randn(cout, cin, k, k, device=DEV) * 0.05
    b = torch.randn(cout, device=DEV)
    sc = torch.rand(cin, device=DEV) + 0.5 if xf else None
    sh = torch.randn(cin, device=DEV) if xf else None
    ho = (H + 2 * pad - k) // stride + 1
    fl = 2.0 * N * ho * ho * cout * cin * k * k
    for tile in tiles:
        try:
            ops.conv2d_nhwc(x, w, b, stride=stride, pad=pad, in_scale=sc, in_shift=sh, want_stats=True, tile=tile)
        except Exception as e:
            print(name, tile, 'ERR', e); continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(args.iters):
            e0.record(); ops.conv2d_nhwc(x, w, b, stride=stride, pad=pad, in_scale=sc, in_shift=sh, want_stats=True, tile=tile); e1.record()
            torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
        print(f'{name:9s} tile {tile:6d} M={N*ho*ho:8d} N={cout:5d} K={cin*k*k:5d}  {best:8.3f} ms  {fl/best/1e9:7.1f} TF/s', flush=True)
