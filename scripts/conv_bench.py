"""Per-layer timing of the HIP conv kernels at the SSD300 batch-32 shapes (forward; --bwd adds dgrad/wgrad of the tail)."""
import os; os.environ.setdefault('SSDHIP_PROBES', '1')   # probe build of the library (make -C object_detection_torch2_amd/csrc probes)
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
    ok_tiles = []
    for tile in tiles:
        try:
            ops.conv2d_nhwc(x, w, b, stride=stride, pad=pad, in_scale=sc, in_shift=sh, want_stats=True, tile=tile); ok_tiles.append(tile)
        except Exception as e:
            print(name, tile, 'ERR', e)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = {t: [] for t in ok_tiles}
    for _ in range(args.iters):            # interleaved rounds: every variant sees the same clock / cache state
        for tile in ok_tiles:
            e0.record(); ops.conv2d_nhwc(x, w, b, stride=stride, pad=pad, in_scale=sc, in_shift=sh, want_stats=True, tile=tile); e1.record()
            torch.cuda.synchronize(); times[tile].append(e0.elapsed_time(e1))
    for tile in ok_tiles:
        ts = sorted(times[tile][1:] or times[tile]); best, med = ts[0], ts[len(ts) // 2]
        print(f'{name:9s} tile {tile:6d} M={N*ho*ho:8d} N={cout:5d} K={cin*k*k:5d}  min {best:7.3f} ms {fl/best/1e9:6.1f} TF/s | median {med:7.3f} ms {fl/med/1e9:6.1f} TF/s', flush=True)

# ---- backward of the trainable tail: wgrad (+bias) and dgrad at batch n ----
if os.environ.get('BWD'):
    T = [('conv_6_1', 19, 512, 1024, 3, 1, 1, False), ('conv_7_1', 19, 1024, 1024, 1, 1, 0, True), ('conv_8_1', 19, 1024, 256, 1, 1, 0, True),
         ('conv_8_2', 19, 256, 512, 3, 2, 1, True), ('conv_9_1', 10, 512, 128, 1, 1, 0, True), ('conv_9_2', 10, 128, 256, 3, 2, 1, True),
         ('det_4_3', 38, 512, 100, 3, 1, 1, False), ('det_7_1', 19, 1024, 150, 3, 1, 1, True), ('det_8_2', 10, 512, 150, 3, 1, 1, True)]
    def timeit(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(args.iters):
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
        return best
    for name, H, cin, cout, k, stride, pad, dg in T:
        N = args.n
        ho = (H + 2 * pad - k) // stride + 1
        x = torch.randn(N, H, H, cin, device=DEV); sc = torch.rand(cin, device=DEV) + 0.5; sh = torch.randn(cin, device=DEV)
        cpad = (cout + 31) // 32 * 32
        dy = torch.randn(N, ho, ho, cpad, device=DEV)
        w = torch.randn(cout, cin, k, k, device=DEV) * 0.05
        fl = 2.0 * N * ho * ho * cout * cin * k * k
        t = timeit(lambda: ops.conv2d_wgrad(x, dy, (cout, cin, k, k), stride=stride, pad=pad, in_scale=sc, in_shift=sh))
        line = f'{name:9s} wgrad {t:8.3f} ms {fl/t/1e9:7.1f} TF/s'
        if k == 3 and stride == 1 and pad == 1:
            tw = timeit(lambda: ops.conv2d_wgrad(x, dy, (cout, cin, k, k), stride=stride, pad=pad, in_scale=sc, in_shift=sh, winograd=True))
            line += f' (winograd {tw:.3f} ms)'
        if dg:
            t = timeit(lambda: ops.conv2d_dgrad(dy, w, (H, H), stride=stride, pad=pad))
            line += f' | dgrad {t:8.3f} ms {fl/t/1e9:7.1f} TF/s'
        print(line, flush=True)
