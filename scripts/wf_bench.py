"""Fused Winograd kernel (wino_fused.hip) vs the paths the engine used before, at the SSD300 batch-32 backbone shapes."""
import os; os.environ.setdefault('SSDHIP_PROBES', '1')   # probe build of the library (make -C object_detection_torch2_amd/csrc probes)
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import ops, _lib
DEV = 'cuda:0'
N = int(os.environ.get('N', '32'))
L = [('conv_1_2', 300, 64, 64), ('conv_2_1', 150, 64, 128), ('conv_2_2', 150, 128, 128), ('conv_3_1', 75, 128, 256), ('conv_3_2', 75, 256, 256),
     ('conv_4_1', 38, 256, 512), ('conv_4_2', 38, 512, 512), ('conv_5_1', 19, 512, 512), ('conv_6_1', 19, 512, 1024)]
only = os.environ.get('LAYERS')
torch.manual_seed(0)
def timeit(fn, reps=6):
    for _ in range(4):       # the first launches of a process run slower (clock ramp): an A/B measured first loses by 10 %
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[0], ts[len(ts) // 2]
for name, H, cin, cout in L:
    if only and name not in only.split(','):
        continue
    x = torch.randn(N, H, H, cin, device=DEV)
    w = torch.randn(cout, cin, 3, 3, device=DEV) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, device=DEV)
    sc, sh = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.1
    upk = ops.wino_fused_pack(w)
    direct = 2.0 * N * H * H * cin * cout * 9
    tiles = N * ((H + 1) // 2) ** 2
    wino = 2.0 * 16 * tiles * cin * cout
    best, med = timeit(lambda: ops.wino_fused_conv(x, w, b, in_scale=sc, in_shift=sh, want_stats=True, upk=upk))
    for sk in [int(v) for v in os.environ.get('SKEWS', '').split(',') if v]:
        os.environ['SSDHIP_WF_SKEW'] = str(sk)
        ba, ma = timeit(lambda: ops.wino_fused_conv(x, w, b, in_scale=sc, in_shift=sh, want_stats=True, upk=upk))
        del os.environ['SSDHIP_WF_SKEW']
        print(f'    skew {sk}: min {ba:7.3f} med {ma:7.3f} ms ({wino / ma / 1e9:6.1f} TF executed)', flush=True)
    for bd in [int(v) for v in os.environ.get('BANDS', '').split(',') if v]:
        os.environ['SSDHIP_WF_BAND'] = str(bd)
        ba, ma = timeit(lambda: ops.wino_fused_conv(x, w, b, in_scale=sc, in_shift=sh, want_stats=True, upk=upk))
        del os.environ['SSDHIP_WF_BAND']
        print(f'    band {bd}: min {ba:7.3f} med {ma:7.3f} ms ({wino / ma / 1e9:6.1f} TF executed)', flush=True)
    for abl in [int(v) for v in os.environ.get('ABLS8', '').split(',') if v]:
        os.environ['SSDHIP_WF8_ABL'] = str(abl)
        ba, ma = timeit(lambda: ops.wino_fused_conv(x, w, b, in_scale=sc, in_shift=sh, want_stats=True, upk=upk))
        del os.environ['SSDHIP_WF8_ABL']
        print(f'    8-wave ablation {abl}: med {ma:7.3f} ms ({wino / ma / 1e9:6.1f} TF executed)', flush=True)
    for abl in [int(v) for v in os.environ.get('ABLS', '').split(',') if v]:
        os.environ['SSDHIP_WF_ABL'] = str(abl)
        ba, ma = timeit(lambda: ops.wino_fused_conv(x, w, b, in_scale=sc, in_shift=sh, want_stats=True, upk=upk))
        os.environ['SSDHIP_WF_ABL'] = '0'
        print(f'    ablation {abl}: med {ma:7.3f} ms ({wino / ma / 1e9:6.1f} TF executed)', flush=True)
    line = f'{name:9s} H={H:3d} {cin:4d}->{cout:4d}  fused min {best:7.3f} ms med {med:7.3f}  ({wino / med / 1e9:6.1f} TF executed, {direct / med / 1e9:6.1f} TF direct-equivalent)'
    if cin == 64:
        b2, m2 = timeit(lambda: ops.conv_c64(x, w, b, in_scale=sc, in_shift=sh, want_stats=True))
        line += f' | conv_c64 med {m2:7.3f}'
    elif cin % 32 == 0 and 256 % (cout // 4) == 0:
        b2, m2 = timeit(lambda: ops.conv3x3_winograd(x, w, b, in_scale=sc, in_shift=sh, want_stats=True))
        line += f' | 3-pass winograd med {m2:7.3f} (incl. weight transform)'
    print(line, flush=True)
