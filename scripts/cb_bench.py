"""bf16 patch convolution (conv_bf16.hip) at the SSD300 batch-32 backbone shapes vs the bf16 template of the implicit-GEMM kernel."""
import os; os.environ.setdefault('SSDHIP_PROBES', '1')
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import ops
DEV = 'cuda:0'
N = int(os.environ.get('N', '32'))
L = [('conv_1_2', 300, 64, 64), ('conv_2_1', 150, 64, 128), ('conv_2_2', 150, 128, 128), ('conv_3_1', 75, 128, 256), ('conv_3_2', 75, 256, 256),
     ('conv_4_1', 38, 256, 512), ('conv_4_2', 38, 512, 512), ('conv_5_1', 19, 512, 512)]
only = os.environ.get('LAYERS')
def timeit(fn, reps=6):
    for _ in range(4):
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
    x = torch.randn(N, H, H, cin, device=DEV).relu_()
    w = torch.randn(cout, cin, 3, 3, device=DEV) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, device=DEV)
    buf, view = ops.haloed_bf16(N, H, H, cin, DEV)
    view[:, 1:-1, 1:-1, :] = x.bfloat16()
    wpk = ops.conv_bf16_pack(w)
    flops = 2.0 * N * H * H * cin * cout * 9
    line = f'{name:9s} H={H:3d} {cin:4d}->{cout:4d}'
    for zb in (False, True):
        best, med = timeit(lambda: ops.conv_bf16(buf, (N, H, H, cin), w, b, want_stats=True, wpk=wpk, z_bf16=zb))
        line += f' | z {"bf16" if zb else "fp32"} min {best:6.3f} med {med:6.3f} ms ({flops / med / 1e9:6.1f} TF)'
    for sk in [int(v) for v in os.environ.get('SKEWS', '').split(',') if v]:
        os.environ['SSDHIP_CB_SKEW'] = str(sk)
        best, med = timeit(lambda: ops.conv_bf16(buf, (N, H, H, cin), w, b, want_stats=True, wpk=wpk, z_bf16=True))
        del os.environ['SSDHIP_CB_SKEW']
        line += f' | skew {sk}: {med:6.3f}'
    sc, sh = torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV)
    b2, m2 = timeit(lambda: ops.conv2d_nhwc(x, w, b, stride=1, pad=1, in_scale=sc, in_shift=sh, want_stats=True, bf16=True))
    line += f' | igemm bf16 template med {m2:6.3f} ms ({flops / m2 / 1e9:6.1f} TF)'
    print(line, flush=True)
