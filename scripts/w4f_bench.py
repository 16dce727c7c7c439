"""Two-pass Winograd F(4x4) form (csrc/wino4_fused.hip) against the one-kernel F(2x2) form the engine runs on these layers, batch 32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import ops, _lib
DEV = 'cuda:0'
N = int(os.environ.get('N', '32'))
L = [('conv_1_2', 300, 64, 64, True), ('conv_2_1', 150, 64, 128, False), ('conv_2_2', 150, 128, 128, True), ('conv_3_1', 75, 128, 256, False)]
only = os.environ.get('LAYERS')
torch.manual_seed(0)
lib = _lib.load()
def timeit(fn, reps=8):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]
for name, H, cin, cout, pooled in L:
    if only and name not in only.split(','): continue
    x = torch.randn(N, H, H, cin, device=DEV)
    w = torch.randn(cout, cin, 3, 3, device=DEV) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, device=DEV)
    xf = pooled          # the pooled layers are the ones that read a BatchNorm.ReLU input
    sc, sh = (torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.1) if xf else (None, None)
    upk2 = ops.wino_fused_pack(w)
    upk4 = ops.wino4f_pack(w)
    vb = torch.empty(lib.ssdhip_wino4f_v_floats(N, H, H, cin), device=DEV)
    st = _lib.stream()
    t_f2 = timeit(lambda: (ops.wino_fused_conv_pool if pooled else ops.wino_fused_conv)(x, w, b, in_scale=sc, in_shift=sh, want_stats=True, upk=upk2))
    t_4 = timeit(lambda: ops.conv3x3_wino4f(x, w, b, in_scale=sc, in_shift=sh, want_stats=True, upk=upk4, pool=pooled, vb=vb))
    t_a = timeit(lambda: _lib.check(lib.ssdhip_wino4f_input(_lib.ptr(x), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(vb), N, H, H, cin, st), 'input'))
    tiles = N * ((H + 3) // 4) ** 2
    gf = 2.0 * 36 * tiles * cin * cout
    tb = t_4 - t_a
    vbytes = vb.numel() * 4 + x.numel() * 4
    print(f'{name:9s} H={H:3d} {cin:4d}->{cout:4d} {"pooled" if pooled else "      "}  F(2x2) one kernel {t_f2:6.3f} ms | two-pass F(4x4) {t_4:6.3f} ms = input {t_a:6.3f} '
          f'({vbytes / t_a / 1e9:5.2f} TB/s) + gemm ~{tb:6.3f} ({gf / tb / 1e9:6.1f} TF executed = {gf / tb / 1e9 / 157.3:.3f} of peak)   x{t_f2 / t_4:.2f}', flush=True)
