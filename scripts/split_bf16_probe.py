"""Probe (NOT a product path, not the headline's arithmetic): the Winograd-domain plane GEMM with each fp32 operand split
into three bf16 terms (8 + 8 + 8 significant bits: a = a1 + a2 + a3 exactly) and the six products a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 run on the
bf16 MFMA with fp32 accumulation (16x the fp32 MFMA rate per product: 6 products = 2.7x fewer matrix cycles).  Measured here with what the
library already has: the BF16 template of the implicit-GEMM kernel (operands rounded to bf16 at staging -- exact for pre-split terms) as
six accumulating 1x1 launches over the conv_4_2 plane shape, against the exact-fp32 MFMA kernel and an fp64 product.
    python scripts/split_bf16_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import _lib, ops
lib = _lib.load(); dev = 'cuda:0'
REPS = 10


def split3(t):
    a1 = t.bfloat16().float(); r = t - a1
    a2 = r.bfloat16().float(); r = r - a2
    a3 = r.bfloat16().float()
    return a1, a2, a3, (r - a3)


def launch(x, packed, KP, y, cout, accumulate, bf16):
    N, H, W, C = x.shape
    p = _lib.ConvParams()
    p.x, p.w, p.y, p.y_batch_stride = x.data_ptr(), packed.data_ptr(), y.data_ptr(), H * W * cout
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = N, H, W, C, H, W, cout
    p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile, p.compute = 1, 1, 1, 0, KP, 0, int(accumulate), 128128, int(bf16)
    _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), _lib.stream()), 'gemm')


def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


for name, rows, cin, cout in (('conv_4_2 planes (36 x 3200 rows, K 512, N 512)', 36 * 3200, 512, 512), ('conv_3_2 planes (36 x 11648 rows, K 256, N 256)', 36 * 11648, 256, 256)):
    g = torch.Generator(device=dev).manual_seed(3)
    # Winograd-domain operands: V = B^T d B of relu'd activations (heavy-tailed, mixed sign), U = G g G^T of kaiming weights
    x = (torch.randn(1, rows // 64, 64, cin, device=dev, generator=g) * torch.rand(1, rows // 64, 64, 1, device=dev, generator=g) * 4)
    w = torch.randn(cout, cin, 1, 1, device=dev, generator=g) * 0.03
    xs, ws = split3(x), split3(w)
    assert float(xs[3].abs().max()) == 0.0 and float(ws[3].abs().max()) == 0.0          # three bf16 terms hold all 24 bits
    packs = [ops.pack_weight(wi)[0] for wi in (w,) + ws[:3]]
    KP = ops.pack_weight(w)[1]
    y32 = torch.empty(1, rows // 64, 64, cout, device=dev)
    y6 = torch.empty_like(y32); y3 = torch.empty_like(y32); y1 = torch.empty_like(y32)
    order6 = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]                             # small terms first
    def run6(y, order):
        for k, (i, j) in enumerate(order):
            launch(xs[i].contiguous(), packs[1 + j], KP, y, cout, k > 0, True)
    xs = tuple(t.contiguous() for t in xs[:3])
    t32 = timed(lambda: launch(x, packs[0], KP, y32, cout, False, False))
    t6 = timed(lambda: run6(y6, order6))
    run6(y3, order6[3:]); launch(xs[0], packs[1], KP, y1, cout, False, True)
    t1 = timed(lambda: launch(xs[0], packs[1], KP, y1, cout, False, True))
    torch.cuda.synchronize()
    sub = slice(0, 8)                                                                     # 512 rows against fp64
    ref = (x[0, sub].reshape(-1, cin).double() @ w.view(cout, cin).double().t())
    scale = ref.abs().max().item()
    def err(y):
        d = (y[0, sub].reshape(-1, cout).double() - ref).abs()
        return d.max().item() / scale, d.pow(2).mean().sqrt().item() / scale
    gf = 2.0 * rows * cin * cout
    print(f'{name}:')
    print(f'  exact-fp32 MFMA (one launch)            {t32:7.1f} us  {gf / t32 / 1e6:6.1f} TF   error vs fp64: max {err(y32)[0]:.2e} rms {err(y32)[1]:.2e} of max |out|')
    print(f'  split-bf16, 6 products (six launches)   {t6:7.1f} us  {gf / t6 / 1e6:6.1f} TF-equivalent   error: max {err(y6)[0]:.2e} rms {err(y6)[1]:.2e}')
    print(f'  split-bf16, 3 products (a1b1+a1b2+a2b1)                                     error: max {err(y3)[0]:.2e} rms {err(y3)[1]:.2e}')
    print(f'  plain bf16 (a1b1 only, one launch)      {t1:7.1f} us  {gf / t1 / 1e6:6.1f} TF   error: max {err(y1)[0]:.2e} rms {err(y1)[1]:.2e}', flush=True)
