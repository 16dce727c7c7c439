"""Work-queue plane GEMM (persistent blocks, per-XCD queues, split last round) against the one-slot-per-block kernel: same results (bit-equal
on whole tiles; the split remainder tiles differ by summation order only), time per launch, bit-reproducibility over repeated launches."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import _lib
lib = _lib.load(); dev = 'cuda:0'
REPS = int(os.environ.get('REPS', '20'))
def timed(fn):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3
SHAPES = [('conv_3_2', 11648, 256, 256), ('conv_4_1', 3200, 256, 512), ('conv_4_2', 3200, 512, 512), ('conv_6_1', 896, 512, 1024), ('det_4_3', 3200, 512, 128),
          ('odd', 1408, 256, 320), ('tiny', 128, 256, 128)]
for name, tp, cin, cout in SHAPES:
    rows_pad = (cout + 127) // 128 * 128
    g = torch.Generator(device=dev).manual_seed(1)
    U = torch.randn(36 * rows_pad * cin, device=dev, generator=g) * 0.02
    V = torch.randn(36 * tp * cin, device=dev, generator=g)
    p = _lib.ConvParams()
    p.x, p.w, p.y_batch_stride = V.data_ptr(), U.data_ptr(), tp * cout
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 36, 1, tp, cin, 1, tp, cout
    p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = 1, 1, 1, 0, cin, 0, 0, 128128
    p.rows_per_img, p.w_img_stride = tp, rows_pad * cin
    st = _lib.stream()
    need = lib.ssdhip_conv2d_ws_bytes(ctypes.byref(p))
    if need <= 0:
        print(f'{name}: tp {tp} K {cin} N {cout}: the host rule keeps the slot kernel'); continue
    ws = torch.zeros((need + 3) // 4, device=dev)
    M0 = torch.full((36 * tp * cout,), float('nan'), device=dev)
    M1 = torch.full((36 * tp * cout,), float('nan'), device=dev)
    p.y = M0.data_ptr(); p.ws = None; p.ws_bytes = 0
    t0 = timed(lambda: _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st), 'gemm'))
    p.y = M1.data_ptr(); p.ws = ws.data_ptr(); p.ws_bytes = ws.numel() * 4
    t1 = timed(lambda: _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st), 'gemm'))
    torch.cuda.synchronize()
    first = M1.clone()
    # the block -> XCD placement is round-robin over ALL dispatches of the device: k one-block launches in front shift it by k, so that
    # blocks sharing a queue (blockIdx & 7) sit on different XCDs -- results and control words must not depend on it
    step_ctr, hyper = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(8, device=dev)
    for shift in (0, 1, 3, 5, 7, 2):
        M1.fill_(float('nan'))
        for _ in range(shift):
            _lib.check(lib.ssdhip_adam_tick(_lib.ptr(step_ctr), _lib.ptr(hyper), st), 'tick')
        _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st), 'gemm')
        torch.cuda.synchronize()
        assert torch.equal(M1, first), f'{name}: result depends on the dispatch phase (shift {shift}): {int((M1 != first).sum())} elements differ'
        assert bool((ws[:1024].view(torch.int32)[:576] == 0).all()), f'{name}: control words not zero after shift {shift}'
    ctl = ws[:1024].view(torch.int32)
    diff = (M0 - M1).abs()
    neq = int((M0 != M1).sum())
    ref = (V.view(36, tp, cin)[7, :64].double() @ U.view(36, rows_pad, cin)[7, :cout].double().t()).float()
    err = (M1.view(36, tp, cout)[7, :64] - ref).abs().max().item() / ref.abs().max().item()
    gf = 2.0 * 36 * tp * cin * cout
    print(f'{name}: tp {tp} K {cin} N {cout}: slot kernel {t0:7.1f} us ({gf / t0 / 1e6:6.1f} TF)  queue kernel {t1:7.1f} us ({gf / t1 / 1e6:6.1f} TF)  '
          f'finite {bool(torch.isfinite(M1).all())}  elements differing {neq} / {M0.numel()} (max {diff.max().item():.2e})  vs fp64 {err:.1e}  '
          f'control words zero {bool((ctl == 0).all())}', flush=True)
