"""The opt-in split-bf16 form of the 128 x 128 plane GEMM (compute = 2: fp32 operands split in registers into three bf16 terms, six
products on the bf16 MFMA, fp32 accumulation) against the exact-fp32 MFMA kernels on the same planes: time and error vs fp64."""
import ctypes, os, sys
os.environ.setdefault('SSDHIP_PROBES', '1')     # the pre-split-weights form exists in the probe build only (make -C object_detection_torch2_amd/csrc probes)
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
SHAPES = [('conv_3_2', 11648, 256, 256), ('conv_4_1', 3200, 256, 512), ('conv_4_2', 3200, 512, 512), ('conv_6_1', 896, 512, 1024), ('det_4_3', 3200, 512, 128)]
for name, tp, cin, cout in SHAPES:
    rows_pad = (cout + 127) // 128 * 128
    g = torch.Generator(device=dev).manual_seed(1)
    U = torch.randn(36 * rows_pad * cin, device=dev, generator=g) * 0.02
    V = torch.randn(36 * tp * cin, device=dev, generator=g) * (torch.rand(36 * tp * cin, device=dev, generator=g) * 4)
    p = _lib.ConvParams()
    p.x, p.w, p.y_batch_stride = V.data_ptr(), U.data_ptr(), tp * cout
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 36, 1, tp, cin, 1, tp, cout
    p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = 1, 1, 1, 0, cin, 0, 0, 128128
    p.rows_per_img, p.w_img_stride = tp, rows_pad * cin
    st = _lib.stream()
    M0 = torch.full((36 * tp * cout,), float('nan'), device=dev)
    M2 = torch.full((36 * tp * cout,), float('nan'), device=dev)
    need = lib.ssdhip_conv2d_ws_bytes(ctypes.byref(p))
    ws = torch.zeros(max(1, (need + 3) // 4), device=dev)
    p.y, p.compute = M0.data_ptr(), 0
    if need > 0: p.ws, p.ws_bytes = ws.data_ptr(), ws.numel() * 4
    t0 = timed(lambda: _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st), 'gemm fp32'))
    p.y, p.compute, p.ws, p.ws_bytes = M2.data_ptr(), 2, None, 0
    t2 = timed(lambda: _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st), 'gemm split'))
    # pre-split weights (probe build): three bf16 term images of U made once, only V is split in the K loop
    t3s, M3 = {}, torch.full((36 * tp * cout,), float('nan'), device=dev)
    if hasattr(lib, 'ssdhip_probe_plane_gemm_bpre'):
        U3 = torch.empty(3 * U.numel(), dtype=torch.bfloat16, device=dev)
        _lib.check(lib.ssdhip_probe_split_bf16x3(ctypes.c_void_p(U.data_ptr()), ctypes.c_void_p(U3.data_ptr()), ctypes.c_longlong(U.numel()), st), 'split U')
        for tpb in (1, 2, 3):
            t3s[tpb] = timed(lambda: _lib.check(lib.ssdhip_probe_plane_gemm_bpre(ctypes.c_void_p(V.data_ptr()), ctypes.c_void_p(U3.data_ptr()), ctypes.c_void_p(M3.data_ptr()),
                36 * tp, cin, cout, tp, ctypes.c_longlong(rows_pad * cin), ctypes.c_longlong(U.numel()), tpb, st), 'gemm bpre'))
    torch.cuda.synchronize()
    assert torch.isfinite(M2).all()
    if t3s:
        assert torch.isfinite(M3).all()
        d = (M3 - M2).abs().max().item()
        tb = min(t3s.values())
        print(f'{name}: pre-split weights: ' + ' '.join(f'tpb {k}: {v:7.1f} us' for k, v in t3s.items()) + f'  best {2.0 * 36 * tp * cin * cout / tb / 1e6:6.1f} TF-equivalent; '
              f'max |M - M(split in registers)| = {d:.3e} (same terms, same order: expected 0)', flush=True)
    errs = []
    for pl, r0 in ((0, 0), (7, 64), (35, tp - 128)):
        ref = V.view(36, tp, cin)[pl, r0:r0 + 128].double() @ U.view(36, rows_pad, cin)[pl, :cout].double().t()
        sc = ref.abs().max().item()
        errs.append(((M0.view(36, tp, cout)[pl, r0:r0 + 128].double() - ref).abs().max().item() / sc, (M2.view(36, tp, cout)[pl, r0:r0 + 128].double() - ref).abs().max().item() / sc,
                     (M0.view(36, tp, cout)[pl, r0:r0 + 128].double() - ref).pow(2).mean().sqrt().item() / sc, (M2.view(36, tp, cout)[pl, r0:r0 + 128].double() - ref).pow(2).mean().sqrt().item() / sc))
    e = [max(x[i] for x in errs) for i in range(4)]
    gf = 2.0 * 36 * tp * cin * cout
    print(f'{name}: tp {tp} K {cin} N {cout}: fp32 MFMA ({"queue" if need > 0 else "slot"} kernel) {t0:7.1f} us ({gf / t0 / 1e6:6.1f} TF)   split-bf16 {t2:7.1f} us ({gf / t2 / 1e6:6.1f} TF-equivalent, x{t0 / t2:.2f})   '
          f'error vs fp64 of max |out|: fp32 max {e[0]:.2e} rms {e[2]:.2e}; split max {e[1]:.2e} rms {e[3]:.2e}', flush=True)
