"""Timeline of the work-queue plane GEMM (PQ_TRACE build of gemm_planes.hip in lib/libssdhip_ab.so): per block start / item ends."""
import ctypes, os, sys
os.environ['SSDHIP_PROBES'] = '1'; os.environ['SSDHIP_PROBE_LIB'] = 'libssdhip_ab.so'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import _lib
lib = _lib.load(); dev = 'cuda:0'
for name, tp, cin, cout in [('conv_6_1', 896, 512, 1024), ('det_4_3', 3200, 512, 128), ('conv_4_2', 3200, 512, 512)]:
    rows_pad = (cout + 127) // 128 * 128
    U = torch.randn(36 * rows_pad * cin, device=dev) * 0.02
    V = torch.randn(36 * tp * cin, device=dev)
    M = torch.empty(36 * tp * cout, device=dev)
    p = _lib.ConvParams()
    p.x, p.w, p.y, p.y_batch_stride = V.data_ptr(), U.data_ptr(), M.data_ptr(), tp * cout
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 36, 1, tp, cin, 1, tp, cout
    p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.tile = 1, 1, 1, 0, cin, 0, 0, 128128
    p.rows_per_img, p.w_img_stride = tp, rows_pad * cin
    need = lib.ssdhip_conv2d_ws_bytes(ctypes.byref(p))
    ws = torch.zeros((need + 3) // 4, device=dev)
    p.ws = ws.data_ptr(); p.ws_bytes = ws.numel() * 4
    st = _lib.stream()
    for _ in range(3):
        _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st), 'gemm')
    torch.cuda.synchronize()
    S = 64
    off = (16384 + 8 * S * 128 * 128 * 4) // 4
    tr = ws[off:off + 512 * 64 * 2].view(torch.int64).view(512, 64).cpu()
    t0 = int(tr[:, 0].min())
    tick = 1e-2      # wall_clock64: 100 MHz -> 10 ns
    print(f'--- {name}: start spread {(int(tr[:, 0].max()) - t0) * tick:.1f} us')
    ends = []
    for b in (0, 1, 8, 9, 63 * 8, 511):
        n = int(tr[b, 63])
        ev = [(int(tr[b, i]) - t0) * tick for i in range(0, n + 1, 2)]
        ids = [int(tr[b, i]) for i in range(1, n, 2)]
        print(f'  block {b:3d} (xcd {b & 7}): times {[round(e, 1) for e in ev]} ids {ids}')
    last = max((int(tr[b, int(tr[b, 63])]) - t0) * tick for b in range(512))
    durs = []
    for b in range(512):
        n = int(tr[b, 63])
        ts = [int(tr[b, i]) for i in range(0, n + 1, 2)]
        durs += [(y - x) * tick for x, y in zip(ts[:-1], ts[1:])][: max(0, len(ts) - 2)]
    durs = torch.tensor(durs)
    print(f'  last block exit {last:.1f} us; item durations: n {len(durs)} mean {durs.mean():.1f} min {durs.min():.1f} max {durs.max():.1f} us')
