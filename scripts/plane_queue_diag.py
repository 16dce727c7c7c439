"""Control words and completeness of the work-queue plane GEMM after each of a few launches on a fresh workspace."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import _lib
lib = _lib.load(); dev = 'cuda:0'
for name, tp, cin, cout in [('conv_3_2', 11648, 256, 256), ('conv_4_2', 3200, 512, 512)]:
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
    for k in range(4):
        M.fill_(float('nan'))
        torch.cuda.synchronize()
        _lib.check(lib.ssdhip_conv2d_nhwc(ctypes.byref(p), st), 'gemm')
        torch.cuda.synchronize()
        ctl = ws[:1024].view(torch.int32).cpu()
        nan_tiles = int(torch.isnan(M.view(-1, 128, cout)[:, 0, ::128]).sum())
        print(f'{name} launch {k}: queue heads {ctl[:8].tolist()} exited {int(ctl[8])} piece counters nonzero {int((ctl[64:576] != 0).sum())} tiles left NaN {nan_tiles}', flush=True)
