import sys; sys.path.insert(0,'/root/repo')
import torch
from object_detection_torch2_amd import ops
x=torch.rand(32,300,300,4,device='cuda'); x[...,3]=0
w=torch.randn(64,3,3,3,device='cuda')*0.2; b=torch.zeros(64,device='cuda')
for name,fn in (('rgb', lambda: ops.conv_rgb3x3(x,w,b,want_stats=True)), ('igemm', lambda: ops.conv2d_nhwc(x,w,b,pad=1,want_stats=True))):
    fn(); torch.cuda.synchronize(); best=1e9
    for _ in range(8):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1))
    print(name, f'{best*1e3:.0f} us', f'{(737e6+46e6)/best/1e9:.2f} TB/s')
