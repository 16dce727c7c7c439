import torch, time
dev='cuda:0'
n=737280000//4
a=torch.empty(n,device=dev); b=torch.empty(n,device=dev)
def timed(fn,reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
t=timed(lambda: a.fill_(1.0)); print(f'fill 737 MB: {t:.1f} us = {0.737e3/t:.2f} TB/s write')
t=timed(lambda: b.copy_(a)); print(f'copy 737 MB: {t:.1f} us = {2*0.737e3/t:.2f} TB/s read+write')
t=timed(lambda: a.sum()); print(f'sum 737 MB: {t:.1f} us = {0.737e3/t:.2f} TB/s read')
