"""Per-stage activation error of the HIP forward and of torch fp32, both vs an fp64 forward; ReLU mask flips."""
import os; os.environ.setdefault('SSD_NO_POOL_FUSE', '1')      # this tool reads every stage's z: keep conv_1_2 / conv_2_2 materialising it
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from oracle import ssd_oracle as O
from object_detection_torch2_amd import synth
from object_detection_torch2_amd.model import SSD
torch.set_num_threads(16)
DEV='cuda:0'
def fwd(sd, x, dt):
    mean = torch.tensor(O.MEAN, dtype=torch.float32).to(dt).reshape(1,3,1,1); std = torch.tensor(O.STD, dtype=torch.float32).to(dt).reshape(1,3,1,1)
    x = x.to(dt).sub(mean).div(std); acts={}; zs={}
    for l in O.feature_layers():
        p='features.'+l['name']
        if l['kind']=='conv': x=F.conv2d(x, sd[p+'.weight'].to(dt), sd[p+'.bias'].to(dt), stride=l['stride'], padding=l['pad']); zs[l['name']]=x
        elif l['kind']=='bn': x=F.batch_norm(x, None, None, sd[p+'.weight'].to(dt), sd[p+'.bias'].to(dt), training=True, eps=1e-5); acts[l['name']]=x
        elif l['kind']=='act': x=F.relu(x)
        else: x=F.max_pool2d(x,2,2,padding=l['pad'])
    return zs, acts
N=4
net = SSD(21); net.load_state_dict(O.init_state(1)); net.to(DEV)
sd = O.init_state(1)
ims = synth.images(40, N)
with torch.no_grad():
    z64, y64 = fwd(sd, torch.from_numpy(ims), torch.float64)
    z32, y32 = fwd(sd, torch.from_numpy(ims), torch.float32)
    net(torch.from_numpy(ims).to(DEV))
eng = net._engine(); b = list(eng._bufs.values())[0]
for i, s in enumerate(eng.stages):
    z = b['z'][i].permute(0,3,1,2).cpu().double()
    st = eng._bn[s.bn_name]
    y = (b['z'][i] * st['scale'] + st['shift']).permute(0,3,1,2).cpu().double()
    r = z64[s.conv.name]; ry = y64[s.bn_name]
    zs = r.abs().max().item()
    fl_h = ((y > 0) != (ry > 0)).sum().item(); fl_t = ((y32[s.bn_name] > 0) != (ry > 0)).sum().item()
    print(f'{s.conv.name:10s} M={r.numel()//r.shape[1]:8d} z: hip {((z-r).abs().max()/zs).item():.1e} t32 {((z32[s.conv.name].double()-r).abs().max()/zs).item():.1e} | y: hip {(y-ry).abs().max().item():.1e} t32 {(y32[s.bn_name].double()-ry).abs().max().item():.1e} | flips hip {fl_h} t32 {fl_t} of {ry.numel()}')
