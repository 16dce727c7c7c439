"""Conditioning check: HIP gradients and torch-fp32 gradients, both against an fp64 run of the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ssd_oracle as O
from object_detection_torch2_amd import synth
from object_detection_torch2_amd.model import SSD
torch.set_num_threads(16)
DEV='cuda:0'
import torch.nn.functional as F
def fwd64(sd, x):
    # fp64 version of O.forward
    mean = torch.tensor(O.MEAN, dtype=torch.float32).double().reshape(1,3,1,1); std = torch.tensor(O.STD, dtype=torch.float32).double().reshape(1,3,1,1)
    x = x.double().sub(mean).div(std)
    dets = {d['src']: d for d in O.detector_layers(21)}; ys=[]
    for l in O.feature_layers():
        p='features.'+l['name']
        if l['kind']=='conv': x=F.conv2d(x, sd[p+'.weight'], sd[p+'.bias'], stride=l['stride'], padding=l['pad'])
        elif l['kind']=='bn': x=F.batch_norm(x, None, None, sd[p+'.weight'], sd[p+'.bias'], training=True, eps=1e-5)
        elif l['kind']=='act':
            x=F.relu(x)
            if l['name'] in dets:
                q='detectors.'+dets[l['name']]['name']; y=F.conv2d(x, sd[q+'.weight'], sd[q+'.bias'], padding=1)
                ys.append(y.permute(0,2,3,1).reshape(x.shape[0],-1,25))
        else: x=F.max_pool2d(x,2,2,padding=l['pad'])
    return torch.cat(ys,1)
for N in (4,):
    net = SSD(21); net.load_state_dict(O.init_state(1)); net.to(DEV)
    sd = O.init_state(1); names = O.trainable_param_names()
    ims = synth.images(40, N); tg = synth.targets(50, N, g_max=8); df = O.default_bboxes()
    for k in names: sd[k].requires_grad_(True)
    sd64 = {k: (v.detach().double().requires_grad_(k in names) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    out_o = O.forward(sd, torch.from_numpy(ims), training=True)
    o = out_o.detach().numpy(); gl = O.loss_grad(o, tg, df)
    out_o.backward(torch.from_numpy(gl))
    out64 = fwd64(sd64, torch.from_numpy(ims)); out64.backward(torch.from_numpy(gl).double())
    out = net(torch.from_numpy(ims).to(DEV)); out.backward(torch.from_numpy(gl).to(DEV))
    print('N', N, 'fwd err hip-vs-64', np.abs(out.detach().cpu().numpy()-out64.detach().numpy()).max(), 'torch32-vs-64', np.abs(o-out64.detach().numpy()).max())
    got = dict(net.named_train_params())
    for k in names:
        if 'features.conv' in k and k.endswith('bias'): continue
        g, r, r64 = got[k].grad.cpu().double(), sd[k].grad.double(), sd64[k].grad
        s = max(r64.abs().max().item(),1e-12)
        print(f'  {k:30s} hip-vs-64 {(g-r64).abs().max().item()/s:.2e}  torch32-vs-64 {(r-r64).abs().max().item()/s:.2e}  l2: hip {((g-r64).norm()/r64.norm()).item():.2e} torch {((r-r64).norm()/r64.norm()).item():.2e}')
