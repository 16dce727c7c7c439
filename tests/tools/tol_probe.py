"""Measures what the tightened tests assert (run on the GPU box): (a) parameters after the 3 golden Adam steps vs the reference
fixture, per tensor; (b) per-tensor gradient error of the HIP path and of torch fp32, both against an fp64 run of the oracle, with
one common dL/d outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ssd_oracle as O
from object_detection_torch2_amd import synth
from object_detection_torch2_amd.model import SSD
from object_detection_torch2_amd.optim import FusedAdam
torch.set_num_threads(16)
DEV = 'cuda:0'
fx = np.load(os.path.join(os.path.dirname(__file__), '..', 'golden', 'g9_train.npz'))
net = SSD(21); net.load_state_dict(O.init_state(1)); net.to(DEV)
init = {k: v.clone() for k, v in O.init_state(1).items()}
opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4)
df = net.default_bboxes.to(DEV)
for step in range(3):
    ims = torch.from_numpy(synth.images(40 + step, 4)).to(DEV); tg = torch.from_numpy(synth.targets(50 + step, 4, g_max=8)).to(DEV)
    opt.zero_grad(); loss = net.loss(net(ims), tg, df); loss.backward(); opt.step()
    print('step', step, loss.item(), fx['losses'][step])
sd = net.state_dict()
for k in [n[2:] for n in fx.files if n.startswith('p_')]:
    ref = fx['p_' + k]; got = sd[k].cpu().numpy().reshape(-1)[::101]; ini = init[k].numpy().reshape(-1)[::101]
    d = np.abs(got - ref)
    print(f'{k:32s} max|hip-ref| {d.max():.3e}  p99 {np.quantile(d, 0.99):.3e} median {np.median(d):.3e} frac>2e-5 {(d > 2e-5).mean():.4f}  max|ref-init| {np.abs(ref - ini).max():.3e}')

for N in (2,):
    ims = synth.images(40, N); tg = synth.targets(50, N, g_max=8); dfn = O.default_bboxes()
    names = O.trainable_param_names()
    sd32 = O.init_state(1); sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in O.init_state(1).items()}
    for k in names:
        sd32[k].requires_grad_(True); sd64[k].requires_grad_(True)
    o32 = O.forward(sd32, torch.from_numpy(ims), training=True)
    gout = torch.from_numpy(O.loss_grad(o32.detach().numpy(), tg, dfn))
    o32.backward(gout)
    o64 = O.forward(sd64, torch.from_numpy(ims).double(), training=True)
    o64.backward(gout.double())
    net = SSD(21); net.load_state_dict(O.init_state(1)); net.to(DEV)
    out = net(torch.from_numpy(ims).to(DEV)); out.backward(gout.to(DEV))
    got = dict(net.named_train_params())
    print('out err hip', (out.detach().cpu().double() - o64.detach()).abs().max().item(), 't32', (o32.detach().double() - o64.detach()).abs().max().item())
    for k in names:
        r = sd64[k].grad; eh = (got[k].grad.cpu().double() - r).norm().item(); et = (sd32[k].grad.double() - r).norm().item(); n = r.norm().item()
        print(f'N={N} {k:32s} |g64| {n:.3e}  hip {eh / max(n, 1e-30):.3e}  t32 {et / max(n, 1e-30):.3e}  ratio {eh / max(et, 1e-30):.2f}')
