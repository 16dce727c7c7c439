"""One-off evidence at the headline batch: the full forward, loss and tail gradients at N = 32 against the CPU oracle (torch CPU ops
+ numpy loss restatement).  Takes a few minutes of host time, so it is a script, not a test."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ssd_oracle as O
from object_detection_torch2_amd import synth
from object_detection_torch2_amd.model import SSD
torch.set_num_threads(16)
DEV = 'cuda:0'
N = 32
ims, tg, df = synth.images(100, N), synth.targets(200, N, g_max=8), O.default_bboxes()
sd = O.init_state(1); names = O.trainable_param_names()
for k in names: sd[k].requires_grad_(True)
t0 = time.time()
out_o = O.forward(sd, torch.from_numpy(ims), training=True); o = out_o.detach().numpy()
lref = float(O.loss(o, tg, df)); out_o.backward(torch.from_numpy(O.loss_grad(o, tg, df)))
print(f'oracle forward + backward on the host: {time.time() - t0:.1f} s', flush=True)
net = SSD(21); net.load_state_dict(O.init_state(1)); net.to(DEV)
net.set_compute_dtype(os.environ.get('DTYPE', 'fp32'))          # DTYPE=bf16: the same comparison for BASELINE configs[3]'s form (fp32 oracle)
out = net(torch.from_numpy(ims).to(DEV)); loss = net.loss(out, torch.from_numpy(tg).to(DEV), torch.from_numpy(df).to(DEV)); loss.backward()
err = np.abs(out.detach().cpu().numpy() - o)
print(f'N = {N}: max |out - oracle| = {err.max():.3e} (mean {err.mean():.2e}, |out| max {np.abs(o).max():.2f}); loss {loss.item():.6f} vs oracle {lref:.6f} (diff {abs(loss.item() - lref):.2e})')
got = dict(net.named_train_params()); worst = (0.0, '')
for k in names:
    if 'features.conv' in k and k.endswith('.bias'): continue
    g, r = got[k].grad.cpu().double(), sd[k].grad.double()
    rel = ((g - r).norm() / r.norm()).item()
    worst = max(worst, (rel, k))
print(f'largest relative L2 gradient difference over the {len(names)} trainable tensors: {worst[0]:.2e} ({worst[1]})')
sdn = net.state_dict()
print('running stats: max rel diff', max(((sdn[k].cpu() - sd[k]).abs().max() / sd[k].abs().max().clamp_min(1e-6)).item() for k in sdn if k.endswith('running_var') or k.endswith('running_mean')))
