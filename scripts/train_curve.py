"""300 train steps (batch 32, fp32 or DTYPE=bf16, all streams on, no host sync inside the loop) cycling over 8 fixed synthetic batches: the loss
curve and a finite-state check -- evidence that the whole step (mining, backward, Adam, pipelining) trains."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import synth
from object_detection_torch2_amd.model import SSD
from object_detection_torch2_amd.optim import FusedAdam
dev = 'cuda:0'
torch.manual_seed(1)
net = SSD(21).to(dev); df = net.default_bboxes.to(dev)
if os.environ.get('DTYPE') == 'bf16':          # BASELINE configs[3] form: bf16 forward / backward, fp32 master weights
    net.set_compute_dtype('bf16')
opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4, on_step=net.hip_weights_changed)
data = [(torch.from_numpy(synth.images(900 + k, 32)).to(dev), torch.from_numpy(synth.targets(950 + k, 32, g_max=8)).to(dev)) for k in range(8)]
torch.cuda.synchronize(); ev = torch.cuda.current_stream().record_event()
losses = []
t0 = time.perf_counter()
for it in range(300):
    ims, tgs = data[it % 8]
    opt.zero_grad(); loss = net.loss(net(ims, ready_event=ev), tgs, df); loss.backward(); opt.step(); losses.append(loss.detach())
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ls = torch.stack(losses).cpu().tolist()
print(f'300 steps in {dt:.2f} s = {300 * 32 / dt:.0f} img/s')
for k in range(0, 300, 20):
    print(f'steps {k:3d}-{k + 19:3d}: mean loss {sum(ls[k:k + 20]) / 20:.4f}')
bad = [k for k, v in net.state_dict().items() if not torch.isfinite(v.float()).all()]
print('non-finite tensors:', bad)
assert not bad and ls[-1] < 0.5 * ls[0]
