"""Data-parallel rehearsal on ONE GPU (2 ranks over gloo, both on cuda:0): the all-reduced flat gradient must equal the
sum of the two ranks' local gradients, and after one FusedAdam step with grad_scale = 1/2 both ranks hold identical weights.
Launch: SSD_DIST_BACKEND=gloo SSD_FORCE_DEVICE=0 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/dp_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from object_detection_torch2_amd import parallel, synth
from object_detection_torch2_amd.model import SSD
from object_detection_torch2_amd.optim import FusedAdam
rank, world, local = parallel.init_from_env()
dev = torch.device('cuda', local)
from object_detection_torch2_amd import train
net = train.build_model(dev, world)          # UNSEEDED on purpose: train.py's path must make the replicas equal (broadcast from rank 0)
w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()] + [b.detach().reshape(-1).float() for b in net.buffers()])
w0s = [torch.empty_like(w0) for _ in range(world)]
dist.all_gather(w0s, w0)
assert all(torch.equal(w0s[0], x) for x in w0s), 'replicas start from different weights'
df = net.default_bboxes.to(dev)
lo, hi = parallel.shard_batch(4 * world, rank, world)
ims = torch.from_numpy(synth.images(5, 4 * world)[lo:hi]).to(dev)
tgs = torch.from_numpy(synth.targets(6, 4 * world)[lo:hi]).to(dev)
# local gradients, no exchange
loss = net.loss(net(ims), tgs, df); loss.backward()
local_flat = net._engine()._flat['buf'].clone()
gathered = [torch.empty_like(local_flat) for _ in range(world)]
dist.all_gather(gathered, local_flat)
expect = sum(gathered)
for p in net.parameters(): p.grad = None
# with the overlapped all-reduce hook
parallel.attach(net)
opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4, grad_scale=1.0 / world, on_step=net.hip_weights_changed)
opt.zero_grad()
loss2 = net.loss(net(ims), tgs, df); loss2.backward()
got = net._engine()._flat['buf']
err = (got - expect).abs().max().item()
opt.step()
w = torch.cat([p.detach().reshape(-1) for p in net.train_params()])
ws = [torch.empty_like(w) for _ in range(world)]
dist.all_gather(ws, w)
same = all(torch.equal(ws[0], x) for x in ws)
if rank == 0:
    print(f'DP check: max|allreduced - sum(local)| = {err:.3e} (scale {expect.abs().max().item():.3e}); weights identical across ranks after step: {same}')
    assert err <= 1e-6 * max(1.0, expect.abs().max().item()) and same
dist.barrier(); dist.destroy_process_group()
