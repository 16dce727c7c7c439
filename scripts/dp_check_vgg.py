"""Data-parallel rehearsal of the classification pre-training path on ONE GPU (2 ranks over gloo, both on cuda:0): the classifier
stack's gradients go through the same asynchronous all-reduce hook as the feature gradients (parallel.GradAllReduce.launch), each
rank drops different units (dropout_seed = rank), and after one FusedAdam step with grad_scale = 1/2 both ranks hold identical weights.
Launch: SSD_DIST_BACKEND=gloo SSD_FORCE_DEVICE=0 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/dp_check_vgg.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from object_detection_torch2_amd import parallel, synth
from object_detection_torch2_amd.model import VGG16
from object_detection_torch2_amd.optim import FusedAdam
rank, world, local = parallel.init_from_env()
dev = torch.device('cuda', local)
torch.cuda.set_device(dev)
net = VGG16(num_classes=20, transfer_learning=True).unfreeze_features().to(dev)      # UNSEEDED: the broadcast must make the replicas equal
parallel.broadcast_module(net)
net.dropout_seed = rank
B = 2
lo, hi = parallel.shard_batch(B * world, rank, world)
ims = torch.from_numpy(synth.images(11, B * world, 200)[lo:hi]).to(dev)
labels = torch.tensor([3, 17, 5, 9][lo:hi])
tg = torch.nn.functional.one_hot(labels, 20).float().to(dev)
params = list(net.train_params())


def flat_grads():
    return torch.cat([p.grad.reshape(-1) for p in params])


# local gradients, no exchange
net._calls = 0
net.loss(net(ims), tg, nll=True).backward()
local_g = flat_grads().clone()
mask_digest = net._saved is None            # consumed
gathered = [torch.empty_like(local_g) for _ in range(world)]
dist.all_gather(gathered, local_g)
expect = sum(gathered)
differ = not torch.equal(gathered[0], gathered[-1])
for p in params:
    p.grad = None
# with the hook: features through the engine's flat buffer, classifier stack through hook.launch
parallel.attach(net)
opt = FusedAdam(params, lr=1e-3, weight_decay=5e-4, grad_scale=1.0 / world, on_step=net.hip_weights_changed)
opt.zero_grad()
net._calls = 0
net.loss(net(ims), tg, nll=True).backward()
got = flat_grads()
err = (got - expect).abs().max().item()
scale = expect.abs().max().item()
opt.step()
w = torch.cat([p.detach().reshape(-1) for p in params])
ws = [torch.empty_like(w) for _ in range(world)]
dist.all_gather(ws, w)
same = all(torch.equal(ws[0], x) for x in ws)
if rank == 0:
    print(f'VGG16 DP check: max|allreduced - sum(local)| = {err:.3e} (scale {scale:.3e}); local gradients differ across ranks: {differ}; '
          f'weights identical across ranks after step: {same}')
    assert err <= 1e-6 * max(1.0, scale) and same and differ
dist.barrier(); dist.destroy_process_group()
