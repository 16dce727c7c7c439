"""Graph-replayed data-parallel steps on the one GPU of the box: a ONE-rank `nccl` group (RCCL), the engine's gradient hook forced on.
The captured tail must be cut at the two hook points (3 graph segments + 2 eager collectives) and match the eager hooked steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29543')
import torch, torch.distributed as dist
from object_detection_torch2_amd import parallel, synth
from object_detection_torch2_amd.graph_step import GraphedTrainStep
from object_detection_torch2_amd.model import SSD
from object_detection_torch2_amd.optim import FusedAdam

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend='nccl', rank=0, world_size=1)
B, STEPS = 4, 4
batches = [(torch.from_numpy(synth.images(20 + k, B)).to(dev), torch.from_numpy(synth.targets(30 + k, B, g_max=6)).to(dev)) for k in range(STEPS)]


def make():
    torch.manual_seed(3)
    net = SSD(num_classes=21).to(dev)
    hook = parallel.GradAllReduce(n_buckets=2, force=True)
    net._dist_hook = hook
    net._engine().dist_hook = hook
    return net, FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4, on_step=net.hip_weights_changed)


net, opt = make()
df = net.default_bboxes.to(dev)
le = []
for ims, tg in batches:
    opt.zero_grad()
    loss = net.loss(net(ims), tg, df)
    loss.backward()
    opt.step()
    le.append(loss.detach().clone())
torch.cuda.synchronize()
we = torch.cat([p.detach().reshape(-1) for p in net.train_params()])

net2, opt2 = make()
runner = GraphedTrainStep(net2, opt2, df, batches[0][0].shape, batches[0][1].shape)
kinds = ['op' if isinstance(i, tuple) else 'graph' for i in runner._tail_items[0]]
print(f'segments per tail: {kinds.count("graph")} graphs + {kinds.count("op")} collectives')
lg = []
runner.feed(batches[0][0])
for k, (_, tg) in enumerate(batches):
    if k + 1 < STEPS:
        runner.feed(batches[k + 1][0])
    lg.append(runner.step(tg).clone())
runner.drain()
torch.cuda.synchronize()
wg = torch.cat([p.detach().reshape(-1) for p in net2.train_params()])
same = [a.item() for a in le] == [a.item() for a in lg] and torch.equal(we, wg)
print(f'losses {[round(a.item(), 5) for a in lg]}; graphed == eager (hooked): {same}')
dist.barrier(device_ids=[0])
dist.destroy_process_group()
assert same
