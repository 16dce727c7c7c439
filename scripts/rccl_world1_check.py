"""RCCL on the one GPU this builder has: a ONE-rank `nccl` process group (torch's ProcessGroupNCCL = RCCL on ROCm) carrying the real
gradient all-reduce pattern of a data-parallel train step -- communicator creation on this image, asynchronous work handles, the
ordering of RCCL's stream against the engine's three streams, the early bucket issued next to the last weight gradient.  With one rank
the sum is the identity, so gradients and updated weights must be bit-identical to the same steps without the hook.
(Two ranks on one GPU are refused by RCCL -- 'Duplicate GPU detected' -- which is why the two-rank rehearsals use gloo.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29541')
import torch, torch.distributed as dist
from object_detection_torch2_amd import parallel, synth
from object_detection_torch2_amd.model import SSD
from object_detection_torch2_amd.optim import FusedAdam

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend='nccl', rank=0, world_size=1)
assert dist.get_backend() == 'nccl'
B, STEPS = 8, 3


def run(hooked):
    torch.manual_seed(3)
    net = SSD(num_classes=21).to(dev)
    if hooked:
        hook = parallel.GradAllReduce(n_buckets=2, force=True)
        net._dist_hook = hook
        net._engine().dist_hook = hook
    df = net.default_bboxes.to(dev)
    opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4, on_step=net.hip_weights_changed)
    losses = []
    for k in range(STEPS):                       # no host synchronisation between the steps: backbone of step k+1 overlaps step k's tail
        ims = torch.from_numpy(synth.images(20 + k, B)).to(dev)
        tg = torch.from_numpy(synth.targets(30 + k, B, g_max=6)).to(dev)
        opt.zero_grad()
        loss = net.loss(net(ims), tg, df)
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    return [l.item() for l in losses], net._engine()._flat['buf'].clone(), torch.cat([p.detach().reshape(-1) for p in net.train_params()])


t = torch.arange(1000, device=dev, dtype=torch.float32)
dist.all_reduce(t)                                # communicator really created, collective really launched
assert torch.equal(t, torch.arange(1000, device=dev, dtype=torch.float32))
l0, g0, w0 = run(False)
l1, g1, w1 = run(True)
same = l0 == l1 and torch.equal(g0, g1) and torch.equal(w0, w1)
print(f'RCCL one-rank rehearsal: backend {dist.get_backend()}, losses {l1}, identical to the unhooked steps: {same}')
dist.barrier(device_ids=[0])
dist.destroy_process_group()
assert same
