"""Data-parallel + stream pipelining rehearsal on ONE GPU (2 ranks over gloo): several steps without host synchronisation, with the
backbone overlap enabled; the run must be bit-identical to the same run with SSD_SINGLE_STREAM semantics and across ranks.
Launch: SSD_DIST_BACKEND=gloo SSD_FORCE_DEVICE=0 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tests/tools/dp_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from oracle import ssd_oracle as O
from object_detection_torch2_amd import parallel, synth
from object_detection_torch2_amd.model import SSD
from object_detection_torch2_amd.optim import FusedAdam
rank, world, local = parallel.init_from_env()
dev = torch.device('cuda', local)
N, STEPS = 8, 6
data = [(torch.from_numpy(synth.images(500 + k, N * world)[rank * N:(rank + 1) * N]).to(dev),
         torch.from_numpy(synth.targets(600 + k, N * world, g_max=8)[rank * N:(rank + 1) * N]).to(dev)) for k in range(3)]
def run(single):
    net = SSD(21); net.load_state_dict(O.init_state(1)); net.to(dev)
    net._engine().single_stream = single
    parallel.attach(net)
    opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4, grad_scale=1.0 / world, on_step=net.hip_weights_changed)
    df = net.default_bboxes.to(dev)
    torch.cuda.synchronize(); ev = None if single else torch.cuda.current_stream().record_event()
    losses = []
    for it in range(STEPS):
        ims, tgs = data[it % 3]
        opt.zero_grad(); loss = net.loss(net(ims, ready_event=ev), tgs, df); loss.backward(); opt.step(); losses.append(loss)
    w = torch.cat([p.detach().reshape(-1) for p in net.train_params()])
    return [l.item() for l in losses], w
l0, w0 = run(True)
l1, w1 = run(False)
ws = [torch.empty_like(w1) for _ in range(world)]
dist.all_gather(ws, w1)
ok = l0 == l1 and torch.equal(w0, w1) and all(torch.equal(ws[0], x) for x in ws) and bool(torch.isfinite(w1).all())
print(f'rank {rank}: pipelined == serialised: {l0 == l1 and torch.equal(w0, w1)}; ranks identical: {all(torch.equal(ws[0], x) for x in ws)}; losses {l1[0]:.4f} -> {l1[-1]:.4f}', flush=True)
assert ok
dist.barrier(); dist.destroy_process_group()
