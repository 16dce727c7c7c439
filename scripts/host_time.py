"""How long does the host need to enqueue one train step (no device synchronisation inside the loop)?
    python scripts/host_time.py            eager step (every launch from Python + ctypes) and the graph-replayed step (graph_step.py)
    DTYPE=bf16 python scripts/host_time.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import synth
from object_detection_torch2_amd.graph_step import GraphedTrainStep
from object_detection_torch2_amd.model import SSD
from object_detection_torch2_amd.optim import FusedAdam
dev = 'cuda:0'
net = SSD(21).to(dev); df = net.default_bboxes.to(dev)
net.set_compute_dtype(os.environ.get('DTYPE', 'fp32'))
opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4, on_step=net.hip_weights_changed)
ims = torch.from_numpy(synth.images(1, 32)).to(dev); tgs = torch.from_numpy(synth.targets(2, 32, g_max=8)).to(dev)
torch.cuda.synchronize(); ev = torch.cuda.current_stream().record_event()
def step():
    opt.zero_grad(); out = net(ims, ready_event=ev); loss = net.loss(out, tgs, df); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
parts = dict(fwd=0.0, loss=0.0, bwd=0.0, opt=0.0)
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter(); opt.zero_grad(); out = net(ims, ready_event=ev)
    b = time.perf_counter(); loss = net.loss(out, tgs, df)
    c = time.perf_counter(); loss.backward()
    d = time.perf_counter(); opt.step()
    e = time.perf_counter()
    parts['fwd'] += b - a; parts['loss'] += c - b; parts['bwd'] += d - c; parts['opt'] += e - d
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'eager: host enqueue {1e3*(t1-t0)/20:.2f} ms/step, device done after {1e3*(t2-t0)/20:.2f} ms/step; parts (ms/step): ' + ', '.join(f'{k} {1e3*v/20:.2f}' for k, v in parts.items()))
runner = GraphedTrainStep(net, opt, df, ims.shape, tgs.shape)
runner.feed(ims, ev)
for _ in range(5):
    runner.feed(ims, ev); runner.step(tgs)
torch.cuda.synchronize()
parts = dict(feed=0.0, step=0.0)
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter(); runner.feed(ims, ev)
    b = time.perf_counter(); runner.step(tgs)
    c = time.perf_counter()
    parts['feed'] += b - a; parts['step'] += c - b
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'graph: host enqueue {1e3*(t1-t0)/20:.3f} ms/step, device done after {1e3*(t2-t0)/20:.2f} ms/step; parts (ms/step): ' + ', '.join(f'{k} {1e3*v/20:.3f}' for k, v in parts.items()))
