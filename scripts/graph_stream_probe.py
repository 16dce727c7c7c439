"""Which stream should the tail graphs be launched on?  Replicates bench.py's stream creation order, then times runner steps on the
caller's stream and on freshly created streams of both priorities (torch hands out pool streams round-robin; the runtime maps HIP
streams to a few hardware queues)."""
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
main_stream = torch.cuda.Stream(device=dev, priority=-1)
torch.cuda.set_stream(main_stream)
def estep():
    opt.zero_grad(); out = net(ims, ready_event=ev); loss = net.loss(out, tgs, df); loss.backward(); opt.step()
for _ in range(3): estep()
torch.cuda.synchronize()
r = GraphedTrainStep(net, opt, df, ims.shape, tgs.shape)
K = 12
def timed(label, stream):
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        r.feed(ims, ev)
        for _ in range(3):
            r.feed(ims, ev); r.step(tgs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            r.feed(ims, ev); r.step(tgs)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r.step(tgs); r.drain()
    torch.cuda.synchronize()
    print(f'{label} {stream.cuda_stream:#x} prio {stream.priority}: {1e3 * (t1 - t0) / K:.3f} ms per step', flush=True)
timed('caller main_stream', main_stream)
for i in range(int(os.environ.get('PROBE_N', '6'))):
    timed(f'fresh high-priority stream #{i}', torch.cuda.Stream(device=dev, priority=-1))
timed('caller main_stream again', main_stream)
