"""Where does the time of a graph-replayed step go?  The two chains of graph_step.py alone and together, replayed and enqueued eagerly."""
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
def estep():
    opt.zero_grad(); out = net(ims); loss = net.loss(out, tgs, df); loss.backward(); opt.step()
for _ in range(3): estep()
torch.cuda.synchronize()
r = GraphedTrainStep(net, opt, df, ims.shape, tgs.shape)
r._x.copy_(ims); r._targets.copy_(tgs)
eng = net._engine()
bb = r._bb_stream
main = torch.cuda.current_stream() if os.environ.get('PROBE_MAIN') == 'default' else torch.cuda.Stream(device=dev, priority=-1 if os.environ.get('PROBE_MAIN', 'prio') == 'prio' else 0)
K = int(os.environ.get('PROBE_K', '20'))

def timed(label, fn, k=K):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    th = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f'{label}: {1e3 * (t1 - t0) / k:.3f} ms per iteration (host {1e3 * (th - t0) / k:.3f})', flush=True)

def bb_graph():
    with torch.cuda.stream(bb): r._bb_graphs[0].replay()
def t_graph():
    with torch.cuda.stream(main):
        for it in r._tail_items[0]: it.replay()
def both_graph():
    bb_graph(); t_graph()
eng._phased = True
def bb_eager():
    with torch.cuda.stream(bb), torch.no_grad(): r._backbone(0)
def t_eager():
    with torch.cuda.stream(main), torch.no_grad(): r._tail(0, r._work)
def both_eager():
    bb_eager(); t_eager()
ev = torch.cuda.current_stream().record_event()
def runner_step():
    with torch.cuda.stream(main):
        r.feed(ims, ev); r.step(tgs)
if os.environ.get('SSD_GS_DEBUG') is not None:
    with torch.cuda.stream(main):
        r.feed(ims, ev)
    timed('runner feed + step [SSD_GS_DEBUG=%s main=%s K=%d]' % (os.environ.get('SSD_GS_DEBUG'), os.environ.get('PROBE_MAIN', 'prio'), K), runner_step)
    sys.exit(0)
timed('backbone graph alone', bb_graph)
timed('tail graph alone', t_graph)
timed('backbone graph || tail graph (two streams, no events)', both_graph)
timed('backbone eager alone', bb_eager)
timed('tail eager alone (3 streams inside)', t_eager)
timed('backbone eager || tail eager', both_eager)
eng.single_stream = True
timed('tail eager alone, single stream', t_eager)
eng.single_stream = False
eng._phased = False
