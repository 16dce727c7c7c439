"""Small-batch inference latency: eager launches vs one HIP-graph replay (SSD.graphed_forward)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_torch2_amd import synth
from object_detection_torch2_amd.model import SSD
dev = 'cuda:0'
net = SSD(21).to(dev)
for N in (2, 4, 8):
    x = torch.from_numpy(synth.images(1, N)).to(dev)
    run = net.graphed_forward(x)
    def eager():
        with torch.no_grad():
            return net(x)
    for name, fn in (('eager', eager), ('graph', lambda: run(x))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        print(f'batch {N} {name}: {dt*1e3:.3f} ms per forward', flush=True)
    net._engine()._readers.clear()
