"""bf16 compute mode vs fp32 mode on the same weights/batch: output error statistics per detector head."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ssd_oracle as O
from object_detection_torch2_amd import synth
from object_detection_torch2_amd.model import SSD
DEV = 'cuda:0'
for N in (2, 8, 32):
    net = SSD(21); net.load_state_dict(O.init_state(1)); net.to(DEV)
    x = torch.from_numpy(synth.images(31, N)).to(DEV)
    with torch.no_grad():
        y32 = net(x).cpu().numpy(); net.load_state_dict(O.init_state(1))
        y16 = net.set_compute_dtype('bf16')(x).cpu().numpy()
    e = np.abs(y16 - y32)
    offs = [0, 5776, 7942, 8542, 8692, 8728, 8732]
    print(f'N={N} all: max {e.max():.3f} mean {e.mean():.4f} relF {np.linalg.norm(y16 - y32) / np.linalg.norm(y32):.4f} |y| mean {np.abs(y32).mean():.3f}')
    for k in range(6):
        s = slice(offs[k], offs[k + 1])
        print(f'   head {k}: max {e[:, s].max():.3f} mean {e[:, s].mean():.4f} relF {np.linalg.norm((y16 - y32)[:, s]) / np.linalg.norm(y32[:, s]):.4f}')
