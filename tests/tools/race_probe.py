"""Stress for cross-stream ordering: many train steps without host synchronisation, several fresh modules in one process;
reports non-finite state and mismatches against the single-stream run (bit-identical results are expected)."""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ssd_oracle as O
from object_detection_torch2_amd import synth
from object_detection_torch2_amd.model import SSD
from object_detection_torch2_amd.optim import FusedAdam
DEV = 'cuda:0'
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 25
def run(single, sync_each, overlap=False):
    net = SSD(21); net.load_state_dict(O.init_state(1)); net.to(DEV)
    net.set_compute_dtype(os.environ.get('DTYPE', 'fp32'))
    net._engine().single_stream = single
    opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4, on_step=net.hip_weights_changed)
    df = net.default_bboxes.to(DEV)
    # different images every step: a backbone pass that overwrote a buffer too early would change the result
    data = [(torch.from_numpy(synth.images(90 + k, N)).to(DEV), torch.from_numpy(synth.targets(190 + k, N, g_max=8)).to(DEV)) for k in range(5)]
    torch.cuda.synchronize(); ev = torch.cuda.current_stream().record_event() if overlap else None
    losses = []
    for it in range(STEPS):
        ims, tgs = data[it % 5]
        opt.zero_grad(); loss = net.loss(net(ims, ready_event=ev), tgs, df); loss.backward(); opt.step()
        losses.append(loss.item() if sync_each else loss)
    ls = [l if isinstance(l, float) else l.item() for l in losses]
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    bad = [k for k, v in sd.items() if not torch.isfinite(v.float()).all()]
    return ls, sd, bad
ref = run(True, True)
print('single-stream: last loss', ref[0][-1], 'bad', len(ref[2]), flush=True)
for rep in range(6):
    ls, sd, bad = run(False, rep % 2 == 0, overlap=rep >= 2)
    diff = [k for k in sd if not torch.equal(sd[k], ref[1][k])]
    first = next((i for i, (a, b) in enumerate(zip(ls, ref[0])) if a != b), None)
    print(f'rep {rep} sync_each={rep % 2 == 0} overlap={rep >= 2}: losses equal {ls == ref[0]} (first differing step {first}) bad {len(bad)} differing tensors {len(diff)} {diff[:3]}', flush=True)
    gc.collect()
