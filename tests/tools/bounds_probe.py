"""Measured model-level differences against the oracle at the small batches the parity tests use (what their bounds are set from):
    python tests/tools/bounds_probe.py        (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ssd_oracle as O
from object_detection_torch2_amd import synth
from object_detection_torch2_amd.model import SSD
DEV = 'cuda:0'


def case(tag, imsize, n, C, seed, im_seed, tg_seed, g_max):
    sd = O.init_state(seed=seed, num_classes=C, imsize=imsize)
    net = SSD(num_classes=C, imsize=imsize); net.load_state_dict(sd); net.to(DEV)
    ims, tg, df = synth.images(im_seed, n, imsize), synth.targets(tg_seed, n, g_max=g_max, num_classes=C), O.default_bboxes(imsize)
    with torch.no_grad():
        o = O.forward(sd, torch.from_numpy(ims), num_classes=C, training=True, imsize=imsize).numpy()
    out = net(torch.from_numpy(ims).to(DEV))
    loss = net.loss(out, torch.from_numpy(tg).to(DEV), torch.from_numpy(df).to(DEV))
    err = np.abs(out.detach().cpu().numpy() - o)
    P = df.shape[0]
    print(f'{tag}: max|out-oracle| {err.max():.3e} (first {P - 40}: {err[:, :P - 40].max():.3e}, last 40: {err[:, P - 40:].max():.3e}); '
          f'|loss-oracle| {abs(loss.item() - float(O.loss(o, tg, df))):.3e} (loss {loss.item():.6f})', flush=True)


case('test_gpu_model gradients case (N=2, seeds 40/50)', 300, 2, 21, 1, 40, 50, 8)
case('five classes N=2 (seeds 70/71)', 300, 2, 5, 3, 70, 71, 5)
case('five classes N=32', 300, 32, 5, 3, 70, 71, 5)
case('SSD512 N=2 (seeds 61/62)', 512, 2, 21, 1, 61, 62, 8)
case('SSD512 N=16', 512, 16, 21, 1, 161, 162, 8)
