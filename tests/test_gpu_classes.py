"""GPU: SSD(num_classes) is parametric in the reference (ssd.py:9,70-77; utils.py:52-55).  Forward, MultiBox loss, its gradient,
decode / score and NMS for class counts other than VOC's 21, against the oracle (whose loss / post-process code is shape-generic
and pinned at C = 21 by the golden fixtures)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('C', [5, 30])
def test_loss_grad_decode_nms_other_class_counts(C):
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import synth, utils
    from object_detection_torch2_amd.model import SSD
    N, P = 3, 8732
    df = O.default_bboxes()
    out = synth.logits(11 + C, N, P, C)
    tg = synth.targets(12 + C, N, g_max=6, num_classes=C)
    net = SSD(num_classes=C)
    o = torch.from_numpy(out).to(DEV).requires_grad_(True)
    loss = net.loss(o, torch.from_numpy(tg).to(DEV), torch.from_numpy(df).to(DEV))
    loss.backward()
    ref = float(O.loss(out, tg, df))
    assert abs(loss.item() - ref) < 1e-4, (loss.item(), ref)                 # north-star bound
    g_ref = O.loss_grad(out, tg, df)
    assert np.abs(o.grad.cpu().numpy() - g_ref).max() < 1e-6
    # decode + score + NMS
    biased = out.copy(); biased[:, :, 4] += 2.0
    got = torch.from_numpy(biased).to(DEV)
    utils.decode_score_(got, torch.from_numpy(df).to(DEV))
    want = biased.copy()
    want[:, :, :4] = O.calc_coordicate(biased, df)
    want[:, :, 4:] = O.calc_score(biased)
    g = got.cpu().numpy()
    assert np.abs(g[:, :, :4] - want[:, :, :4]).max() < 1e-5
    assert np.array_equal(g[:, :, 4:] != 0, want[:, :, 4:] != 0)            # arg-max class exact
    assert np.abs(g[:, :, 4:] - want[:, :, 4:]).max() < 1e-6
    # NMS on identical boxes / scores: identical keep set
    t = torch.from_numpy(want.copy()).to(DEV)
    utils.non_maximum_suppression(t)
    ref_nms = O.non_maximum_suppression(want.copy())
    assert np.array_equal(t.cpu().numpy()[:, :, 4:] != 0, ref_nms[:, :, 4:] != 0)


def test_forward_and_train_step_with_five_classes():
    """End to end with k = 5: head widths A * (k + 4), zero-copy concat at the prior offsets, loss, backward, Adam."""
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import synth
    from object_detection_torch2_amd.model import SSD
    from object_detection_torch2_amd.optim import FusedAdam
    C = 5
    sd = O.init_state(seed=3, num_classes=C)
    net = SSD(num_classes=C)
    net.load_state_dict(sd)
    net.to(DEV)
    ims = synth.images(70, 2)
    tg = synth.targets(71, 2, g_max=5, num_classes=C)
    df = O.default_bboxes()
    names = O.trainable_param_names(C)
    for k in names:
        sd[k].requires_grad_(True)
    out_o = O.forward(sd, torch.from_numpy(ims), num_classes=C, training=True)
    o = out_o.detach().numpy()
    out_o.backward(torch.from_numpy(O.loss_grad(o, tg, df)))
    opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4)
    opt.zero_grad()
    out = net(torch.from_numpy(ims).to(DEV))
    assert out.shape == (2, 8732, 4 + C)
    err = np.abs(out.detach().cpu().numpy() - o)
    assert err[:, :8692].max() < 1e-3             # measured 3.3e-4 (F(2x2) everywhere) / 4.4e-4 (default: conv_3_2 .. conv_5_3 on F(4x4))
    # the 3x3 and 1x1 maps (last 40 priors) of a batch of TWO: BatchNorm over 18 and over 2 samples per channel turns rounding differences
    # of the backbone into output differences an order above those of the large maps; measured 1.93e-3 with every backbone layer on
    # the F(2x2) kernels (SSD_WINOGRAD=f2), 2.04e-3 with conv_3_2 .. conv_5_3 on F(4x4) (at batch 32: 3.7e-4 / 3.9e-4 over all priors,
    # profiles/r03_d_full_batch_parity.txt)
    assert err[:, 8692:].max() < 4e-3             # the same network at batch 32 holds 1e-3 on every prior: tests/test_gpu_configs.py
    loss = net.loss(out, torch.from_numpy(tg).to(DEV), torch.from_numpy(df).to(DEV))
    assert abs(loss.item() - float(O.loss(o, tg, df))) < 1e-4      # north-star bound (measured 3.8e-6: tests/tools/bounds_probe.py)
    loss.backward()
    got = dict(net.named_train_params())
    for k in names:
        if k.startswith('detectors.'):
            g, r = got[k].grad.cpu().double(), sd[k].grad.double()
            # det_10_2 / det_11_2 sit on the 3x3 / 1x1 maps of two images: one ReLU unit of conv_11_2 within fp32 rounding of 0 that
            # switches with the summation order of a build moves them by 1e-3 (measured 1.2e-3 on det_11_2); the large maps stay at 1e-3
            assert ((g - r).norm() / r.norm()).item() < (3e-3 if k.split('.')[1] in ('det_10_2', 'det_11_2') else 1e-3), k
    before = net.detectors['det_7_1'].weight.detach().clone()
    opt.step()
    assert not torch.equal(before, net.detectors['det_7_1'].weight)
