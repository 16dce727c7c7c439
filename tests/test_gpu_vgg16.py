"""GPU: VGG16 classification pre-training on the HIP path (reference model/vgg16.py, train.py:69-74).
Pinned by G13 (the reference's own VGG16 run on CPU with a local weights file: logits, literal loss, classifier2 gradients) and,
for what the reference cannot run (trainable features + dropout), by the oracle's torch-CPU restatement with the same masks."""
import sys
import pathlib

import numpy as np
import pytest
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import voc_fixture  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _load(net, sd):
    own = net.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if v.dim() == 0 or v.stride()[0] != 0 or v.numel() == 1:          # skip the stride-0 placeholder stack
                own[k].copy_(v)
    return net


def test_vgg16_surface_and_errors():
    from object_detection_torch2_amd.model import VGG16
    net = VGG16(num_classes=20, transfer_learning=True)
    keys = list(net.state_dict().keys())
    assert len(keys) == 103 and keys[0] == 'features.0.weight' and 'classifier.6.bias' in keys and 'classifier2.6.weight' in keys
    assert tuple(net.classifier[0].weight.shape) == (4096, 25088) and tuple(net.classifier2[6].weight.shape) == (20, 4096)
    assert all(not p.requires_grad for p in net.features.parameters()) and len(list(net.train_params())) == 6
    full = VGG16(num_classes=20)
    assert len(list(full.train_params())) == 52 + 6
    assert abs(net.classifier2[0].weight.std().item() - 0.01) < 2e-4 and net.classifier2[3].bias.abs().max().item() == 0
    net.to(DEV)
    with pytest.raises(ValueError, match='184..215'):                       # 300 px leaves a 10x10 map: Linear(512*7*7) cannot take it
        net(torch.zeros(2, 3, 300, 300, device=DEV))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net(torch.zeros(2, 3, 200, 200))


def test_vgg16_transfer_learning_vs_reference_golden(golden):
    """G13: logits, literal loss (vgg16.py:127-129) and classifier2 gradients of the REFERENCE module (Dropout in eval mode)."""
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import synth
    from object_detection_torch2_amd.model import VGG16
    fx = golden('g13_vgg16.npz')
    net = _load(VGG16(num_classes=20, transfer_learning=True), O.vgg16_init_state(seed=5, num_classes=20)).to(DEV)
    net.classifier2[2].p = net.classifier2[5].p = 0.0                         # the fixture was made with the Dropout modules in eval mode
    x = torch.from_numpy(synth.images(77, 2, 200)).to(DEV)
    tgt = torch.zeros(2, 20, device=DEV); tgt[0, 3] = 1; tgt[1, 11] = 1
    out = net(x)
    assert tuple(out.shape) == (2, 20)
    assert np.abs(out.detach().cpu().numpy() - fx['logits']).max() < 2e-3 * max(1.0, np.abs(fx['logits']).max())
    loss = net.loss(out, tgt)
    assert abs(loss.item() - float(fx['loss'])) < 2e-3 and loss.item() < 0      # literal: a log-likelihood, negative
    loss.backward()
    for i in (0, 3, 6):
        gw = net.classifier2[i].weight.grad.cpu().numpy().reshape(-1)[::(997 if i < 6 else 7)]
        gb = net.classifier2[i].bias.grad.cpu().numpy()
        assert np.linalg.norm(gw - fx[f'gw_{i}']) <= 2e-2 * np.linalg.norm(fx[f'gw_{i}']) + 1e-9, i
        assert np.linalg.norm(gb - fx[f'gb_{i}']) <= 2e-2 * np.linalg.norm(fx[f'gb_{i}']) + 1e-9, i
    assert all(p.grad is None for p in net.features.parameters())
    assert np.allclose(net.features[1].running_mean.cpu().numpy(), fx['bn0_running_mean'], rtol=2e-4, atol=2e-6)
    assert int(fx['ref_has_train_params']) == 0 and hasattr(net, 'train_params')   # the reference's train.py:97 cannot reach its own model
    assert abs(net.loss(out.detach(), tgt, nll=True).item() + loss.item()) < 1e-6


def test_vgg16_full_training_step_vs_oracle():
    """Everything trainable, 1000-way stack, dropout ON with the counter-hash masks the oracle regenerates: logits, NLL loss, every
    gradient (features through all five pools), then one Adam step."""
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import synth
    from object_detection_torch2_amd.model import VGG16
    from object_detection_torch2_amd.optim import FusedAdam
    sd = O.vgg16_init_state(seed=6, num_classes=20, head='classifier')
    net = _load(VGG16(num_classes=20), sd).to(DEV)
    net.dropout_seed = 9
    x = synth.images(78, 3, 208)
    tgt = torch.zeros(3, 1000); tgt[0, 5] = 1; tgt[1, 999] = 1; tgt[2, 123] = 1
    masks = [torch.from_numpy(O.dropout_mask(9, k, 3 * 4096).reshape(3, 4096)) for k in (0, 1)]
    sdo = {k: v.clone() for k, v in sd.items() if not k.startswith('classifier2.')}
    names = [k for k in sdo if sdo[k].is_floating_point() and not k.endswith('running_mean') and not k.endswith('running_var')]
    for k in names:
        sdo[k].requires_grad_(True)
    out_o = O.vgg16_forward(sdo, torch.from_numpy(x), 'classifier', training=True, masks=masks)
    loss_o = -O.vgg16_loss(out_o, tgt)
    loss_o.backward()
    opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4)
    opt.zero_grad()
    out = net(torch.from_numpy(x).to(DEV))
    ref = out_o.detach().numpy()
    assert np.abs(out.detach().cpu().numpy() - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())
    loss = net.loss(out, tgt.to(DEV), nll=True)
    assert abs(loss.item() - loss_o.item()) < 2e-3 and loss.item() > 0
    loss.backward()
    got = dict(net.named_parameters())
    for k in names:
        if k.startswith('features.') and k.endswith('.bias') and sdo[k].dim() == 1 and int(k.split('.')[1]) in (0, 3, 7, 10, 14, 17, 20, 24, 27, 30, 34, 37, 40):
            continue                       # conv bias in front of a train-mode BatchNorm: analytically zero (rounding noise on both sides)
        g, r = got[k].grad.cpu().double(), sdo[k].grad.double()
        rel = ((g - r).norm() / max(r.norm().item(), 1e-30)).item()
        assert rel < (1e-3 if k.startswith('classifier') else 5e-2), (k, rel)
    before = net.classifier[6].weight.detach().clone()
    opt.step()
    assert not torch.equal(before, net.classifier[6].weight)
    # the second forward draws new masks (streams 2, 3), so the logits differ from a replay with the first masks
    with torch.no_grad():
        a = net(torch.from_numpy(x).to(DEV)); b = net(torch.from_numpy(x).to(DEV))
    assert not torch.equal(a, b)
    net.eval()
    with torch.no_grad():
        assert torch.equal(net(torch.from_numpy(x).to(DEV)), net(torch.from_numpy(x).to(DEV)))   # eval: no dropout, running stats


def test_train_entry_point_classification(tmp_path, monkeypatch, capsys):
    """train.py --purpose classification on the tiny VOC tree (one crop per annotated object, train.py:69-74): one epoch, weights
    saved in the reference's layout, and SSD picks them up as its frozen backbone (ssd.py:25, vgg16.py:64-66)."""
    from object_detection_torch2_amd import train
    from object_detection_torch2_amd.model import SSD
    root = voc_fixture.make_tree(tmp_path / 'voc')
    monkeypatch.setattr(sys, 'argv', ['train', '--purpose', 'classification', '--imsize', '200', '--data_dirs', str(root), '--batch_size', '3',
                                      '--num_workers', '0', '--epochs', '1', '--result_dir', str(tmp_path / 'result')])
    train.main()
    out = capsys.readouterr().out
    assert 'Finished Training' in out and '[Epoch 1/1] loss:' in out
    wp = tmp_path / 'result' / 'classification' / 'weights.pth'
    sd = torch.load(wp, map_location='cpu', weights_only=True)
    assert len(sd) == 103 and all(torch.isfinite(v.float()).all() for v in sd.values())
    ssd = SSD(num_classes=21, weights_path_vgg16=wp)
    assert torch.equal(ssd.features['conv_3_2'].weight, sd['features.17.weight'])
    assert torch.equal(ssd.features['bn_5_3'].running_var, sd['features.41.running_var'])
