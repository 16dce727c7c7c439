"""Full fine-tuning (extension: the reference keeps VGG frozen, ssd.py:31-32): max-pool backward kernel
against torch autograd, and every gradient of the un-frozen network against autograd through the oracle forward."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('N,H,W,C,pad', [(2, 8, 8, 8, 0), (2, 75, 75, 32, 1), (1, 7, 9, 4, 0), (3, 38, 38, 64, 0), (1, 5, 5, 4, 1)])
def test_pool_backward_matches_autograd(N, H, W, C, pad):
    from object_detection_torch2_amd import _lib, synth
    z = torch.from_numpy(synth.pseudo_normal(5, 1, N * H * W * C).reshape(N, H, W, C).astype(np.float32))
    scale = torch.from_numpy((0.5 + synth.uniform(6, 1, C)).astype(np.float32))
    shift = torch.from_numpy((synth.uniform(7, 1, C) - 0.5).astype(np.float32))
    a = torch.relu(z * scale + shift).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    pooled = torch.nn.functional.max_pool2d(a, 2, 2, padding=pad)
    g = torch.from_numpy(synth.pseudo_normal(8, 1, pooled.numel()).reshape(pooled.shape).astype(np.float32))
    pooled.backward(g)
    want = a.grad.permute(0, 2, 3, 1).contiguous().numpy()
    da = torch.full((N, H, W, C), 7.0, device=DEV)                          # every element must be overwritten
    gd = g.permute(0, 2, 3, 1).contiguous().to(DEV)
    zd, scd, shd = z.to(DEV), scale.to(DEV), shift.to(DEV)                  # keep the device copies alive across the launch
    _lib.check(_lib.load().ssdhip_bn_relu_pool_bwd(_lib.ptr(zd), _lib.ptr(scd), _lib.ptr(shd), _lib.ptr(gd),
                                                   _lib.ptr(da), N, H, W, C, pad, _lib.stream()), 'pool_bwd')
    got = da.cpu().numpy()
    # identical argmax decisions (ties only among the ReLU zeros, where both sides take the window's first element)
    assert np.array_equal(got, want)


def test_full_finetune_gradients_vs_oracle_autograd():
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import synth
    from object_detection_torch2_amd.model import SSD
    net = SSD(num_classes=21)
    net.load_state_dict(O.init_state(seed=1))
    net.to(DEV).finetune_backbone()
    names = [k for k, _ in net.named_train_params()]
    assert len(names) == 13 * 4 + 10 * 4 + 12 and names[0] == 'features.conv_1_1.weight' and names[-1] == 'detectors.det_11_2.bias'
    sd = O.init_state(seed=1)
    ims, tg, df = synth.images(40, 2), synth.targets(50, 2, g_max=8), O.default_bboxes()
    for k in names:
        sd[k].requires_grad_(True)
    out_o = O.forward(sd, torch.from_numpy(ims), training=True)
    o = out_o.detach().numpy()
    out_o.backward(torch.from_numpy(O.loss_grad(o, tg, df)))
    out = net(torch.from_numpy(ims).to(DEV))
    loss = net.loss(out, torch.from_numpy(tg).to(DEV), torch.from_numpy(df).to(DEV))
    loss.backward()
    got = dict(net.named_train_params())
    worst = {}
    for k in names:
        if 'features.conv' in k and k.endswith('.bias'):
            continue                                        # analytically zero (train-mode BN cancels the conv bias)
        g, r = got[k].grad.cpu().double(), sd[k].grad.double()
        assert g.shape == r.shape, k
        rel = ((g - r).norm() / r.norm()).item()
        cos = (g * r).sum().item() / (g.norm().item() * r.norm().item())
        worst[k] = (rel, cos)
        # same conditioning as the frozen-backbone test (ReLU units within rounding of 0 may switch between two fp32
        # implementations; 36 more layers of it here): layer gradients in L2 / direction, the kernels carry the tight bounds
        tol = 1e-3 if k.startswith('detectors.') else 1e-1
        assert rel < tol and cos > 0.995, (k, rel, cos)
    # one Adam step over all 104 tensors, then a frozen-again module must rebuild its plan
    from object_detection_torch2_amd.optim import FusedAdam
    opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4, on_step=net.hip_weights_changed)
    before = net.features['conv_1_1'].weight.detach().clone()
    opt.step()
    assert not torch.equal(before, net.features['conv_1_1'].weight)
    net.finetune_backbone(False)
    assert [k for k, _ in net.named_train_params()] == O.trainable_param_names()
    out2 = net(torch.from_numpy(ims).to(DEV))
    net.loss(out2, torch.from_numpy(tg).to(DEV), torch.from_numpy(df).to(DEV)).backward()
    assert net.features['conv_1_1'].weight.requires_grad is False
