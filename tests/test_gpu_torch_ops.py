"""GPU: torch.ops.ssdhip.* (TORCH_LIBRARY wrappers of csrc/torch_ops.cpp) against torch's own operators and against
the ctypes binding of the same C entry points: same launcher, so the results must be bit-identical to the ctypes path."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.fixture(scope='module')
def S():
    from object_detection_torch2_amd import torch_ops
    return torch_ops.load()


def test_conv_bn_pool_ops_match_torch_and_the_ctypes_binding(S):
    from object_detection_torch2_amd import ops
    N, H, W, cin, cout = 2, 19, 19, 64, 96
    x = rnd(N, cin, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=0.1)
    b = rnd(cout, seed=3)
    xh = ops.to_nhwc(x.to(DEV))
    packed, KP, inner = ops.pack_weight(w.to(DEV), 0)
    p2 = S.pack_conv_weight(w.to(DEV), (cout + 127) // 128 * 128, inner, KP, 0)
    assert torch.equal(p2, packed)
    z, st = S.conv2d_nhwc(xh, p2, b.to(DEV), None, None, cout, 3, 1, 1, KP, 0, [H, W], True, False)
    z0, st0 = ops.conv2d_nhwc(xh, w.to(DEV), b.to(DEV), pad=1, want_stats=True)
    assert torch.equal(z, z0) and torch.equal(st, st0)
    ref = F.conv2d(x, w, b, padding=1)
    assert ((z.permute(0, 3, 1, 2).cpu() - ref).abs().max() / ref.abs().max()).item() < 2e-5
    # BatchNorm(train) from the partial table, running statistics updated in place like nn.BatchNorm2d
    gamma, beta = rnd(cout, seed=4) + 1.5, rnd(cout, seed=5)
    rm, rv = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    scale, shift, mean, invstd = S.bn_finalize(st, N * H * W, gamma.to(DEV), beta.to(DEV), rm, rv, 0.1, 1e-5)
    bn = torch.nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
    y_ref = F.max_pool2d(F.relu(bn(ref)), 2, 2, padding=1)
    assert torch.allclose(rm.cpu(), bn.running_mean, rtol=1e-4, atol=1e-5) and torch.allclose(rv.cpu(), bn.running_var, rtol=1e-4, atol=1e-5)
    y = S.bn_relu_pool(z, scale, shift, True, 1)
    assert ((y.permute(0, 3, 1, 2).cpu() - y_ref.detach()).abs().max()).item() < 2e-4
    # weight gradient: direct, F(2x2), F(4x4)
    wg = w.clone().requires_grad_(True)
    dy = rnd(N, cout, H, W, seed=6)
    F.conv2d(x, wg, None, padding=1).backward(dy)
    dyh = ops.to_nhwc(dy.to(DEV), pad_to=(cout + 31) // 32 * 32)
    for mode in (0, 2, 4):
        dw, db = S.conv2d_wgrad(xh, dyh, list(w.shape), 1, 1, None, None, True, mode, False)
        assert ((dw.cpu() - wg.grad).abs().max() / wg.grad.abs().max()).item() < 5e-5, mode
        assert ((db.cpu() - dy.sum((0, 2, 3))).abs().max() / dy.sum((0, 2, 3)).abs().max()).item() < 2e-5
    with pytest.raises(ValueError):
        S.conv2d_wgrad(xh, dyh, list(w.shape), 1, 1, None, None, True, 3, False)
    with pytest.raises(ValueError):
        S.conv2d_nhwc(xh[..., :48].contiguous(), p2, None, None, None, cout, 3, 1, 1, KP, 0, [H, W], False, False)      # Cin not a multiple of 32
    with pytest.raises(RuntimeError, match='GPU'):
        S.bn_relu_pool(z.cpu(), scale, shift, False, 0)


def test_loss_decode_nms_ops_match_the_oracle(S):
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import synth
    from object_detection_torch2_amd.model import topology as T
    N, P, C = 3, 8732, 21
    df = O.default_bboxes()
    maps = [int(v) for m in T.SSD300.box_maps for v in m]
    got_df = S.default_boxes(maps, float(T.SSD300.s_min), float(T.SSD300.s_max), torch.device(DEV))
    assert np.array_equal(got_df.cpu().numpy(), df)                                  # bit-exact (G1)
    out = synth.logits(21, N, P, C)
    tg = synth.targets(22, N, g_max=6)
    o, t, d = torch.from_numpy(out).to(DEV), torch.from_numpy(tg).to(DEV), torch.from_numpy(df).to(DEV)
    loss, terms, sel = S.multibox_loss_fwd(o, t, d, 1.0, 0.25)
    assert abs(loss.item() - float(O.loss(out, tg, df))) < 1e-4
    grad = S.multibox_loss_bwd(o, t, d, 1.0, 0.25, terms, sel, torch.ones(1, device=DEV))
    assert np.abs(grad.cpu().numpy() - O.loss_grad(out, tg, df)).max() < 1e-6
    biased = out.copy(); biased[:, :, 4] += 2.0
    got = torch.from_numpy(biased).to(DEV)
    assert S.decode_score_(got, d) is None
    want = biased.copy()
    want[:, :, :4] = O.calc_coordicate(biased, df)
    want[:, :, 4:] = O.calc_score(biased)
    g = got.cpu().numpy()
    assert np.abs(g[:, :, :4] - want[:, :, :4]).max() < 1e-5 and np.array_equal(g[:, :, 4:] != 0, want[:, :, 4:] != 0)
    tt = torch.from_numpy(want.copy()).to(DEV)
    ws = torch.empty(S.nms_ws_bytes(N, P), dtype=torch.uint8, device=DEV)
    keep = S.nms_(tt, 0.5, ws, False)
    ref = O.non_maximum_suppression(want.copy())
    assert np.array_equal(tt.cpu().numpy()[:, :, 4:] != 0, ref[:, :, 4:] != 0)
    t2 = torch.from_numpy(want.copy()).to(DEV)
    keep2 = S.nms_(t2, 0.5, None, False)                                              # all-LDS variant: same keep set
    assert torch.equal(keep, keep2) and torch.equal(tt, t2)
    iou = S.calc_iou(tt[:, :64].contiguous(), tt[:, 64:200].contiguous())
    with pytest.raises(ValueError, match='contiguous'):
        S.calc_iou(tt[:, :64], tt[:, 64:200])
    assert np.array_equal(iou.cpu().numpy(), O.calc_iou(tt[:, :64].cpu().numpy(), tt[:, 64:200].cpu().numpy()))     # bit-exact (G6)
    with pytest.raises(ValueError):
        S.multibox_loss_fwd(o, t[:, :, :20].contiguous(), d, 1.0, 0.25)
    with pytest.raises(ValueError):
        S.nms_(tt, 0.5, ws[:100], False)


def test_normalize_and_adam_ops(S):
    from object_detection_torch2_amd.optim import FusedAdam
    x = rnd(2, 3, 30, 30, seed=9).abs()
    y = S.normalize_nchw_to_nhwc4(x.to(DEV))
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    assert torch.allclose(y[..., :3].permute(0, 3, 1, 2).cpu(), (x - mean) / std, atol=1e-6) and (y[..., 3] == 0).all()
    # the fused Adam kernels through the operator, on the plan FusedAdam builds, against torch.optim.Adam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(1000, device=DEV)), torch.nn.Parameter(torch.randn(33, 7, device=DEV))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt, ref = FusedAdam(ps, lr=1e-2, weight_decay=5e-4), torch.optim.Adam(qs, lr=1e-2, weight_decay=5e-4)
    for k in range(3):
        for p, q in zip(ps, qs):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        work = opt.prepare()
        for plan, _ in work:
            S.adam_multi_tensor_(plan['table'], plan['sizes'], plan['blockmap'], plan['n_blocks'], 8192, plan['step'], plan['hyper'])
        opt.committed(work)
        ref.step()
    for p, q in zip(ps, qs):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
