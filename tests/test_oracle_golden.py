"""CPU: the oracle (oracle/ssd_oracle.py) against the golden vectors generated from the imported reference
(tests/golden/make_golden.py).  Inputs are regenerated from the counter-hash streams."""
import numpy as np
import torch

from conftest import unpack
from oracle import ssd_oracle as O
from object_detection_torch2_amd import synth

LOSS_CASES = {
    'A': lambda: (synth.targets(11, 4, g_max=8), synth.logits(12, 4)),
    'B': lambda: (synth.targets(13, 2, g_max=16, wh_lo=0.3, wh_hi=0.6, g_fixed=16), synth.logits(14, 2)),
    'C': lambda: (_case_c(), synth.logits(16, 3)),
}


def _case_c():
    tg = synth.targets(15, 3, g_max=5)
    tg[1] = 0
    tg[1, 0, :4] = (0.5, 0.5, 0.004, 0.004)
    tg[1, 0, 4 + 7] = 1
    return tg


def test_default_boxes_bit_exact(golden):
    g = golden('g1_default_bboxes.npz')['default_bboxes']
    assert g.shape == (8732, 4)
    assert np.array_equal(O.default_bboxes(), g)


def test_loss_cases(golden):
    fx = golden('g2_loss.npz')
    df = O.default_bboxes()
    for name, make in LOSS_CASES.items():
        tg, out = make()
        t = O.loss_terms(out, tg, df)
        assert np.array_equal(t['is_match'], unpack(fx[name + '_match_bits'], fx[name + '_match_shape'])), name
        assert np.array_equal(t['pos_num'], fx[name + '_pos_num']) and np.array_equal(t['neg_num'], fx[name + '_neg_num'])
        for k in ('l_loc', 'l_conf_pos', 'l_conf_neg'):
            assert np.abs(t[k] - fx[f'{name}_{k}']).max() < 2e-5, (name, k)
        assert np.array_equal(t['pos_valid'], unpack(fx[name + '_pos_valid_bits'], t['pos_valid'].shape))
        assert np.array_equal(t['neg_valid'], unpack(fx[name + '_neg_valid_bits'], t['neg_valid'].shape))
        assert abs(float(t['loss']) - float(fx[name + '_loss'])) < 1e-4
        g = O.loss_grad(out, tg, df)
        assert np.abs(g[:, ::37, :] - fx[name + '_grad_rows']).max() < 1e-6
        assert np.allclose(g.sum(axis=(1, 2), dtype=np.float64), fx[name + '_grad_sum'], atol=1e-6)
    # branch coverage the reference's own code has: neg//3 split (B) and pos_num == 0 (C image 1)
    assert (fx['B_pos_num'] * 3 < fx['B_neg_num']).all() and fx['C_pos_num'][1] == 0


def test_loss_headline_shape(golden):
    fx = golden('g2_loss.npz')
    tg, out = synth.targets(17, 32, g_max=8), synth.logits(18, 32)
    t = O.loss_terms(out, tg, O.default_bboxes())
    assert np.array_equal(t['pos_num'], fx['D_pos_num'])
    assert abs(float(t['loss']) - float(fx['D_loss'])) < 1e-4


def test_postprocess(golden):
    fx = golden('g6_post.npz')
    df = O.default_bboxes()
    for tag, bias, seed in (('biased', 3.0, 21), ('unbiased', 0.0, 22)):
        raw = synth.logits(seed, 2, void_bias=bias, loc_scale=0.3)
        dec, sc = O.calc_coordicate(raw, df), O.calc_score(raw)
        assert np.abs(dec[:, ::29] - fx[tag + '_decode_rows']).max() < 1e-6
        assert np.abs(sc[:, ::29] - fx[tag + '_score_rows']).max() < 1e-6
        assert np.array_equal(sc.argmax(axis=2), fx[tag + '_score_argmax'])
        o = np.concatenate([dec, sc], axis=2)
        keep = np.stack([np.isin(np.arange(8732), O.nms_keep(o[n])) for n in range(2)])
        assert np.array_equal(keep, unpack(fx[tag + '_keep_bits'], keep.shape))
    t = synth.targets(23, 2, g_max=6)
    pr = np.concatenate([np.clip(synth.logits(24, 2)[:, :500, :4] * 0.2 + 0.5, 0.01, None), np.zeros((2, 500, 21), np.float32)], 2)
    assert np.array_equal(O.calc_iou(pr, t), fx['iou_pr_gt'])


def test_forward_and_stats(golden):
    fx = golden('g8_forward.npz')
    assert tuple(fx['literal_shape']) == (2, 0, 25)          # the literal forward never fires a detector
    assert int(fx['n_state']) == 173 and int(fx['n_train']) == 52 and int(fx['n_train_elems']) == 11578222
    sd = O.init_state(seed=1)
    assert len(sd) == 173
    torch.set_num_threads(8)
    x = torch.from_numpy(synth.images(31, 2))
    with torch.no_grad():
        y = O.forward(sd, x, training=True).numpy()
    assert y.shape == (2, 8732, 25)
    assert np.abs(y[:, ::7] - fx['fwd_train_rows']).max() < 1e-4
    for k in sd:
        if k.endswith('running_mean') or k.endswith('running_var'):
            assert np.allclose(sd[k].numpy(), fx['stat_' + k], rtol=1e-5, atol=1e-6), k
    with torch.no_grad():
        ye = O.forward(sd, x, training=False).numpy()
    assert np.abs(ye[:, ::7] - fx['fwd_eval_rows']).max() < 1e-4


def test_vgg16_oracle_matches_reference_golden(golden):
    """G13: the oracle's VGG16 restatement (model/vgg16.py:85-130) against the reference's own module run on CPU at generation time
    (logits and literal loss bit-equal there; here the committed samples of the logits, loss and classifier2 gradients)."""
    import torch
    from object_detection_torch2_amd import synth
    fx = golden('g13_vgg16.npz')
    sd = {k: v.clone() for k, v in O.vgg16_init_state(seed=5, num_classes=20).items() if not k.startswith('classifier.')}
    for k in sd:
        if k.startswith('classifier2.'):
            sd[k].requires_grad_(True)
    x = torch.from_numpy(synth.images(77, 2, 200))
    tgt = torch.zeros(2, 20); tgt[0, 3] = 1; tgt[1, 11] = 1
    torch.set_num_threads(8)
    out = O.vgg16_forward(sd, x, 'classifier2', training=True, masks=None)
    loss = O.vgg16_loss(out, tgt)
    loss.backward()
    assert np.allclose(out.detach().numpy(), fx['logits'], rtol=1e-5, atol=1e-6) and abs(loss.item() - float(fx['loss'])) < 1e-6
    for i in (0, 3, 6):
        assert np.allclose(sd[f'classifier2.{i}.weight'].grad.numpy().reshape(-1)[::(997 if i < 6 else 7)], fx[f'gw_{i}'], rtol=1e-4, atol=1e-9)
        assert np.allclose(sd[f'classifier2.{i}.bias'].grad.numpy(), fx[f'gb_{i}'], rtol=1e-4, atol=1e-9)
    assert np.allclose(sd['features.1.running_mean'].numpy(), fx['bn0_running_mean'], rtol=1e-5, atol=1e-8)
    m = O.dropout_mask(9, 0, 1000)
    assert set(np.unique(m).tolist()) == {0.0, 2.0} and 0.4 < (m > 0).mean() < 0.6


def test_oracle_bf16_restatement_rounds_where_the_design_says():
    """oracle.forward_bf16 (checker of the package's bf16 mode; the reference has no bf16 path): with weights and input chosen so that every
    operand is exactly representable in bf16 the restatement must equal the fp32 forward bit for bit on the first layers, its gradient
    passes through the roundings (straight-through), and the dy handed to a weight gradient is bf16-rounded."""
    import torch
    import torch.nn.functional as F
    torch.manual_seed(0)
    # the two primitives it is built from
    t = torch.randn(1000) * 3
    r = t.bfloat16().float()
    x = t.clone().requires_grad_(True)
    y = x + (x.detach().bfloat16().float() - x.detach())
    assert torch.equal(y.detach(), r)
    y.sum().backward()
    assert torch.equal(x.grad, torch.ones_like(x))                  # straight-through
    # end to end at a tiny batch: close to the fp32 forward at bf16 level, finite gradients for every trainable tensor
    from object_detection_torch2_amd import synth
    ims = torch.from_numpy(synth.images(7, 2))
    sd32, sd16 = O.init_state(1), O.init_state(1)
    names = O.trainable_param_names()
    for k in names:
        sd16[k].requires_grad_(True)
    with torch.no_grad():
        a = O.forward(sd32, ims, training=True)
    b = O.forward_bf16(sd16, ims)
    rel = ((a - b.detach()).norm() / a.norm()).item()
    assert 0.01 < rel < 0.2, rel                                    # bf16 operands: percent-level, not fp32-level, not garbage
    for k in ('features.bn_4_3.running_mean', 'features.bn_11_2.running_var'):
        assert torch.allclose(sd16[k], sd32[k], rtol=0.2, atol=0.05) and int(sd16[k.rsplit('.', 1)[0] + '.num_batches_tracked']) == 1
    b.square().mean().backward()
    assert all(sd16[k].grad is not None and torch.isfinite(sd16[k].grad).all() for k in names)
    assert sd16['features.conv_6_1.bias'].grad.abs().max() < 1e-6   # a conv bias in front of a train-mode BatchNorm: analytically zero
