"""Generate tests/golden/*.npz by importing the REFERENCE (read-only, /root/reference/src) on CPU.

Run in the build container only:   python tests/golden/make_golden.py
The reference never travels: fixtures hold outputs (data) only; inputs are regenerated on both sides from the
counter-hash streams of object_detection_torch2_amd.synth.  While generating, every oracle function is checked
against the reference result, so a fixture is only written if oracle == reference (within the stated bound).

Never executed: torch.hub.load (vgg16.py:68).  SSD(...) is given a local VGG16 state-dict file so the reference
takes its own local-file branch (vgg16.py:64-66).
"""
import os
import sys
import pathlib
import numpy as np

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, '/root/reference/src')

import torch  # noqa: E402
from model import SSD  # noqa: E402  (reference)
import utils as ref_utils  # noqa: E402  (reference)
from oracle import ssd_oracle as O  # noqa: E402
from object_detection_torch2_amd import synth  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent
torch.set_num_threads(8)


def bare_ssd(num_classes=21):
    s = SSD.__new__(SSD)
    torch.nn.Module.__init__(s)
    s.num_classes = num_classes
    return s


def packbits(a):
    return np.packbits(np.asarray(a, dtype=np.uint8).reshape(-1))


def loss_case(ref, df_t, name, targets, outputs, save_full=True):
    """Run reference loss + autograd, compare with the oracle, return fixture dict."""
    N, P, _ = outputs.shape
    out_t = torch.from_numpy(outputs.copy()).requires_grad_(True)
    tg_t = torch.from_numpy(targets)
    is_match = ref._match(gt=tg_t, df=df_t)
    lval = ref.loss(out_t, tg_t, df_t)
    lval.backward()
    grad = out_t.grad.numpy()
    # reference intermediates (re-derived with the reference's own helpers)
    with torch.no_grad():
        l = out_t[:, :, :4].unsqueeze(2)
        g = ref._calc_delta(gt=tg_t, df=df_t)
        l_loc = (ref._smooth_l1(l - g).sum(dim=3) * is_match).sum(dim=2)
        sp = ref._softmax_cross_entropy(pr=out_t[:, :, 4:], gt=tg_t[:, :, 4:])
        l_conf_pos = (sp * is_match).sum(dim=2)
        gt_void = torch.eye(21)[0].unsqueeze(0).unsqueeze(1)
        sn = ref._softmax_cross_entropy(pr=out_t[:, :, 4:], gt=gt_void)
        l_conf_neg = (sn * (is_match.sum(dim=2, keepdims=True) == 0)).sum(dim=2)
        pos_num = (is_match.sum(dim=2) != 0).sum(dim=1)
        neg_num = P - pos_num
        pos_k, neg_k = ref._split_pos_neg(pos_num, neg_num)
        thr_pos = torch.stack([ref._k_plus_1_th_value(l_conf_pos[i], pos_k[i]) for i in range(N)])
        thr_neg = torch.stack([ref._k_plus_1_th_value(l_conf_neg[i], neg_k[i]) for i in range(N)])
        pos_valid = l_conf_pos > thr_pos.unsqueeze(1)
        neg_valid = l_conf_neg > thr_neg.unsqueeze(1)
    t = O.loss_terms(outputs, targets, df_t.numpy())
    og = O.loss_grad(outputs, targets, df_t.numpy())
    assert np.array_equal(t['is_match'], is_match.numpy()), name + ': match mask differs'
    assert np.array_equal(t['pos_num'], pos_k.numpy()) and np.array_equal(t['neg_num'], neg_k.numpy()), name
    for k, r in (('l_loc', l_loc), ('l_conf_pos', l_conf_pos), ('l_conf_neg', l_conf_neg)):
        err = np.abs(t[k] - r.numpy()).max()
        assert err < 2e-5, (name, k, err)
    assert np.array_equal(t['pos_valid'], pos_valid.numpy()), name + ': pos_valid'
    assert np.array_equal(t['neg_valid'], neg_valid.numpy()), name + ': neg_valid'
    assert abs(float(t['loss']) - float(lval)) < 1e-5 * max(1, abs(float(lval))), (name, t['loss'], float(lval))
    gerr = np.abs(og - grad).max()
    assert gerr < 1e-6, (name, 'grad', gerr)
    print(f'  {name}: loss={float(lval):.6f} pos={pos_k.tolist()[:6]} neg={neg_k.tolist()[:6]} '
          f'branch_neg//3={(pos_num * 3 > neg_num).tolist()[:6]} graderr={gerr:.2e}')
    fx = dict(loss=np.float32(lval.item()), pos_num=pos_k.numpy(), neg_num=neg_k.numpy(),
              match_bits=packbits(is_match.numpy()), match_shape=np.array(is_match.shape),
              thr_pos=thr_pos.numpy(), thr_neg=thr_neg.numpy(),
              pos_valid_bits=packbits(pos_valid.numpy()), neg_valid_bits=packbits(neg_valid.numpy()),
              grad_sum=grad.sum(axis=(1, 2), dtype=np.float64), grad_abs_sum=np.abs(grad).sum(axis=(1, 2), dtype=np.float64),
              grad_rows=grad[:, ::37, :].copy())
    if save_full:
        fx.update(l_loc=l_loc.numpy(), l_conf_pos=l_conf_pos.numpy(), l_conf_neg=l_conf_neg.numpy())
    return fx


def gen_boxes_loss_post():
    ref = bare_ssd()
    df_t = ref._get_default_bboxes()                       # G1
    df = df_t.numpy()
    assert np.array_equal(df, O.default_bboxes()), 'default boxes differ'
    np.savez_compressed(OUT / 'g1_default_bboxes.npz', default_bboxes=df)
    print('G1 default boxes ok', df.shape)

    fx = {}
    # case A: N=4, G<=8 ordinary boxes
    tg = synth.targets(11, 4, g_max=8); out = synth.logits(12, 4)
    fx.update({'A_' + k: v for k, v in loss_case(ref, df_t, 'A', tg, out).items()})
    # case B: N=2, 16 mid-size boxes -> 3*pos > neg branch of _split_pos_neg
    tg = synth.targets(13, 2, g_max=16, wh_lo=0.3, wh_hi=0.6, g_fixed=16); out = synth.logits(14, 2)
    fx.update({'B_' + k: v for k, v in loss_case(ref, df_t, 'B', tg, out).items()})
    # case C: N=3 with an image whose only GT matches nothing (tiny box) -> pos_num = 0 path, plus ragged G
    tg = synth.targets(15, 3, g_max=5)
    tg[1] = 0; tg[1, 0, :4] = (0.5, 0.5, 0.004, 0.004); tg[1, 0, 4 + 7] = 1
    out = synth.logits(16, 3)
    fx.update({'C_' + k: v for k, v in loss_case(ref, df_t, 'C', tg, out).items()})
    # case D: N=32, G=8 headline shape, digests only
    tg = synth.targets(17, 32, g_max=8); out = synth.logits(18, 32)
    fx.update({'D_' + k: v for k, v in loss_case(ref, df_t, 'D', tg, out, save_full=False).items()})
    np.savez_compressed(OUT / 'g2_loss.npz', **fx)
    print('G2-G5 loss fixtures ok')

    # G6 decode / score, G7 NMS keep sets
    fx = {}
    for tag, bias, seed in (('biased', 3.0, 21), ('unbiased', 0.0, 22)):
        raw = synth.logits(seed, 2, void_bias=bias, loc_scale=0.3)
        o = torch.from_numpy(raw.copy())
        dec = ref_utils.calc_coordicate(pr=o, df=df_t)
        sc = ref_utils.calc_score(pr=o)
        assert np.abs(O.calc_coordicate(raw, df) - dec.numpy()).max() < 1e-6
        assert np.abs(O.calc_score(raw) - sc.numpy()).max() < 1e-6
        assert np.array_equal(O.calc_score(raw) > 0, sc.numpy() > 0)
        o[:, :, :4] = dec
        o[:, :, 4:] = sc
        pre = o.numpy().copy()
        post = ref_utils.non_maximum_suppression(outputs=o).numpy()
        mine = O.non_maximum_suppression(pre.copy())
        keep_ref = [np.nonzero(post[n, :, 5:].max(axis=1) > 0)[0] for n in range(2)]
        keep_mine = [np.sort(O.nms_keep(pre[n])) for n in range(2)]
        for a, b in zip(keep_ref, keep_mine):
            assert np.array_equal(a, b), 'NMS keep set differs'
        assert np.array_equal(mine, post)
        cand = [(pre[n, :, 5:].max(axis=1) > 0).sum() for n in range(2)]
        print(f'  nms {tag}: candidates={cand} kept={[len(k) for k in keep_ref]}')
        fx[tag + '_decode_rows'] = dec.numpy()[:, ::29, :]
        fx[tag + '_score_rows'] = sc.numpy()[:, ::29, :]
        fx[tag + '_score_argmax'] = sc.numpy().argmax(axis=2).astype(np.int8)
        fx[tag + '_keep_bits'] = packbits(np.stack([np.isin(np.arange(8732), k) for k in keep_ref]))
        fx[tag + '_post_sum'] = post[:, :, 4:].sum(axis=(1, 2), dtype=np.float64)
    # calc_iou on its own (evaluate.py:132 use): predictions vs GT
    t = synth.targets(23, 2, g_max=6)
    pr = np.concatenate([np.clip(synth.logits(24, 2)[:, :500, :4] * 0.2 + 0.5, 0.01, None), np.zeros((2, 500, 21), np.float32)], 2)
    iou = ref_utils.calc_iou(torch.from_numpy(pr), torch.from_numpy(t)).numpy()
    assert np.array_equal(iou, O.calc_iou(pr, t)), 'calc_iou differs'
    fx['iou_pr_gt'] = iou
    np.savez_compressed(OUT / 'g6_post.npz', **fx)
    print('G6-G7 post-process fixtures ok')


def vgg16_local_file(tmp):
    """Local VGG16 state-dict so the reference takes vgg16.py:64-66.  The two unused classifier stacks are
    stride-0 views (1 element of storage each), so the file stays tiny."""
    sd = {}
    idx, cin = 0, 3
    for v in O.VGG_CFG:
        if v in ('M', 'M_P'):
            idx += 1
            continue
        sd[f'features.{idx}.weight'] = torch.zeros(v, cin, 3, 3)
        sd[f'features.{idx}.bias'] = torch.zeros(v)
        sd[f'features.{idx + 1}.weight'] = torch.ones(v)
        sd[f'features.{idx + 1}.bias'] = torch.zeros(v)
        sd[f'features.{idx + 1}.running_mean'] = torch.zeros(v)
        sd[f'features.{idx + 1}.running_var'] = torch.ones(v)
        sd[f'features.{idx + 1}.num_batches_tracked'] = torch.tensor(0)
        idx += 3
        cin = v
    for name, last in (('classifier', 1000), ('classifier2', 20)):
        for i, (o, c) in zip((0, 3, 6), ((4096, 25088), (4096, 4096), (last, 4096))):
            sd[f'{name}.{i}.weight'] = torch.zeros(1).expand(o, c)
            sd[f'{name}.{i}.bias'] = torch.zeros(1).expand(o)
    p = pathlib.Path(tmp) / 'vgg16_local.pth'
    torch.save(sd, p)
    return p


def intended_forward(net, x):
    """SSD.forward with the documented contract (ssd.py:93): membership test on the det_ name."""
    n = x.size(0)
    ys = []
    x = net.normalize(x)
    for name, layer in net.features.items():
        x = layer(x)
        dn = name.replace('act', 'det')
        if dn in net.detectors:
            ys.append(net.detectors[dn](x).permute(0, 2, 3, 1).reshape(n, -1, net.num_classes + 4))
    return torch.cat(ys, dim=1)


def gen_forward_train():
    import tempfile
    with tempfile.TemporaryDirectory(dir=str(ROOT / 'gpurun_out') if (ROOT / 'gpurun_out').exists() else None) as tmp:
        net = SSD(num_classes=21, weights_path=None, weights_path_vgg16=vgg16_local_file(tmp))
    n_sd = len(net.state_dict())
    n_train = sum(1 for _ in net.train_params())
    n_train_el = sum(p.numel() for p in net.train_params())
    print('reference SSD built: state_dict entries', n_sd, 'trainable tensors', n_train, 'elems', n_train_el)
    assert [k for k, p in net.named_parameters() if p.requires_grad] == O.trainable_param_names()
    literal = net(torch.zeros(2, 3, 300, 300))   # N=1 raises ValueError at bn_11_2 in train mode (1 value per channel)
    print('literal forward shape (Q1):', tuple(literal.shape))

    sd = O.init_state(seed=1)
    assert list(sd.keys()) == list(net.state_dict().keys()), 'state_dict key order differs'
    net.load_state_dict(sd)
    # load_state_dict copies, so the oracle keeps its own tensors
    fx = dict(literal_shape=np.array(literal.shape), n_state=np.int64(n_sd), n_train=np.int64(n_train),
              n_train_elems=np.int64(n_train_el))

    x = torch.from_numpy(synth.images(31, 2))
    with torch.no_grad():
        y = intended_forward(net, x)                     # train-mode BN (Q3)
        sd_o = {k: v.clone() for k, v in sd.items()}
        yo = O.forward(sd_o, x, training=True)
    err = (y - yo).abs().max().item()
    print('forward train-mode: shape', tuple(y.shape), 'max|ref-oracle| =', err, 'max|y| =', y.abs().max().item())
    assert err < 1e-4
    rsd = net.state_dict()
    for k in rsd:
        if 'running' in k or 'num_batches' in k:
            assert torch.allclose(rsd[k].float(), sd_o[k].float(), rtol=1e-5, atol=1e-6), k
    fx['fwd_train_rows'] = y.numpy()[:, ::7, :].copy()
    fx['fwd_train_sum'] = y.numpy().sum(axis=(1, 2), dtype=np.float64)
    fx['fwd_train_abs_sum'] = np.abs(y.numpy()).sum(axis=(1, 2), dtype=np.float64)
    for k in rsd:
        if k.endswith('running_mean') or k.endswith('running_var'):
            fx['stat_' + k] = rsd[k].numpy().copy()
    # eval-mode BN with the just-updated running stats
    net.eval()
    with torch.no_grad():
        ye = intended_forward(net, x)
        yeo = O.forward(sd_o, x, training=False)
    net.train()
    err = (ye - yeo).abs().max().item()
    print('forward eval-mode: max|ref-oracle| =', err)
    assert err < 1e-4
    fx['fwd_eval_rows'] = ye.numpy()[:, ::7, :].copy()
    np.savez_compressed(OUT / 'g8_forward.npz', **fx)
    print('G8 forward fixtures ok')

    # G9: 3 reference train-loop bodies (train.py:115-124) at N=4, Adam lr 1e-3 wd 5e-4
    sd = O.init_state(seed=1)
    net.load_state_dict(sd)
    sd_o = {k: v.clone() for k, v in sd.items()}
    opt = torch.optim.Adam(net.train_params(), lr=1e-3, weight_decay=5e-4)
    df_t = net.default_bboxes
    df = df_t.numpy()
    opt_state, losses_ref, losses_o = {}, [], []
    for step in range(3):
        ims = torch.from_numpy(synth.images(40 + step, 4))
        tg = synth.targets(50 + step, 4, g_max=8)
        opt.zero_grad()
        out = intended_forward(net, ims)
        l = net.loss(out, torch.from_numpy(tg), df_t)
        l.backward()
        opt.step()
        losses_ref.append(l.item())
        losses_o.append(O.train_step(sd_o, opt_state, ims, tg, df))
        print(f'  step {step}: ref loss {losses_ref[-1]:.6f}  oracle loss {losses_o[-1]:.6f}')
    assert np.allclose(losses_ref, losses_o, rtol=2e-3), (losses_ref, losses_o)
    fx = dict(losses=np.array(losses_ref, np.float64))
    rsd = net.state_dict()
    for k in ('detectors.det_4_3.weight', 'detectors.det_11_2.bias', 'features.conv_6_1.weight', 'features.bn_7_1.weight',
              'features.conv_9_2.weight', 'features.bn_11_2.bias'):
        fx['p_' + k] = rsd[k].numpy().reshape(-1)[::101].copy()
        d = (rsd[k] - sd_o[k].detach()).abs().max().item()
        print(f'    {k}: max|ref-oracle| after 3 steps = {d:.3e}')
    np.savez_compressed(OUT / 'g9_train.npz', **fx)
    print('G9 train trajectory ok')


def gen_eval():
    """G10: the reference's AP bookkeeping (evaluate.py:31-67).  evaluate.py imports `augmentation` (needs torchvision,
    absent here) at module level; a stub module with the two names it imports is enough to reach the two pure functions."""
    import types
    stub = types.ModuleType('augmentation'); stub.Compose = object; stub.ToTensor = object
    sys.modules['augmentation'] = stub
    import evaluate as ref_eval  # noqa: E402  (reference)
    from object_detection_torch2_amd import evaluate as my_eval
    fx = {}
    rng = synth.uniform(77, 5, 4000)
    k = 0
    for case, n in enumerate((1, 7, 40, 200)):
        tp = (rng[k:k + n] > 0.6).astype(np.float32); k += n
        sc = rng[k:k + n].astype(np.float32); k += n
        res = torch.from_numpy(np.stack([tp, sc], 1))
        cnt = int(tp.sum()) + case + 1
        ap = ref_eval.calc_average_precision(result=res.clone(), count=cnt)
        assert abs(ap.item() - my_eval.calc_average_precision(res.clone(), cnt, True).item()) < 1e-7
        fx[f'ap_ref_{case}'] = np.float32(ap.item()); fx[f'ap_cnt_{case}'] = np.int64(cnt)
        fx[f'ap_voc_{case}'] = np.float32(my_eval.calc_average_precision(res.clone(), cnt, False).item())
    t = torch.from_numpy(synth.uniform(78, 6, 50 * 25).astype(np.float32).reshape(50, 25)); t[::3, 5 + 4] = 0
    o = ref_eval.get_order(t, 4)
    assert torch.equal(o, my_eval.get_order(t, 4))
    fx['order_c4'] = o.numpy()
    np.savez_compressed(OUT / 'g10_eval.npz', **fx)
    print('G10 evaluation fixtures ok')


def gen_data():
    """G11: the reference's dataset.PascalVOCDataset (dataset.py:12-116) and utils.collate_fn (utils.py:8-16) on the tiny
    synthetic VOC tree of tests/voc_fixture.py: ground-truth rows, resized pixels, padded batch."""
    import tempfile
    import dataset as ref_ds  # noqa: E402  (reference; imports only utils, PIL, torch)
    sys.path.insert(0, str(ROOT / 'tests'))
    import voc_fixture
    from object_detection_torch2_amd import dataset as my_ds, utils as my_utils
    fx = {}
    with tempfile.TemporaryDirectory() as tmp:
        root = voc_fixture.make_tree(pathlib.Path(tmp))
        ref = ref_ds.PascalVOCDataset('detection', [str(root)], 'trainval.txt', 300)
        mine = my_ds.PascalVOCDataset('detection', [str(root)], 'trainval.txt', 300)
        assert len(ref) == len(mine) == len(voc_fixture.ITEMS)
        gts = []
        for i in range(len(ref)):
            (im_r, gt_r), (im_m, gt_m) = ref[i], mine[i]
            assert torch.equal(gt_r, gt_m), i
            assert np.array_equal(np.asarray(im_r), np.asarray(im_m)), i
            fx[f'gt_{i}'] = gt_r.numpy()
            fx[f'px_sum_{i}'] = np.asarray(im_r, dtype=np.int64).sum(axis=(0, 1))
            gts.append(gt_r)
        batch = [(torch.full((3, 4, 4), float(i)), g) for i, g in enumerate(gts)]
        ims_r, pad_r = ref_utils.collate_fn(batch)
        ims_m, pad_m = my_utils.collate_fn(batch)
        assert torch.equal(ims_r, ims_m) and torch.equal(pad_r, pad_m)
        fx['padded'] = pad_r.numpy()
        # purpose='classification' cannot be compared: the reference builds its list before it sets self.labelmap
        # (dataset.py:29-30 vs :83) and raises AttributeError; recorded so the behaviour is pinned as a known quirk
        try:
            ref_ds.PascalVOCDataset('classification', str(root), 'trainval.txt', 224)
            fx['ref_classification_raises'] = np.int64(0)
        except AttributeError:
            fx['ref_classification_raises'] = np.int64(1)
    np.savez_compressed(OUT / 'g11_data.npz', **fx)
    print('G11 data-path fixtures ok')


def gen_vgg16():
    """G13: the reference's VGG16 (model/vgg16.py) with transfer_learning=True on a 200-px batch (the only input sizes its
    Linear(512*7*7) accepts are 184..215 px): logits, the literal loss and its gradients w.r.t. the classifier2 stack, with the
    Dropout modules in eval mode (their masks come from torch's RNG: unpinnable) and BatchNorm in train mode like everywhere else.
    The weights file (hash-seeded, ~0.6 GB because of classifier2.0) lives in a temporary directory and is never committed."""
    import tempfile
    from model import VGG16 as RefVGG16
    sd = O.vgg16_init_state(seed=5, num_classes=20)
    x = torch.from_numpy(synth.images(77, 2, 200))
    tgt = torch.zeros(2, 20); tgt[0, 3] = 1; tgt[1, 11] = 1
    with tempfile.TemporaryDirectory(dir=str(ROOT / 'gpurun_out') if (ROOT / 'gpurun_out').exists() else None) as tmp:
        path = pathlib.Path(tmp) / 'vgg16_cls.pth'
        torch.save(sd, path)
        net = RefVGG16(weights_path=path, num_classes=20, transfer_learning=True)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    out = net(x)
    loss = net.loss(out, tgt)
    loss.backward()
    sdo = {k: v.clone() for k, v in O.vgg16_init_state(seed=5, num_classes=20).items() if not k.startswith('classifier.')}
    for k in sdo:
        if k.startswith('classifier2.'):
            sdo[k].requires_grad_(True)
    out_o = O.vgg16_forward(sdo, x, 'classifier2', training=True, masks=None)
    loss_o = O.vgg16_loss(out_o, tgt)
    loss_o.backward()
    print('VGG16 logits max|ref-oracle| =', (out - out_o).abs().max().item(), ' loss', loss.item(), loss_o.item())
    assert torch.equal(out, out_o) and loss.item() == loss_o.item()
    fx = dict(logits=out.detach().numpy(), loss=np.float64(loss.item()))
    for i in (0, 3, 6):
        gw, gb = net.classifier2[i].weight.grad, net.classifier2[i].bias.grad
        assert torch.allclose(gw, sdo[f'classifier2.{i}.weight'].grad, rtol=1e-5, atol=1e-9)
        fx[f'gw_{i}'] = gw.numpy().reshape(-1)[::(997 if i < 6 else 7)].copy()
        fx[f'gb_{i}'] = gb.numpy().copy()
    rm = net.features[1].running_mean
    fx['bn0_running_mean'] = rm.numpy().copy()
    # the reference's own train.py cannot reach this model: VGG16 has no train_params() (train.py:97)
    fx['ref_has_train_params'] = np.int64(hasattr(net, 'train_params'))
    np.savez_compressed(OUT / 'g13_vgg16.npz', **fx)
    print('G13 VGG16 classifier fixtures ok')


def gen_augment():
    """G12: the pixel arithmetic of the reference's colour jitter as it really runs -- train.py:32-36 applies RandomColorJitter
    BEFORE ToTensor, so torchvision's ColorJitter works on the PIL image: PIL.ImageEnhance for brightness / contrast / saturation,
    and for hue torchvision's PIL path (convert('HSV'), uint8 add on H, convert back; restated here with PIL primitives because
    torchvision is not installed).  Outputs below are PIL's own; the oracle must reproduce them bit for bit.  Also checks the
    oracle's two HSV conversions against PIL over ALL 2^24 colours."""
    from PIL import Image, ImageEnhance
    import PIL
    rng_u = synth.uniform(1234, 9, 24 * 20 * 3 + 16)
    imgs = [(rng_u[:24 * 20 * 3] * 256).astype(np.uint8).reshape(24, 20, 3),
            np.linspace(0, 255, 16 * 30 * 3).reshape(16, 30, 3).astype(np.uint8)[:, :, ::-1].copy()]
    imgs[0][:6, :5] = 77                                                  # gray patch: max == min branch of the HSV conversion
    enh = {0: ImageEnhance.Brightness, 1: ImageEnhance.Contrast, 2: ImageEnhance.Color}

    def pil_apply(arr, ops):
        im = Image.fromarray(arr, 'RGB')
        for op, prm in ops:
            if op < 3:
                im = enh[op](im).enhance(float(prm))
            else:                                                          # torchvision.transforms.functional_pil.adjust_hue
                h, s_, v = im.convert('HSV').split()
                np_h = np.array(h, dtype=np.uint8)
                np_h = (np_h.astype(np.int32) + (int(float(prm) * 255) & 255)).astype(np.uint8)   # np_h += np.uint8(hue_factor * 255), wrapping
                im = Image.merge('HSV', (Image.fromarray(np_h, 'L'), s_, v)).convert('RGB')
        return np.array(im)

    cases = []
    for f in (0.0, 0.25, 0.5, 0.77, 1.0, 1.2, 1.5):
        cases += [[(0, f)], [(1, f)], [(2, f)]]
    for hf in (-0.5, -0.37, -0.1, 0.0, 0.004, 0.2, 0.37, 0.5):
        cases.append([(3, hf)])
    cases += [[(0, 1.3), (1, 0.6), (2, 1.4), (3, 0.37)], [(3, -0.2), (2, 0.5), (1, 1.5), (0, 0.8)], [(1, 0.7), (3, 0.45), (0, 1.1), (2, 1.5)],
              [(2, 0.9), (0, 0.6), (3, -0.45), (1, 1.25)]]
    fx = dict(pil_version=np.array(PIL.__version__), n_cases=np.int64(len(cases)))
    for i, im in enumerate(imgs):
        fx[f'img_{i}'] = im
    for ci, ops in enumerate(cases):
        fx[f'ops_{ci}'] = np.array([[op, prm] for op, prm in ops], dtype=np.float64)
        for i, im in enumerate(imgs):
            want = pil_apply(im, ops)
            x = im.copy()
            for op, prm in ops:
                x = (O.pil_brightness, O.pil_contrast, O.pil_saturation)[op](x, prm) if op < 3 else O.pil_hue(x, O.hue_shift(prm))
            assert np.array_equal(x, want), (ci, i, ops)
            fx[f'out_{ci}_{i}'] = want
    # exhaustive: every RGB colour through convert('HSV'), every HSV triple through convert('RGB')
    a, b, c = np.meshgrid(*[np.arange(256, dtype=np.uint8)] * 3, indexing='ij')
    cube = np.stack([a, b, c], -1).reshape(4096, 4096, 3)
    assert np.array_equal(O.pil_rgb2hsv(cube), np.array(Image.fromarray(cube, 'RGB').convert('HSV')))
    assert np.array_equal(O.pil_hsv2rgb(cube), np.array(Image.fromarray(cube, 'HSV').convert('RGB')))
    assert np.array_equal(O.pil_luma(cube), np.array(Image.fromarray(cube, 'RGB').convert('L')))
    fx['exhaustive_hsv_checked'] = np.int64(1)
    np.savez_compressed(OUT / 'g12_augment_pil.npz', **fx)
    print(f'G12 augmentation (PIL {PIL.__version__}) fixtures ok: {len(cases)} cases x {len(imgs)} images, HSV conversions exhaustive')


if __name__ == '__main__':
    which = sys.argv[1:] or ['post', 'fwd', 'eval', 'data', 'augment', 'vgg16']
    if 'vgg16' in which:
        gen_vgg16()
    if 'augment' in which:
        gen_augment()
    if 'data' in which:
        gen_data()
    if 'eval' in which:
        gen_eval()
    if 'post' in which:
        gen_boxes_loss_post()
    if 'fwd' in which:
        gen_forward_train()
