"""CPU oracle for the SSD300 hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (object_detection_torch2_amd) never does and
fails loudly when its HIP library is missing.

It is a restatement of the reference algorithm written from SURVEY.md, not a
copy: box / matching / mining / loss / NMS arithmetic is numpy fp32 in the
reference's operation order (bit-exact targets), and the dense floating-point
layers (conv / batch-norm / max-pool) call stock torch CPU ops, because that is
the third-party arithmetic the reference itself delegates to (PyTorch, version
unpinned by the reference: docker/Dockerfile:2).

Pinning: the reference has no tests or golden vectors, so this
oracle is pinned by tests/golden/*.npz, generated HERE by importing the
reference's own model/utils modules on CPU (tests/golden/make_golden.py).
"""
import math
import numpy as np

f32 = np.float32

# ----------------------------------------------------------------------------
# topology (src/model/vgg16.py:25-37, src/model/ssd.py:28-77)
# ----------------------------------------------------------------------------
VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M_P', 512, 512, 512, 'M', 512, 512, 512, 'M_P']
EXTRA_CFG = [[(3, 1024, 1, 1)], [(1, 1024, 1, 0)], [(1, 256, 1, 0), (3, 512, 2, 1)],
             [(1, 128, 1, 0), (3, 256, 2, 1)], [(1, 128, 1, 0), (3, 256, 1, 0)], [(1, 128, 1, 0), (3, 256, 1, 0)]]
DET_CFG = [('4_3', 512, 4), ('7_1', 1024, 6), ('8_2', 512, 6), ('9_2', 256, 6), ('10_2', 256, 4), ('11_2', 256, 4)]
BOX_CFG = [[38, 38, 4], [19, 19, 6], [10, 10, 6], [5, 5, 6], [3, 3, 4], [1, 1, 4]]
# SSD512 (build-side extension, no reference: ssd.py:118 hard-codes the 300-px grid): BASELINE configs[4]'s 24,564 priors =
# maps 64/32/16/8/4/2/1; pool_3 without padding, five stride-2 extras.  Same rules as above otherwise.
TOPOLOGIES = {
    300: dict(vgg=VGG_CFG, extra=EXTRA_CFG, det=DET_CFG, box=BOX_CFG),
    512: dict(vgg=[64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M_P'],
              extra=[[(3, 1024, 1, 1)], [(1, 1024, 1, 0)], [(1, 256, 1, 0), (3, 512, 2, 1)]] + [[(1, 128, 1, 0), (3, 256, 2, 1)]] * 4,
              det=[('4_3', 512, 4), ('7_1', 1024, 6), ('8_2', 512, 6), ('9_2', 256, 6), ('10_2', 256, 6), ('11_2', 256, 4), ('12_2', 256, 4)],
              box=[[64, 64, 4], [32, 32, 6], [16, 16, 6], [8, 8, 6], [4, 4, 6], [2, 2, 4], [1, 1, 4]]),
}
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def feature_layers(imsize=300):
    """Ordered feature layer table: dicts with kind conv/bn/act/pool (ssd.py:28-66)."""
    layers, cin, stage, sub = [], 3, 1, 1
    for v in TOPOLOGIES[imsize]['vgg']:
        if v in ('M', 'M_P'):
            if stage < 5:                                   # pool_5 dropped, ssd.py:38
                layers.append(dict(kind='pool', name=f'pool_{stage}', pad=1 if v == 'M_P' else 0))
            stage, sub = stage + 1, 1
        else:
            tag = f'{stage}_{sub}'
            layers.append(dict(kind='conv', name=f'conv_{tag}', cin=cin, cout=v, k=3, stride=1, pad=1, frozen=True))
            layers.append(dict(kind='bn', name=f'bn_{tag}', c=v, frozen=True))
            layers.append(dict(kind='act', name=f'act_{tag}'))
            cin, sub = v, sub + 1
    for group in TOPOLOGIES[imsize]['extra']:
        sub = 1
        for (k, v, s, p) in group:
            tag = f'{stage}_{sub}'
            layers.append(dict(kind='conv', name=f'conv_{tag}', cin=cin, cout=v, k=k, stride=s, pad=p, frozen=False))
            layers.append(dict(kind='bn', name=f'bn_{tag}', c=v, frozen=False))
            layers.append(dict(kind='act', name=f'act_{tag}'))
            cin, sub = v, sub + 1
        stage += 1
    return layers


def detector_layers(num_classes=21, imsize=300):
    return [dict(name=f'det_{t}', src=f'act_{t}', cin=c, cout=a * (num_classes + 4), anchors=a) for t, c, a in TOPOLOGIES[imsize]['det']]


def trainable_param_names(num_classes=21, imsize=300):
    """Order of SSD.train_params() (ssd.py:160-179)."""
    out = []
    for l in feature_layers(imsize):
        if l['kind'] in ('conv', 'bn') and not l['frozen']:
            out += [f"features.{l['name']}.weight", f"features.{l['name']}.bias"]
    for d in detector_layers(num_classes, imsize):
        out += [f"detectors.{d['name']}.weight", f"detectors.{d['name']}.bias"]
    return out


# ----------------------------------------------------------------------------
# A9 default boxes (ssd.py:108-133): python doubles rounded once to fp32
# ----------------------------------------------------------------------------
def default_bboxes(imsize=300):
    box_cfg = TOPOLOGIES[imsize]['box']

    def s_(k, m=len(box_cfg), s_min=0.2, s_max=0.9):
        return s_min + (s_max - s_min) * (k - 1) / (m - 1)
    rows = []
    for k, (m, n, a_num) in enumerate(box_cfg, start=1):
        aspects = [1, 2, 1 / 2, 'add'] if a_num == 4 else [1, 2, 1 / 2, 3, 1 / 3, 'add']
        for i in range(m):
            for j in range(n):
                for a in aspects:
                    if a == 'add':
                        w = h = (s_(k) * s_(k + 1)) ** 0.5
                    else:
                        w = s_(k) * (a ** 0.5)
                        h = s_(k) * ((1 / a) ** 0.5)
                    rows.append(((i + 0.5) / m, (j + 0.5) / n, w, h))   # cx follows the ROW index (Q4)
    return np.asarray(rows, dtype=np.float64).astype(f32)


# ----------------------------------------------------------------------------
# A10-A16 MultiBox loss (ssd.py:181-328), numpy fp32, reference op order
# ----------------------------------------------------------------------------
def match(gt, df, threshold=0.25):
    """(N,G,>=4),(P,4) -> (N,P,G) bool.  ssd.py:231-250; every op is a separately rounded fp32 op."""
    g = gt[:, None, :, :4].astype(f32)
    d = df[None, :, None, :].astype(f32)
    g_cx, g_cy, g_w, g_h = (g[..., i] for i in range(4))
    d_cx, d_cy, d_w, d_h = (d[..., i] for i in range(4))
    two = f32(2)
    w = np.maximum(np.minimum(g_cx + g_w / two, d_cx + d_w / two) - np.maximum(g_cx - g_w / two, d_cx - d_w / two), f32(0))
    h = np.maximum(np.minimum(g_cy + g_h / two, d_cy + d_h / two) - np.maximum(g_cy - g_h / two, d_cy - d_h / two), f32(0))
    ga = g_w * g_h
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = np.where(ga > 0, (w * h) / (ga + d_w * d_h - w * h), ga)
    return iou > f32(threshold)


def calc_delta(gt, df):
    """g-hat (N,P,G,4), ssd.py:252-272 (no variance scaling)."""
    g = gt[:, None, :, :4].astype(f32)
    d = df[None, :, None, :].astype(f32)
    out = np.empty(np.broadcast_shapes(g.shape, d.shape), dtype=f32)
    out[..., 0] = (g[..., 0] - d[..., 0]) / d[..., 2]
    out[..., 1] = (g[..., 1] - d[..., 1]) / d[..., 3]
    with np.errstate(divide='ignore', invalid='ignore'):
        out[..., 2] = np.where(g[..., 2] > 0, np.log(np.where(g[..., 2] > 0, g[..., 2], f32(1)) / d[..., 2]), g[..., 2])
        out[..., 3] = np.where(g[..., 3] > 0, np.log(np.where(g[..., 3] > 0, g[..., 3], f32(1)) / d[..., 3]), g[..., 3])
    return out


def smooth_l1(x):
    ax = np.abs(x)
    return np.where(ax < 1, f32(0.5) * x * x, ax - f32(0.5)).astype(f32)


def log_softmax(x):
    """F.log_softmax(dim=-1) (ssd.py:298): x - max - log(sum(exp(x - max)))."""
    m = x.max(axis=-1, keepdims=True)
    z = (x - m).astype(f32)
    return (z - np.log(np.exp(z).sum(axis=-1, keepdims=True, dtype=f32))).astype(f32)


def split_pos_neg(pos_num, neg_num):
    """ssd.py:300-311, integer exact."""
    cond = pos_num * 3 > neg_num
    return np.where(cond, neg_num // 3, pos_num), np.where(cond, neg_num, pos_num * 3)


def k_plus_1_th_value(v, k):
    """ssd.py:313-328: k>0 -> (k+1)-th largest (= kthvalue(P-k) ascending); k==0 -> max."""
    if k > 0:
        return np.partition(v, len(v) - k - 1)[len(v) - k - 1]
    return v.max()


def loss_terms(outputs, targets, df, a=1.0):
    """Per-(n,p) terms of SSD.loss.  Returns dict with is_match (N,P,G), l_loc, l_conf_pos, l_conf_neg (N,P),
    pos_num/neg_num after split (N,), kth thresholds, pos_valid/neg_valid (N,P), loss scalar (fp32)."""
    outputs = outputs.astype(f32); targets = targets.astype(f32); df = df.astype(f32)
    N, P, _ = outputs.shape
    G = targets.shape[1]
    is_match = match(targets, df)                                    # ssd.py:199
    ghat = calc_delta(targets, df)                                   # ssd.py:203
    lsm = log_softmax(outputs[:, :, 4:])                             # (N,P,C)
    l_loc = np.zeros((N, P), f32); l_conf_pos = np.zeros((N, P), f32)
    for g in range(G):                                               # sum(dim=2) runs g = 0..G-1 in order
        d = outputs[:, :, :4] - ghat[:, :, g, :]
        sl = smooth_l1(d)
        s4 = ((sl[..., 0] + sl[..., 1]) + sl[..., 2]) + sl[..., 3]
        m = is_match[:, :, g].astype(f32)
        l_loc = l_loc + s4 * m                                       # ssd.py:204
        ce = -(targets[:, None, g, 4:] * lsm).sum(axis=-1, dtype=f32)   # ssd.py:298, one-hot -> -lsm[label]
        l_conf_pos = l_conf_pos + ce * m                             # ssd.py:209
    not_match = is_match.sum(axis=2) == 0
    l_conf_neg = (-lsm[:, :, 0]) * not_match.astype(f32)             # ssd.py:212-215
    pos_num = (is_match.sum(axis=2) != 0).sum(axis=1).astype(np.int64)
    neg_num = P - pos_num
    pos_k, neg_k = split_pos_neg(pos_num, neg_num)                   # ssd.py:218-220
    thr_pos = np.array([k_plus_1_th_value(l_conf_pos[i], int(pos_k[i])) for i in range(N)], f32)
    thr_neg = np.array([k_plus_1_th_value(l_conf_neg[i], int(neg_k[i])) for i in range(N)], f32)
    pos_valid = l_conf_pos > thr_pos[:, None]                        # strict, ssd.py:222
    neg_valid = l_conf_neg > thr_neg[:, None]
    inv = np.where(pos_k > 0, f32(1) / np.maximum(pos_k, 1).astype(f32), f32(0)).astype(f32)
    per_prior = (f32(a) * l_loc + l_conf_pos) * pos_valid + l_conf_neg * neg_valid
    per_image = per_prior.sum(axis=1, dtype=np.float64).astype(f32) * inv
    loss = f32(per_image.sum(dtype=np.float64) / N)
    return dict(is_match=is_match, l_loc=l_loc, l_conf_pos=l_conf_pos, l_conf_neg=l_conf_neg, pos_num=pos_k,
                neg_num=neg_k, thr_pos=thr_pos, thr_neg=thr_neg, pos_valid=pos_valid, neg_valid=neg_valid,
                inv_pos=inv, per_image=per_image, loss=loss, ghat=ghat, lsm=lsm)


def loss(outputs, targets, df, a=1.0):
    return loss_terms(outputs, targets, df, a)['loss']


def loss_grad(outputs, targets, df, a=1.0, grad_out=1.0):
    """Analytic dL/d outputs (N,P,4+C) of SSD.loss: the masks (is_match, pos_valid, neg_valid) and the
    normaliser are constants under autograd (comparisons / integer ops), ssd.py:218-227."""
    t = loss_terms(outputs, targets, df, a)
    outputs = outputs.astype(f32); targets = targets.astype(f32)
    N, P, _ = outputs.shape
    G = targets.shape[1]
    grad = np.zeros_like(outputs, dtype=f32)
    sm = np.exp(t['lsm'])
    coef = (t['inv_pos'] * f32(grad_out) / f32(N))[:, None]          # (N,1)
    wpos = t['pos_valid'].astype(f32) * coef
    wneg = (t['neg_valid'] & (t['is_match'].sum(axis=2) == 0)).astype(f32) * coef
    for g in range(G):
        m = t['is_match'][:, :, g].astype(f32) * wpos                # (N,P)
        d = outputs[:, :, :4] - t['ghat'][:, :, g, :]
        with np.errstate(invalid='ignore'):
            dsl = np.where(np.abs(d) < 1, d, np.sign(d)).astype(f32)
        grad[:, :, :4] += f32(a) * dsl * m[..., None]
        onehot = targets[:, None, g, 4:]                             # (N,1,C)
        grad[:, :, 4:] += (sm * onehot.sum(axis=-1, keepdims=True) - onehot) * m[..., None]
    void = np.zeros(outputs.shape[2] - 4, f32); void[0] = 1
    grad[:, :, 4:] += (sm - void) * wneg[..., None]
    return grad


# ----------------------------------------------------------------------------
# A19-A22 post-process (utils.py:19-116)
# ----------------------------------------------------------------------------
def calc_coordicate(pr, df):
    pr = pr.astype(f32); d = df[None].astype(f32)
    out = np.empty(pr.shape[:2] + (4,), f32)
    out[..., 0] = d[..., 2] * pr[..., 0] + d[..., 0]
    out[..., 1] = d[..., 3] * pr[..., 1] + d[..., 1]
    out[..., 2] = d[..., 2] * np.exp(pr[..., 2])
    out[..., 3] = d[..., 3] * np.exp(pr[..., 3])
    return out


def calc_score(pr):
    """softmax(pr[:,:,4:]) * onehot(argmax): one non-zero per row (utils.py:43-55)."""
    x = pr[:, :, 4:].astype(f32)
    m = x.max(axis=-1, keepdims=True)
    e = np.exp((x - m).astype(f32))
    sm = (e / e.sum(axis=-1, keepdims=True, dtype=f32)).astype(f32)
    arg = x.argmax(axis=-1)                                           # first maximal index
    mask = np.zeros_like(sm); np.put_along_axis(mask, arg[..., None], 1.0, axis=-1)
    return sm * mask


def calc_iou(t, s):
    """(N,T,>=4),(N,S,>=4) -> (N,T,S), utils.py:58-77 (guard on w*h>0, differs from match())."""
    t = t[:, :, None, :4].astype(f32); s = s[:, None, :, :4].astype(f32)
    two = f32(2)
    w = np.maximum(np.minimum(t[..., 0] + t[..., 2] / two, s[..., 0] + s[..., 2] / two) -
                   np.maximum(t[..., 0] - t[..., 2] / two, s[..., 0] - s[..., 2] / two), f32(0))
    h = np.maximum(np.minimum(t[..., 1] + t[..., 3] / two, s[..., 1] + s[..., 3] / two) -
                   np.maximum(t[..., 1] - t[..., 3] / two, s[..., 1] - s[..., 3] / two), f32(0))
    inter = w * h
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(inter > 0, inter / (t[..., 2] * t[..., 3] + s[..., 2] * s[..., 3] - inter), inter)


def nms_keep(output, iou_thresh=0.5, per_class=False):
    """Greedy class-agnostic NMS keep list for one image (P,4+C), utils.py:90-109.
    Candidates: rows with max(cols 5:) > 0, sorted by that score descending, ties by lower index.
    per_class (north-star option, not the reference): only boxes of the same arg-max class suppress each other."""
    score = output[:, 5:].max(axis=1)
    order = np.argsort(-score.astype(np.float64), kind='stable')
    order = order[score[order] > 0]
    boxes = output[order, :4].astype(f32)
    cls = output[order, 5:].argmax(axis=1)
    alive = np.ones(len(order), bool)
    for i in range(len(order)):
        if not alive[i] or i + 1 >= len(order):
            continue
        iou = calc_iou(boxes[None, i:i + 1], boxes[None, i + 1:])[0, 0]
        hit = iou > f32(iou_thresh)
        if per_class:
            hit &= cls[i + 1:] == cls[i]
        alive[i + 1:] &= ~hit
    return order[alive]


def non_maximum_suppression(outputs, iou_thresh=0.5):
    """Mutates outputs[:,:,4:] in place like the reference (utils.py:114) and returns it."""
    N, P, _ = outputs.shape
    for n in range(N):
        mask = np.zeros(P, f32)
        mask[nms_keep(outputs[n], iou_thresh)] = 1
        outputs[n, :, 4:] *= mask[:, None]
    return outputs


# ----------------------------------------------------------------------------
# evaluation bookkeeping (src/evaluate.py:31-42 get_order, :134-151 the per-image / per-class loop body)
# ----------------------------------------------------------------------------
def get_order(t, class_id):
    """evaluate.py:31-42: rows with a positive value in column 5 + class_id, by descending value (ties: lower row first; the
    reference's torch.sort leaves them unspecified)."""
    v = t[:, 5 + class_id]
    order = np.argsort(-v.astype(np.float64), kind='stable')
    return order[v[order] > 0]


def eval_image(output, gt, num_classes=20, iou_thresh=0.5):
    """evaluate.py:134-151 for one image: output (P,4+C) decoded / scored / NMS-filtered, gt (G,4+C).
    Returns ({class: (X,2) rows [correct, score] in score order}, {class: number of GT boxes})."""
    iou = calc_iou(output[None], gt[None])[0]
    rows, counts = {}, {}
    for c in range(num_classes):
        pr_order, gt_order = get_order(output, c), get_order(gt, c)
        counts[c] = len(gt_order)
        if len(pr_order) == 0:
            continue
        if len(gt_order) == 0:
            correct = np.zeros((len(pr_order), 1), f32)
        else:
            iou_one = iou[pr_order][:, gt_order]
            valid = np.eye(len(gt_order), dtype=f32)[iou_one.argmax(axis=1)] * (iou_one > f32(iou_thresh))
            correct = ((np.cumsum(valid, axis=0) == valid) * valid).sum(axis=1, keepdims=True)
        rows[c] = np.concatenate([correct.astype(f32), output[pr_order][:, [5 + c]]], axis=1)
    return rows, counts


# ----------------------------------------------------------------------------
# A2-A8 forward on torch CPU ops (floating-point layers)
# ----------------------------------------------------------------------------
def init_state(seed=1, num_classes=21, vgg_random=True, imsize=300):
    """Reference-keyed state_dict (173 entries) with hash-seeded weights: kaiming fan_out pseudo-normal for every
    conv (the VGG part too -- no pretrained weights exist offline), bias 0, BN gamma 1 / beta 0 / stats 0,1."""
    import torch
    from object_detection_torch2_amd import synth
    sd, stream = {}, 100
    def conv(prefix, cout, cin, k):
        nonlocal stream
        std = math.sqrt(2.0 / (cout * k * k))
        w = synth.pseudo_normal(seed, stream, cout * cin * k * k) * std
        stream += 1
        sd[prefix + '.weight'] = torch.from_numpy(w.astype(f32).reshape(cout, cin, k, k))
        sd[prefix + '.bias'] = torch.zeros(cout)
    for l in feature_layers(imsize):
        if l['kind'] == 'conv':
            conv('features.' + l['name'], l['cout'], l['cin'], l['k'])
        elif l['kind'] == 'bn':
            p = 'features.' + l['name']
            sd[p + '.weight'] = torch.ones(l['c']); sd[p + '.bias'] = torch.zeros(l['c'])
            sd[p + '.running_mean'] = torch.zeros(l['c']); sd[p + '.running_var'] = torch.ones(l['c'])
            sd[p + '.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)
    for d in detector_layers(num_classes, imsize):
        conv('detectors.' + d['name'], d['cout'], d['cin'], 3)
    return sd


def forward(sd, x, num_classes=21, training=True, collect=None, imsize=300):
    """Intended SSD.forward contract (N,P,4+C) (ssd.py:86-106 with the Q1 membership fix).
    sd: state dict of torch tensors (requires_grad honoured); running stats updated in place when training."""
    import torch
    import torch.nn.functional as F
    mean = torch.tensor(MEAN, dtype=torch.float32).reshape(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).reshape(1, 3, 1, 1)
    x = x.sub(mean).div(std)                                          # vgg16.py:112-114
    dets = {d['src']: d for d in detector_layers(num_classes, imsize)}
    ys = []
    for l in feature_layers(imsize):
        p = 'features.' + l['name']
        if l['kind'] == 'conv':
            x = F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], stride=l['stride'], padding=l['pad'])
        elif l['kind'] == 'bn':
            if training:
                sd[p + '.num_batches_tracked'] += 1
            x = F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                             training=training, momentum=0.1, eps=1e-5)
        elif l['kind'] == 'act':
            x = F.relu(x)
            if collect is not None:
                collect[l['name']] = x
            if l['name'] in dets:
                d = dets[l['name']]
                q = 'detectors.' + d['name']
                y = F.conv2d(x, sd[q + '.weight'], sd[q + '.bias'], padding=1)
                ys.append(y.permute(0, 2, 3, 1).reshape(x.shape[0], -1, num_classes + 4))
        else:
            x = F.max_pool2d(x, 2, 2, padding=l['pad'])
    return torch.cat(ys, dim=1)


def forward_bf16(sd, x, num_classes=21, imsize=300):
    """CPU restatement of the PACKAGE's bf16 compute mode (BASELINE configs[3] / [4]; no reference counterpart: the reference is fp32 only).
    It rounds where the kernels round, so that a bf16 run can be compared at rounding level instead of against fp32:
      * conv_1_1 (RGB) stays fp32; every other convolution and every detector head multiplies bf16 operands with fp32 accumulation:
        the input is relu(bn(z_prev)) (+ 2x2 max-pool) computed in fp32 and rounded ONCE to bf16, the weights are rounded to bf16;
      * BatchNorm (train mode) takes its statistics from the fp32 accumulators; where a frozen layer's pre-BN map is only read by the
        next layer's apply pass (conv_1_2 .. conv_5_2 except conv_4_3, which feeds a head) that map is stored as bf16 and the apply pass
        reads the rounded values;
      * backward: the gradient dy arriving at a convolution or head output is rounded to bf16 for its weight and data gradient (the
        staging of wgrad_bf16_kernel / the BF16 template); BatchNorm / ReLU backward, bias gradients, loss and parameters are fp32.
    Train mode only (the reference never calls eval()); running statistics are updated like nn.BatchNorm2d."""
    import torch
    import torch.nn.functional as F

    def r16(t):                                    # round to nearest even, straight-through for autograd
        return t + (t.detach().bfloat16().float() - t.detach())

    class GradRound(torch.autograd.Function):      # identity; the gradient passing back through is rounded to bf16
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            return g.bfloat16().float()

    mean = torch.tensor(MEAN, dtype=torch.float32).reshape(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).reshape(1, 3, 1, 1)
    x = x.sub(mean).div(std)
    layers = feature_layers(imsize)
    dets = {d['src']: d for d in detector_layers(num_classes, imsize)}
    convs = [l for l in layers if l['kind'] == 'conv']
    last_frozen = [l['name'] for l in convs if l['frozen']][-1]
    ys, z, z_used, first = [], None, None, True
    for li, l in enumerate(layers):
        p = 'features.' + l['name']
        if l['kind'] == 'conv':
            w, b = sd[p + '.weight'], sd[p + '.bias']
            if first:                              # conv_1_1: fp32 kernel on the normalised image
                z = F.conv2d(x, w, b, stride=l['stride'], padding=l['pad'])
                first = False
            else:
                z = GradRound.apply(F.conv2d(x, r16(w), None, stride=l['stride'], padding=l['pad'])) + b.view(1, -1, 1, 1)
            tag = l['name'].replace('conv_', '')
            stored_bf16 = l['frozen'] and l['name'] not in ('conv_1_1', last_frozen) and ('act_' + tag) not in dets
            z_used = r16(z) if stored_bf16 else z
        elif l['kind'] == 'bn':
            g, be = sd[p + '.weight'], sd[p + '.bias']
            mu = z.mean(dim=(0, 2, 3))
            var = z.var(dim=(0, 2, 3), unbiased=False)
            with torch.no_grad():
                n = z.numel() / z.shape[1]
                sd[p + '.num_batches_tracked'] += 1
                sd[p + '.running_mean'].mul_(0.9).add_(0.1 * mu)
                sd[p + '.running_var'].mul_(0.9).add_(0.1 * var * n / (n - 1))
            scale = g / torch.sqrt(var + 1e-5)
            shift = be - mu * scale
            x = z_used * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        elif l['kind'] == 'act':
            x = F.relu(x)
            if l['name'] in dets:
                d = dets[l['name']]
                q = 'detectors.' + d['name']
                y = GradRound.apply(F.conv2d(r16(x), r16(sd[q + '.weight']), None, padding=1)) + sd[q + '.bias'].view(1, -1, 1, 1)
                ys.append(y.permute(0, 2, 3, 1).reshape(x.shape[0], -1, num_classes + 4))
            nxt = layers[li + 1] if li + 1 < len(layers) else None
            if nxt is None or nxt['kind'] != 'pool':
                x = r16(x)                         # the next convolution's operand: one rounding behind BN.ReLU
        else:
            x = r16(F.max_pool2d(x, 2, 2, padding=l['pad']))          # ... or behind BN.ReLU + pool (max commutes with the rounding)
    return torch.cat(ys, dim=1)


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=5e-4):
    """torch.optim.Adam single-tensor semantics (train.py:97): L2 decay added to grad, bias-corrected."""
    g = g + wd * p
    m[:] = b1 * m + (1 - b1) * g
    v[:] = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step; bc2 = 1 - b2 ** step
    p[:] = p - (lr / bc1) * m / (np.sqrt(v) / math.sqrt(bc2) + eps)
    return p


def train_step(sd, opt_state, images, targets_np, df, lr=1e-3, wd=5e-4, num_classes=21, imsize=300):
    """One reference train-loop body (train.py:115-122) on CPU: forward -> loss -> backward -> Adam.
    Loss value and dL/d outputs come from the numpy restatement; torch autograd carries it through the layers."""
    import torch
    names = trainable_param_names(num_classes, imsize)
    for k in names:
        sd[k].requires_grad_(True); sd[k].grad = None
    out = forward(sd, images, num_classes, training=True, imsize=imsize)
    o = out.detach().numpy()
    lval = loss(o, targets_np, df)
    out.backward(torch.from_numpy(loss_grad(o, targets_np, df)))
    opt_state['step'] = opt_state.get('step', 0) + 1
    with torch.no_grad():
        for k in names:
            st = opt_state.setdefault(k, dict(m=np.zeros(sd[k].shape, f32), v=np.zeros(sd[k].shape, f32)))
            adam_step(sd[k].numpy(), sd[k].grad.numpy(), st['m'], st['v'], opt_state['step'], lr=lr, wd=wd)
    return float(lval)


# ----------------------------------------------------------------------------
# VGG16 classifier (src/model/vgg16.py:7-130): reference-keyed state dict, forward on torch CPU ops, the literal loss
# ----------------------------------------------------------------------------
def vgg16_init_state(seed=5, num_classes=20, head='classifier2'):
    """State dict with the reference VGG16's keys (features.<idx>.*, classifier.*, classifier2.*).  Hash-seeded: kaiming fan_out
    pseudo-normal convs, BN gamma 1 / beta 0 / stats 0, 1; the stack named by `head` uniform(-0.01, 0.01) weights and small biases,
    the other stack zeros as stride-0 views (it is never used by forward(); a real tensor would be 0.5 GB)."""
    import torch
    from object_detection_torch2_amd import synth
    sd, stream, idx, cin = {}, 300, 0, 3
    for v in VGG_CFG:
        if v in ('M', 'M_P'):
            idx += 1
            continue
        std = math.sqrt(2.0 / (v * 9))
        sd[f'features.{idx}.weight'] = torch.from_numpy((synth.pseudo_normal(seed, stream, v * cin * 9) * std).astype(f32).reshape(v, cin, 3, 3))
        stream += 1
        sd[f'features.{idx}.bias'] = torch.zeros(v)
        sd[f'features.{idx + 1}.weight'] = torch.ones(v); sd[f'features.{idx + 1}.bias'] = torch.zeros(v)
        sd[f'features.{idx + 1}.running_mean'] = torch.zeros(v); sd[f'features.{idx + 1}.running_var'] = torch.ones(v)
        sd[f'features.{idx + 1}.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)
        idx, cin = idx + 3, v
    for name, last in (('classifier', 1000), ('classifier2', num_classes)):
        for i, (o, c) in zip((0, 3, 6), ((4096, 25088), (4096, 4096), (last, 4096))):
            if name == head:
                sd[f'{name}.{i}.weight'] = torch.from_numpy(((synth.uniform(seed, stream, o * c) - 0.5) * 0.02).astype(f32).reshape(o, c))
                sd[f'{name}.{i}.bias'] = torch.from_numpy(((synth.uniform(seed, stream + 1, o) - 0.5) * 0.2).astype(f32))
                stream += 2
            else:
                sd[f'{name}.{i}.weight'] = torch.zeros(1).expand(o, c)
                sd[f'{name}.{i}.bias'] = torch.zeros(1).expand(o)
    return sd


def dropout_mask(seed, stream, n, p=0.5):
    """The device pass's mask (csrc/classifier.hip): keep where the 24-bit counter-hash uniform >= p; returned already scaled by
    1 / (1 - p).  (torch's own Philox stream is not reproduced: which units drop is unpinned by nature.)"""
    from object_detection_torch2_amd import synth
    keep = (synth.hash_u32(seed, stream, n) >> np.uint64(8)) >= np.uint64(int(p * 16777216.0))
    return keep.astype(f32) * f32(1.0 / (1.0 - p))


def vgg16_forward(sd, x, head='classifier2', training=True, masks=None):
    """vgg16.py:85-101: normalize -> features -> x.view(N, -1) -> classifier stack.  masks: None = no dropout (eval-mode Dropout
    modules), else the two (N, 4096) scaled masks applied after the ReLUs."""
    import torch
    import torch.nn.functional as F
    mean = torch.tensor(MEAN, dtype=torch.float32).reshape(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).reshape(1, 3, 1, 1)
    x = x.sub(mean).div(std)
    idx = 0
    for v in VGG_CFG:
        if v in ('M', 'M_P'):
            x = F.max_pool2d(x, 2, 2, padding=1 if v == 'M_P' else 0)
            idx += 1
            continue
        p, q = f'features.{idx}', f'features.{idx + 1}'
        x = F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], padding=1)
        if training:
            sd[q + '.num_batches_tracked'] += 1
        x = F.relu(F.batch_norm(x, sd[q + '.running_mean'], sd[q + '.running_var'], sd[q + '.weight'], sd[q + '.bias'], training=training,
                                momentum=0.1, eps=1e-5))
        idx += 3
    x = x.reshape(x.shape[0], -1)
    for k, i in enumerate((0, 3)):
        x = F.relu(F.linear(x, sd[f'{head}.{i}.weight'], sd[f'{head}.{i}.bias']))
        if masks is not None:
            x = x * masks[k]
    return F.linear(x, sd[f'{head}.6.weight'], sd[f'{head}.6.bias'])


def vgg16_loss(outputs, targets):
    """vgg16.py:127-129, literally (no minus sign): mean_n sum_c t * log_softmax(outputs)."""
    import torch.nn.functional as F
    return (targets * F.log_softmax(outputs, dim=1)).sum(dim=1).mean()


# ----------------------------------------------------------------------------
# batch augmentation (src/augmentation/random.py:6-42 as the reference runs it: train.py:32-36 puts RandomColorJitter BEFORE
# ToTensor, i.e. torchvision's ColorJitter on a PIL image -> PIL.ImageEnhance for brightness / contrast / saturation, an HSV
# round trip for hue, uint8 after every operation; then PIL hflip, ToTensor (x / 255) and RandomErasing with value 0).
# torchvision is not importable here; Pillow (the third-party arithmetic: 12.2.0 in this image, unpinned by the reference) is, so
# these restatements are pinned by PIL's own outputs: tests/golden/g12_augment_pil.npz, and the two HSV conversions were compared
# with PIL over all 2^24 colours (tests/golden/make_golden.py gen_augment).
# ----------------------------------------------------------------------------
def pil_luma(rgb):
    """Image.convert('L'): ITU-R 601-2 luma in 16.16 fixed point."""
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def pil_blend(d, x, alpha):
    """Image.blend(d, x, alpha) on uint8 arrays (libImaging/Blend.c): float arithmetic d + alpha (x - d), truncated; outside
    0 <= alpha <= 1 clipped to [0, 255] first.  alpha is a C float."""
    a = f32(alpha)
    di, xi = d.astype(np.int32), x.astype(np.int32)
    t = (di.astype(f32) + (a * (xi - di).astype(f32)).astype(f32)).astype(f32)
    if 0.0 <= float(a) <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def pil_brightness(x, f):          # ImageEnhance.Brightness: degenerate = black
    return pil_blend(np.zeros_like(x), x, f)


def pil_contrast(x, f):            # ImageEnhance.Contrast: degenerate = int(mean(L) + 0.5)
    L = pil_luma(x)
    mean = int(int(L.astype(np.int64).sum()) / L.size + 0.5)
    return pil_blend(np.full_like(x, mean), x, f)


def pil_saturation(x, f):          # ImageEnhance.Color: degenerate = L
    return pil_blend(np.repeat(pil_luma(x)[..., None], 3, axis=-1), x, f)


def pil_rgb2hsv(rgb):
    """Image.convert('HSV') (libImaging/Convert.c rgb2hsv_row): float / double mix as in the C source."""
    r, g, b = (rgb[..., k].astype(np.int32) for k in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(f32)
    crs = np.where(cr == 0, f32(1), cr)
    s = (cr / np.where(maxc == 0, 1, maxc).astype(f32)).astype(f32)
    rc, gc, bc = (((maxc - c).astype(f32) / crs).astype(f32) for c in (r, g, b))
    f64 = np.float64
    h = np.where(r == maxc, (bc - gc).astype(f32),
                 np.where(g == maxc, (f64(2.0) + rc.astype(f64) - bc.astype(f64)).astype(f32),
                          (f64(4.0) + gc.astype(f64) - rc.astype(f64)).astype(f32)))
    h = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(f32)
    uh = np.clip((h.astype(f64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(f64) * 255.0).astype(np.int32), 0, 255)
    eq = maxc == minc
    return np.stack([np.where(eq, 0, uh), np.where(eq, 0, us), maxc], -1).astype(np.uint8)


def pil_hsv2rgb(hsv):
    """Image.convert('RGB') from HSV (libImaging/Convert.c hsv2rgb)."""
    f64 = np.float64
    s, v = hsv[..., 1], hsv[..., 2]
    hh = hsv[..., 0].astype(f32).astype(f64) * 6.0 / 255.0
    i = np.floor(hh)
    f = (hh - i.astype(f32).astype(f64)).astype(f32).astype(f64)
    fs = (s.astype(f32).astype(f64) / 255.0).astype(f32).astype(f64)
    vf = v.astype(f32).astype(f64)
    rnd = lambda a: np.clip(np.floor(a + 0.5), 0, 255).astype(np.uint8)          # C round() of a non-negative value, CLIP8
    p, q, t = rnd(vf * (1.0 - fs)), rnd(vf * (1.0 - fs * f)), rnd(vf * (1.0 - fs * (1.0 - f)))
    ii = i.astype(np.int32) % 6
    out = np.zeros(hsv.shape, np.uint8)
    for k, (a_, b_, c_) in enumerate([(v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q)]):
        m = ii == k
        out[..., 0][m], out[..., 1][m], out[..., 2][m] = a_[m], b_[m], c_[m]
    g = s == 0
    for c in range(3):
        out[..., c][g] = v[g]
    return out


def hue_shift(hue_factor):
    """torchvision adjust_hue (PIL path): np_h += np.uint8(hue_factor * 255) with uint8 wrap-around; the cast truncates toward
    zero and wraps (numpy >= 2 refuses the negative cast outright, the wrap is what the function documents and relies on)."""
    return int(float(hue_factor) * 255) & 255


def pil_hue(x, shift):
    hsv = pil_rgb2hsv(x)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + int(shift)) & 255
    return pil_hsv2rgb(hsv)


def augment_batch(u8, rows):
    """u8 (N,H,W,3) uint8, rows (N,24) fp32 (layout: csrc/augment.hip header) -> (N,3,H,W) fp32."""
    N, H, W, _ = u8.shape
    out = np.zeros((N, 3, H, W), dtype=f32)
    for n in range(N):
        x = u8[n].copy()
        row = rows[n]
        for k in range(int(row[0])):
            op = int(row[1 + k])
            if op == 0:
                x = pil_brightness(x, row[5])
            elif op == 1:
                x = pil_contrast(x, row[6])
            elif op == 2:
                x = pil_saturation(x, row[7])
            else:
                x = pil_hue(x, int(row[8]))
        if row[9] != 0:
            x = x[:, ::-1]
        xf = (x.astype(f32) / f32(255)).astype(f32)           # ToTensor
        for k in range(int(row[10])):
            top, left, h, w = (int(v) for v in row[11 + 4 * k: 15 + 4 * k])
            xf[top:top + h, left:left + w] = 0
        out[n] = xf.transpose(2, 0, 1)
    return out
