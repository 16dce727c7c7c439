"""GPU: the BASELINE.json configs at their PER-GPU size, on the HIP path, against the CPU oracle.

configs[3] = SSD300 bf16 forward / backward + fp32 master at batch 32 per GPU; configs[4] = SSD512 bf16 at batch 16 per GPU (their
8-GPU forms shard these by image, tests/test_dist_cpu.py / test_gpu_dist.py).  The reference is fp32 only, so what pins a bf16 run
is the fp32 oracle of the same network at bf16-sized bounds (the bounds of tests/test_gpu_bf16.py, taken at batch 2 / 4) plus the
size-independent properties of tests/test_gpu_model.py: bit-identical replay, batch-permutation equivariance, a finite Adam step.
The oracle's forward + tail backward at these sizes is a few seconds of torch CPU convolutions on the box's host cores."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


# Bounds of the HIP bf16 mode against oracle.forward_bf16, the CPU restatement that rounds where the kernels round.  Measured (SSD300 N=32 /
# SSD512 N=16): outputs 3.2 % / 3.3 % relative L2 (8.5 % / 8.8 % against the fp32 oracle), loss within 3e-5, det_4_3 / det_7_1 gradients
# 0.17 % / 0.66 %.  It cannot get tighter at model level: rounding to bf16 is a step function, an fp32-level difference in an accumulator
# (1e-6) moves a rounded activation by a whole 2^-9 with probability 2.5e-4, i.e. ~6e-5 of relative noise injected per layer, and this
# random-init network carries per-layer noise to its outputs ~300-500x (fp32's 6e-8 shows as 3e-5 against the fp32 oracle).
BF16_ORACLE_OUT, BF16_ORACLE_LOSS = 0.06, 5e-4
BF16_ORACLE_REL = {'detectors.det_4_3': 0.01, 'detectors.det_7_1': 0.02, 'features.bn_7_1': 0.03, 'detectors.det_8_2': 0.05, 'features.bn_8_2': 0.06,
                   'detectors.det_9_2': 0.10}
BF16_ORACLE_COS = 0.55        # every other trainable tensor (measured 0.77 on bn_12_1.bias of SSD512 .. 1.0000)


def _oracle_run(O, imsize, ims, tg, num_classes=21, seed=1):
    sd = O.init_state(seed=seed, num_classes=num_classes, imsize=imsize)
    names = O.trainable_param_names(num_classes, imsize=imsize)
    df = O.default_bboxes(imsize)
    for k in names:
        sd[k].requires_grad_(True)
    out_o = O.forward(sd, torch.from_numpy(ims), num_classes=num_classes, training=True, imsize=imsize)
    o = out_o.detach().numpy()
    out_o.backward(torch.from_numpy(O.loss_grad(o, tg, df)))
    return sd, names, df, o, float(O.loss(o, tg, df))


def _bf16_config_case(imsize, N, first_head_priors, seeds):
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import synth
    from object_detection_torch2_amd.model import SSD
    from object_detection_torch2_amd.optim import FusedAdam
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    ims, tg = synth.images(seeds[0], N, imsize), synth.targets(seeds[1], N, g_max=8)
    sd, names, df, o, loss_ref = _oracle_run(O, imsize, ims, tg)
    net = SSD(num_classes=21, imsize=imsize)
    net.load_state_dict(O.init_state(seed=1, imsize=imsize))
    net.to(DEV).set_compute_dtype('bf16')
    opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4, on_step=net.hip_weights_changed)
    x, t, d = torch.from_numpy(ims).to(DEV), torch.from_numpy(tg).to(DEV), torch.from_numpy(df).to(DEV)
    P = df.shape[0]
    # ---- size-independent properties first (they need the untouched weights): replay and permutation
    with torch.no_grad():
        y1 = net(x).clone()
        y2 = net(x).clone()
        assert torch.equal(y1, y2)                                   # no atomics, no LDS race: same bits
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(3)).to(DEV)
        yp = net(x[perm].contiguous())
        # train-mode BN statistics are permutation invariant up to summation order; a bf16 rounding boundary crossed by that
        # order moves one operand by 2^-9: relative L2 (the absolute bound of the fp32 test does not apply to bf16 operands)
        assert ((yp - y1[perm]).norm() / y1.norm()).item() < 2e-2
    assert tuple(y1.shape) == (N, P, 25) and torch.isfinite(y1).all()
    # ---- forward / loss / gradients vs the fp32 oracle
    opt.zero_grad()
    out = net(x)
    dlt = out.detach().cpu().numpy() - o
    rel_all = np.linalg.norm(dlt) / np.linalg.norm(o)
    rel_head = np.linalg.norm(dlt[:, :first_head_priors]) / np.linalg.norm(o[:, :first_head_priors])
    assert rel_all < 0.10 and rel_head < 0.05, (rel_all, rel_head)   # bounds of the batch-2 / batch-4 bf16 tests
    loss = net.loss(out, t, d)
    assert abs(loss.item() - loss_ref) < 0.02 * abs(loss_ref), (loss.item(), loss_ref)
    loss.backward()
    got = dict(net.named_train_params())
    report = {}
    for k in names:
        if not k.endswith('.weight') or '.bn_' in k:
            continue
        g, r = got[k].grad.cpu().double(), sd[k].grad.double()
        assert torch.isfinite(g).all(), k
        report[k] = ((g * r).sum().item() / (g.norm().item() * r.norm().item()), ((g - r).norm() / r.norm()).item())
    print(f'bf16 SSD{imsize} N={N}: out rel L2 {rel_all:.4f} (first head {rel_head:.4f}), loss {loss.item():.5f} vs {loss_ref:.5f}; weight gradients '
          '(cos, rel L2): ' + ', '.join(f'{k.split(".")[1]} {c:.3f}/{e:.3f}' for k, (c, e) in report.items()))
    for k, (cos, rel) in report.items():
        # Against the FP32 oracle.  The heads on the frozen / first trainable maps see bf16 noise once: det_4_3 / det_7_1 agree to
        # 0.3 % / 1.3 % (measured, both configs).  Everything deeper sits behind 2-12 more bf16 layers with train-mode BatchNorm over few
        # pixels in a random-init network that amplifies input noise ~16x: rounding at 2^-9 flips ReLU units all the
        # way down, and a trunk gradient differs from the fp32 one by 30-90 % in L2 while staying positively aligned (measured cos 0.94
        # on conv_6_1 down to 0.50 on conv_12_1; fp32 against fp32 already differs by 1e-2 for the same reason, test_gpu_model.py).
        # So: tight bounds where one rounding is all there is, direction only below, and the training curve (profiles/r02_r_*) for the rest.
        name = k.split('.')[1]
        if name in ('det_4_3', 'det_7_1'):
            assert cos > 0.999 and rel < 0.03, (k, cos, rel)
        elif name in ('det_8_2', 'det_9_2'):
            assert cos > 0.98 and rel < 0.2, (k, cos, rel)
        else:
            assert cos > 0.3, (k, cos, rel)
    # ---- against the oracle's restatement of the bf16 mode itself (same rounding points: oracle.forward_bf16): rounding-level agreement
    sd16 = O.init_state(seed=1, imsize=imsize)
    for k in names:
        sd16[k].requires_grad_(True)
    out16 = O.forward_bf16(sd16, torch.from_numpy(ims), imsize=imsize)
    o16 = out16.detach().numpy()
    out16.backward(torch.from_numpy(O.loss_grad(o16, tg, df)))
    d16 = out.detach().cpu().numpy() - o16
    rel16 = np.linalg.norm(d16) / np.linalg.norm(o16)
    loss16 = float(O.loss(o16, tg, df))
    rep16 = {}
    for k in names:
        if 'features.conv' in k and k.endswith('.bias'):
            continue
        g, r = got[k].grad.cpu().double(), sd16[k].grad.double()
        rep16[k] = ((g * r).sum().item() / (g.norm().item() * r.norm().item() + 1e-300), ((g - r).norm() / (r.norm() + 1e-300)).item())
    print(f'bf16 SSD{imsize} N={N} vs the bf16-rounding oracle: out rel L2 {rel16:.5f} max |d| {np.abs(d16).max():.4f}, loss {loss.item():.5f} vs {loss16:.5f}; '
          'gradients (cos, rel L2): ' + ', '.join(f'{k.split(".", 1)[1]} {c:.4f}/{e:.4f}' for k, (c, e) in rep16.items()))
    assert rel16 < BF16_ORACLE_OUT and abs(loss.item() - loss16) < BF16_ORACLE_LOSS * abs(loss16), (rel16, loss.item(), loss16)
    for k, (cos, rel) in rep16.items():
        assert cos > BF16_ORACLE_COS and rel < BF16_ORACLE_REL.get(k.rsplit('.', 1)[0], 1.0), (k, cos, rel)
    before = {k: got[k].detach().clone() for k in ('detectors.det_4_3.weight', 'features.conv_6_1.weight', 'features.bn_7_1.bias')}
    opt.step()
    torch.cuda.synchronize()
    for k, v in before.items():
        assert not torch.equal(v, got[k]) and torch.isfinite(got[k]).all() and got[k].dtype == torch.float32, k   # fp32 master moved
        assert (got[k] - v).abs().max().item() <= 1.001e-3 * (1 + 5e-4 * v.abs().max().item()) + 1e-9, k       # |Adam step| <= lr
    # second step runs on the updated weights (repacked images, bf16 weight copies) and lowers the loss on the same batch
    opt.zero_grad()
    loss2 = net.loss(net(x), t, d)
    assert torch.isfinite(loss2) and loss2.item() < loss.item()
    # running statistics stay fp32 state, finite, and every BatchNorm counted its five forward passes (two replays, one permuted, two steps)
    st = net.state_dict()
    assert all(torch.isfinite(v.float()).all() for v in st.values())
    assert int(st['features.bn_4_3.num_batches_tracked']) == 5 and int(st['features.bn_7_1.num_batches_tracked']) == 5


def test_config3_ssd300_bf16_batch32_vs_oracle():
    """BASELINE configs[3], one GPU's share: SSD300, batch 32, bf16 forward / backward on fp32 master parameters."""
    _bf16_config_case(300, 32, 5776, (100, 200))


def test_config4_ssd512_bf16_batch16_vs_oracle():
    """BASELINE configs[4], one GPU's share: SSD512 (24,564 priors), batch 16, bf16."""
    _bf16_config_case(512, 16, 64 * 64 * 4, (161, 162))


def test_config4_ssd512_fp32_batch16_vs_oracle():
    """The fp32 form of the same shard: forward 2e-3, loss at the north-star 1e-4, detector gradients 1e-3."""
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import synth
    from object_detection_torch2_amd.model import SSD
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    N = 16
    ims, tg = synth.images(161, N, 512), synth.targets(162, N, g_max=8)
    sd, names, df, o, loss_ref = _oracle_run(O, 512, ims, tg)
    net = SSD(num_classes=21, imsize=512)
    net.load_state_dict(O.init_state(seed=1, imsize=512))
    net.to(DEV)
    out = net(torch.from_numpy(ims).to(DEV))
    assert np.abs(out.detach().cpu().numpy() - o).max() < 2e-3
    loss = net.loss(out, torch.from_numpy(tg).to(DEV), torch.from_numpy(df).to(DEV))
    assert abs(loss.item() - loss_ref) < 1e-4, (loss.item(), loss_ref)
    loss.backward()
    got = dict(net.named_train_params())
    for k in names:
        if k.startswith('detectors.'):
            g, r = got[k].grad.cpu().double(), sd[k].grad.double()
            assert ((g - r).norm() / r.norm()).item() < 1e-3, k


def test_five_classes_batch32_tight_bounds():
    """The five-class forward at a non-degenerate batch: the batch-of-two case of test_gpu_classes.py needs 4e-3 on its last 40 priors
    (BatchNorm over 2 and 18 samples per channel amplifies backbone rounding); at batch 32 EVERY prior, the 3x3 and 1x1 maps included,
    is held at 1e-3 and the loss at the north-star 1e-4 -- the widened small-batch bound is a BatchNorm artefact, not drift."""
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import synth
    from object_detection_torch2_amd.model import SSD
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    C, N = 5, 32
    ims, tg = synth.images(70, N), synth.targets(71, N, g_max=5, num_classes=C)
    sd, names, df, o, loss_ref = _oracle_run(O, 300, ims, tg, num_classes=C, seed=3)
    net = SSD(num_classes=C)
    net.load_state_dict(O.init_state(seed=3, num_classes=C))
    net.to(DEV)
    out = net(torch.from_numpy(ims).to(DEV))
    err = np.abs(out.detach().cpu().numpy() - o)
    assert err.max() < 1e-3, (err[:, :8692].max(), err[:, 8692:].max())
    loss = net.loss(out, torch.from_numpy(tg).to(DEV), torch.from_numpy(df).to(DEV))
    assert abs(loss.item() - loss_ref) < 1e-4, (loss.item(), loss_ref)
    loss.backward()
    got = dict(net.named_train_params())
    for k in names:
        if k.startswith('detectors.'):
            g, r = got[k].grad.cpu().double(), sd[k].grad.double()
            assert ((g - r).norm() / r.norm()).item() < 1e-3, k


def test_entry_points_reach_ssd512_and_bf16(tmp_path, capsys):
    """train.py / inference.py / evaluate.py with `--imsize 512 --dtype bf16` (the reference's flags, train.py:15-27, plus --dtype): the
    flags reach SSD(...) -- the checkpoint holds the 24,564-prior network (det_12_2, 189 entries) and the other two load it."""
    from object_detection_torch2_amd import evaluate, inference, train
    res = tmp_path / 'result'
    torch.manual_seed(5)
    train.main(['--imsize', '512', '--dtype', 'bf16', '--batch_size', '2', '--epochs', '1', '--steps_per_epoch', '2', '--result_dir', str(res)])
    out = capsys.readouterr().out
    assert 'Finished Training' in out
    sd = torch.load(res / 'detection' / 'weights.pth', map_location='cpu', weights_only=True)
    assert len(sd) == 189 and 'detectors.det_12_2.weight' in sd and all(torch.isfinite(v.float()).all() for v in sd.values())
    inference.main(['--imsize', '512', '--dtype', 'bf16', '--batch_size', '2', '--batches', '1', '--result_dir', str(res)])
    out = capsys.readouterr().out
    assert 'weights loaded.' in out and 'Finished Inference' in out and 'image 2:' in out
    evaluate.main(['--imsize', '512', '--dtype', 'bf16', '--batch_size', '2', '--batches', '1', '--result_dir', str(res)])
    out = capsys.readouterr().out
    assert 'weights loaded.' in out and 'Finished Evaluate' in out
    with pytest.raises(SystemExit):
        train.main(['--dtype', 'fp16'])
