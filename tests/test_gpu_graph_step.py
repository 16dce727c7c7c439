"""GPU: the train step replayed as HIP graphs (graph_step.py: backbone(i+1) on its own stream beside the tail of step i, hand-over
buffers alternating) against the eager step of the reference loop (train.py:115-122): losses, every parameter, every buffer (BN running
statistics, num_batches_tracked) and the Adam moments must be bit-identical -- same kernels, same per-stream order."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(dtype, imsize=300, seed=1):
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd.model import SSD
    from object_detection_torch2_amd.optim import FusedAdam
    net = SSD(num_classes=21, imsize=imsize)
    net.load_state_dict(O.init_state(seed=seed, imsize=imsize))
    net.to(DEV).set_compute_dtype(dtype)
    opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4, on_step=net.hip_weights_changed)
    return net, opt


def _batches(k, n, imsize=300):
    from object_detection_torch2_amd import synth
    return [(torch.from_numpy(synth.images(300 + i, n, imsize)).to(DEV), torch.from_numpy(synth.targets(400 + i, n, g_max=8)).to(DEV)) for i in range(k)]


def _state(net, opt):
    st = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for i, p in enumerate(net.train_params()):
        st[f'adam.{i}.m'] = opt.state[p]['exp_avg'].clone()
        st[f'adam.{i}.v'] = opt.state[p]['exp_avg_sq'].clone()
        st[f'adam.{i}.step'] = torch.tensor(int(opt.state[p]['step']))
    return st


def _run_eager(net, opt, df, batches, lrs=None):
    losses = []
    for i, (ims, tg) in enumerate(batches):
        if lrs is not None:
            opt.param_groups[0]['lr'] = lrs[i]
        opt.zero_grad()
        loss = net.loss(net(ims), tg, df)
        loss.backward()
        opt.step()
        losses.append(loss.detach().clone())
    torch.cuda.synchronize()
    return [l.item() for l in losses]


def _run_graph(net, opt, df, batches, lrs=None):
    from object_detection_torch2_amd.graph_step import GraphedTrainStep
    before = _state(net, opt) if opt.state else None
    runner = GraphedTrainStep(net, opt, df, batches[0][0].shape, batches[0][1].shape)
    if before is not None:                       # capturing (and its warm-up steps) left the training state alone
        after = _state(net, opt)
        assert all(torch.equal(before[k], after[k]) for k in before)
    losses = []
    runner.feed(batches[0][0])
    for i, (_, tg) in enumerate(batches):
        if lrs is not None:
            opt.param_groups[0]['lr'] = lrs[i]
        if i + 1 < len(batches):
            runner.feed(batches[i + 1][0])
        losses.append(runner.step(tg).clone())
    runner.drain()
    torch.cuda.synchronize()
    return [l.item() for l in losses], runner


@pytest.mark.parametrize('dtype,n', [('fp32', 4), ('bf16', 4), ('fp32', 32)])
def test_graphed_steps_are_bit_identical_to_eager_steps(dtype, n):
    steps = 5 if n <= 4 else 3
    batches = _batches(steps, n)
    lrs = [1e-3, 1e-3, 5e-4, 5e-4, 2.5e-4][:steps]              # an lr schedule reaches the captured Adam through its device table
    net_e, opt_e = _make(dtype)
    df = net_e.default_bboxes.to(DEV)
    want = _run_eager(net_e, opt_e, df, batches, lrs)
    net_g, opt_g = _make(dtype)
    got, runner = _run_graph(net_g, opt_g, df, batches, lrs)
    assert got == want, (got, want)
    se, sg = _state(net_e, opt_e), _state(net_g, opt_g)
    assert set(se) == set(sg)
    bad = [k for k in se if not torch.equal(se[k], sg[k])]
    assert not bad, bad[:8]
    assert all(np.isfinite(got))
    # the module is usable eagerly afterwards (packed weight images follow the replayed updates): one more eager step on both
    more = _batches(1, n)
    assert _run_eager(net_g, opt_g, df, more) == _run_eager(net_e, opt_e, df, more)
    bad = [k for k, v in _state(net_e, opt_e).items() if not torch.equal(v, _state(net_g, opt_g)[k])]
    assert not bad, bad[:8]


def test_graphed_step_ssd512_bf16_matches_eager():
    batches = _batches(3, 2, 512)
    net_e, opt_e = _make('bf16', 512)
    df = net_e.default_bboxes.to(DEV)
    want = _run_eager(net_e, opt_e, df, batches)
    net_g, opt_g = _make('bf16', 512)
    got, _ = _run_graph(net_g, opt_g, df, batches)
    assert got == want
    bad = [k for k, v in _state(net_e, opt_e).items() if not torch.equal(v, _state(net_g, opt_g)[k])]
    assert not bad, bad[:8]


def test_graphed_step_after_eager_steps_and_protocol_errors():
    from object_detection_torch2_amd.graph_step import GraphedTrainStep
    batches = _batches(4, 2)
    net_e, opt_e = _make('fp32')
    df = net_e.default_bboxes.to(DEV)
    want = _run_eager(net_e, opt_e, df, batches)
    net_g, opt_g = _make('fp32')
    first = _run_eager(net_g, opt_g, df, batches[:2])            # Adam moments and step count exist when the runner is built
    rest, runner = _run_graph(net_g, opt_g, df, batches[2:])
    assert first + rest == want
    with pytest.raises(RuntimeError, match='feed'):
        runner.step(batches[0][1])
    runner.feed(batches[0][0]); runner.feed(batches[1][0])
    with pytest.raises(RuntimeError, match='in flight'):
        runner.feed(batches[2][0])
    with pytest.raises(RuntimeError, match='never trained'):
        runner.drain()
    with pytest.raises(ValueError):
        runner.step(batches[0][1][:1])
    runner.step(batches[0][1]); runner.step(batches[1][1]); runner.drain()
    with pytest.raises(ValueError):
        runner.feed(batches[0][0][:1])
    opt_g.load_state_dict(opt_g.state_dict())                    # replaces the moment tensors: the captured pointer table is stale
    runner.feed(batches[0][0])
    with pytest.raises(RuntimeError, match='stale'):
        runner.step(batches[0][1])
    net_f, opt_f = _make('fp32')
    net_f.finetune_backbone()
    with pytest.raises(RuntimeError, match='frozen-backbone'):
        GraphedTrainStep(net_f, opt_f, df, batches[0][0].shape, batches[0][1].shape)
    with pytest.raises(TypeError):
        GraphedTrainStep(net_e, torch.optim.Adam(net_e.train_params()), df, batches[0][0].shape, batches[0][1].shape)


def test_graphed_step_cut_at_the_allreduce_points_with_one_rank_rccl():
    """Data-parallel form: the captured tail is cut at the engine's two hook points and the RCCL all-reduce runs eagerly between the
    segments (tests/tools/graph_dp_probe.py: a one-rank `nccl` group, so the sum is the identity and everything must equal the eager
    hooked steps bit for bit)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'tools', 'graph_dp_probe.py')], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert 'segments per tail: 3 graphs + 2 collectives' in r.stdout and 'graphed == eager (hooked): True' in r.stdout


def test_train_entry_point_graph_mode_matches_eager(tmp_path, capsys):
    """`train.py --graph` on synthetic batches: same epoch losses and the same weights.pth as the eager loop."""
    from object_detection_torch2_amd import train
    outs = []
    for mode in ([], ['--graph']):
        torch.manual_seed(11)
        d = tmp_path / ('g' if mode else 'e')
        train.main(['--batch_size', '2', '--epochs', '2', '--steps_per_epoch', '3', '--result_dir', str(d)] + mode)
        outs.append((capsys.readouterr().out, torch.load(d / 'detection' / 'weights.pth', weights_only=True)))
    le = [l.split('(')[0] for l in outs[0][0].splitlines() if l.startswith('[Epoch')]
    lg = [l.split('(')[0] for l in outs[1][0].splitlines() if l.startswith('[Epoch')]
    assert le == lg and len(le) == 2, (le, lg)
    assert all(torch.equal(outs[0][1][k], outs[1][1][k]) for k in outs[0][1])
