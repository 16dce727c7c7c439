"""GPU: rehearsal of the data-parallel step with two ranks sharing the one GPU of the box (gloo instead of RCCL): the
overlapped all-reduce must deliver exactly sum(local gradients) and identical weights on every rank after the step."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_data_parallel_step_on_one_gpu():
    env = dict(os.environ, SSD_DIST_BACKEND='gloo', SSD_FORCE_DEVICE='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'scripts', 'dp_check.py')]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'weights identical across ranks after step: True' in r.stdout


def test_rccl_one_rank_group_carries_the_gradient_allreduce():
    """RCCL itself on this box's one GPU (scripts/rccl_world1_check.py): a one-rank `nccl` group runs the step's real all-reduce pattern
    (early bucket next to the last weight gradient, asynchronous handles, RCCL's stream against the engine's three) over three
    unsynchronised steps; with one rank the sum is the identity, so everything must be bit-identical to the steps without the hook."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'rccl_world1_check.py')], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert 'backend nccl' in r.stdout and 'identical to the unhooked steps: True' in r.stdout


def test_two_rank_classification_step_on_one_gpu():
    """`train.py --purpose classification` under data parallelism: the classifier stack's gradients (fc1: 411 MB) travel
    through the asynchronous hook like the feature gradients, per-rank dropout seeds, identical weights after the step."""
    env = dict(os.environ, SSD_DIST_BACKEND='gloo', SSD_FORCE_DEVICE='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29537', os.path.join(ROOT, 'scripts', 'dp_check_vgg.py')]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'weights identical across ranks after step: True' in r.stdout


def test_two_rank_pipelined_steps_match_serialised_streams():
    """Several data-parallel steps without host synchronisation, backbone overlap enabled (tests/tools/dp_probe.py): bit-identical to the
    same run with the streams serialised, and identical weights on both ranks."""
    env = dict(os.environ, SSD_DIST_BACKEND='gloo', SSD_FORCE_DEVICE='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29535', os.path.join(ROOT, 'tests', 'tools', 'dp_probe.py')]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('pipelined == serialised: True; ranks identical: True') == 2


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_bench_self_launch_two_ranks_on_one_gpu(dtype):
    """The driver's command line for N > 1 (`python bench.py --gpus N ...`, no launcher): the parent starts the ranks; here 2 ranks share
    the box's one GPU over gloo.  One JSON line, world size 2 seen by the all-reduce, communication fields present.  bf16 = the
    data-parallel form of BASELINE configs[3] (fp32 gradients in the all-reduce, fp32 master weights)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(SSD_DIST_BACKEND='gloo', SSD_FORCE_DEVICE='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2', '--batch', '4',
                        '--no-cpu-baseline', '--dtype', dtype], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = lines[0]
    assert line['n_gpus'] == 2 and line['ranks'] == 2 and line['config']['global_batch'] == 8 and line['scaling'] == 'weak'
    assert line['value'] > 0 and line['comm_ms_per_step'] > 0 and 'comm_exposed_ms_per_step' in line
    assert line['roofline']['bound'] == 'mfma' and line['dtype'] == ('bf16' if dtype == 'bf16' else 'f32')


@pytest.mark.parametrize('graph', [False, True])
def test_train_device_count_self_launch_two_ranks_on_one_gpu(tmp_path, graph):
    """`train.py --device-count 2` without a launcher: the script starts its two ranks itself (gloo, both on the box's one GPU).
    With --graph the captured tail is cut at the two all-reduce points and REAL two-rank collectives run between the segments; the
    epoch losses must equal the eager data-parallel run's."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(SSD_DIST_BACKEND='gloo', SSD_FORCE_DEVICE='0', PYTHONPATH=ROOT + os.pathsep + env.get('PYTHONPATH', ''))
    outs = []
    for mode in ([], ['--graph']) if graph else ([],):
        r = subprocess.run([sys.executable, '-m', 'object_detection_torch2_amd.train', '--device-count', '2', '--batch_size', '2', '--epochs', '2',
                            '--steps_per_epoch', '3', '--seed', '7', '--result_dir', str(tmp_path / ('g' if mode else 'e'))] + mode,
                           env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert 'Finished Training' in r.stdout
        outs.append([l.split('(')[0] for l in r.stdout.splitlines() if l.startswith('[Epoch')])
        assert len(outs[-1]) == 2
    if graph:
        assert outs[0] == outs[1], outs
