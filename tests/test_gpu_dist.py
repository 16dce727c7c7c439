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


def test_two_rank_pipelined_steps_match_serialised_streams():
    """Several data-parallel steps without host synchronisation, backbone overlap enabled (scripts/dp_probe.py): bit-identical to the
    same run with the streams serialised, and identical weights on both ranks."""
    env = dict(os.environ, SSD_DIST_BACKEND='gloo', SSD_FORCE_DEVICE='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29535', os.path.join(ROOT, 'scripts', 'dp_probe.py')]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('pipelined == serialised: True; ranks identical: True') == 2
