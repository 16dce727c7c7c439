"""CPU, world_size 2 over gloo: the data-parallel exchange step (bucketed all-reduce of the flat gradient buffer),
batch sharding and BN-stat averaging.  The kernels themselves need a GPU; the collective logic does not."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from object_detection_torch2_amd import parallel
    from object_detection_torch2_amd.model import SSD
    r, w, _ = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    lo, hi = parallel.shard_batch(64, rank, world)
    # per-rank "gradient": deterministic function of the shard, 46.3 MB-like layout shrunk to 100k floats (odd size on purpose)
    n = 100003
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    hook = parallel.GradAllReduce(n_buckets=3)
    ranges = hook.bucket_ranges(n)
    cut = 40000                      # gradients behind `cut` are final first (overlap with the last wgrad), then the head
    hook.start(g, cut, n)
    hook.finish(g)
    expect = torch.arange(n, dtype=torch.float32) * sum(range(1, world + 1))
    ok_sum = torch.equal(g, expect)
    g2 = torch.ones(1000) * (rank + 1)
    parallel.GradAllReduce(n_buckets=1, average=True)(g2)
    ok_avg = torch.allclose(g2, torch.full((1000,), sum(range(1, world + 1)) / world))
    # BN running statistics averaging on a real module (CPU tensors)
    net = SSD(21)
    with torch.no_grad():
        for name, b in net.named_buffers():
            if name.endswith('running_mean'):
                b.fill_(float(rank))
    parallel.sync_bn_stats(net)
    rm = net.features['bn_6_1'].running_mean
    ok_bn = torch.allclose(rm, torch.full_like(rm, (world - 1) / 2))
    # selectable exchange: reduce-scatter + all-gather in place, odd sizes (remainder path), bf16 payload
    ok_modes = True
    for mode, payload in (('rs_ag', 'fp32'), ('allreduce', 'bf16'), ('rs_ag', 'bf16')):
        h = parallel.GradAllReduce(n_buckets=3, mode=mode, payload=payload)
        v = torch.arange(n, dtype=torch.float32).remainder(251.0) * (rank + 1)       # integers <= 250 * 3: exact in bf16 sums? (<= 256 * 3)
        h.start(v, cut, n)
        h.finish(v)
        want = torch.arange(n, dtype=torch.float32).remainder(251.0) * sum(range(1, world + 1))
        if payload == 'fp32':
            ok_modes &= torch.equal(v, want)
        else:
            ok_modes &= bool(((v - want).abs() <= want.abs() * 2 ** -7 + 1e-6).all())    # three bf16 roundings of 8-bit mantissas
        d = h.describe()
        ok_modes &= d['ranks_seen'] == world and d['mode'] == mode and d['payload'] == payload and len(d['collectives_last_step']) >= 3
        ok_modes &= sum(e[1] for e in d['collectives_last_step']) == n                # every element travelled exactly once
    try:
        parallel.GradAllReduce(mode='ring')
        ok_modes = False
    except ValueError:
        pass
    q.put((rank, lo, hi, ranges, ok_sum, ok_avg, ok_bn, ok_modes))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_allreduce_and_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 32), (32, 64)]          # contiguous shards: rank r gets [32r, 32r+32)
    for r in res:
        ranges = r[3]
        assert ranges[0][0] == 0 and ranges[-1][1] == 100003 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        assert all(lo % 4 == 0 for lo, _ in ranges)                   # 16-byte aligned buckets
        assert r[7], 'rs_ag / bf16 exchange modes'
        assert r[4] and r[5] and r[6]


def test_shard_batch_rejects_ragged():
    from object_detection_torch2_amd import parallel
    with pytest.raises(ValueError):
        parallel.shard_batch(65, 0, 2)


# ---------------------------------------------------------------------------------------------------------------------
# bench.py --gpus N without a launcher: the parent starts the ranks itself (the driver's command line)
# ---------------------------------------------------------------------------------------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(extra)
    return env


def test_bench_self_launches_two_ranks():
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--dry-run'],
                       env=_bench_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    assert lines[0]['ranks'] == 2 and lines[0]['n_gpus'] == 2 and lines[0]['allreduce_ok'] and lines[0]['dry_run']


def test_bench_failing_rank_fails_the_job():
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--dry-run'],
                       env=_bench_env(SSD_BENCH_TEST_FAIL_RANK='1'), capture_output=True, text=True, timeout=300)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


# ---------------------------------------------------------------------------------------------------------------------
# train.py under data parallelism: same start weights on every rank, rank-consistent checkpoint decision
# ---------------------------------------------------------------------------------------------------------------------
def _train_worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import pathlib
    from object_detection_torch2_amd import parallel, train
    from object_detection_torch2_amd.model import SSD
    parallel.init_from_env(backend='gloo')
    torch.manual_seed(1234 + 77 * rank)              # what unseeded processes amount to: a different RNG state per rank
    own = SSD(21)
    own_sum = float(sum(p.double().sum() for p in own.parameters()))
    torch.manual_seed(1234 + 77 * rank)
    net = train.build_model(torch.device('cpu'), world)          # train.py's construction path
    with torch.no_grad():
        net.features['bn_7_1'].running_mean.fill_(float(rank + 1))
    digest = torch.cat([p.detach().reshape(-1) for p in net.parameters()] + [b.detach().reshape(-1).float() for b in net.buffers()])
    gathered = [torch.empty_like(digest) for _ in range(world)]
    # buffers were set per rank AFTER the broadcast on purpose: compare parameters only here
    n_par = sum(p.numel() for p in net.parameters())
    dist.all_gather(gathered, digest)
    same_start = all(torch.equal(gathered[0][:n_par], g[:n_par]) for g in gathered)
    # epoch ends with rank-divergent shard losses: rank 0 keeps improving, rank 1 gets worse after epoch 1.  With per-rank
    # decisions rank 1 would skip sync_bn_stats (a collective) in epochs 2 and 3 and rank 0 would hang in it.
    wp, pp = pathlib.Path(tmp) / 'detection' / 'weights.pth', pathlib.Path(tmp) / 'detection' / 'params.json'
    local = {0: [5.0, 4.0, 3.0, 9.0], 1: [5.0, 6.0, 6.5, 9.0]}[rank]
    min_loss, hist = None, []
    for epoch, l in enumerate(local, start=1):
        g, min_loss = train.end_of_epoch(net, rank, world, epoch, l, min_loss, 1e-3, wp, pp)
        hist.append((g, min_loss))
    rm = float(net.features['bn_7_1'].running_mean[0])
    q.put((rank, own_sum, same_start, hist, rm, wp.exists() if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_train_dp_same_start_and_rank_consistent_checkpoint_decision(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] != res[1][1]                           # the two processes really drew different initial weights ...
    assert res[0][2] and res[1][2]                          # ... and build_model made them equal (rank 0's)
    assert res[0][3] == res[1][3]                           # same global loss and same min_loss on every rank, every epoch
    assert [round(g, 6) for g, _ in res[0][3]] == [5.0, 5.0, 4.75, 9.0]
    assert [m for _, m in res[0][3]] == [5.0, 5.0, 4.75, 4.75]     # improved in epochs 1 and 3 only
    assert res[0][4] == res[1][4] == 1.5                    # BN running stats were averaged (collective ran on both ranks)
    assert res[0][5]


def test_attach_survives_finetune_backbone_in_both_orders():
    from object_detection_torch2_amd import parallel
    from object_detection_torch2_amd.model import SSD
    a = SSD(21)
    parallel.attach(a)
    hook = a._engine().dist_hook
    assert hook is not None
    a.finetune_backbone()
    assert a._engine().dist_hook is hook                    # the rebuilt engine got the same hook
    b = SSD(21)
    b.finetune_backbone()
    parallel.attach(b)
    assert b._engine().dist_hook is not None


# ---------------------------------------------------------------------------------------------------------------------
# resume under data parallelism: only rank 0 loaded optimizer state; the other ranks hold none (the collective
# sequence must not depend on what a rank happens to hold)
# ---------------------------------------------------------------------------------------------------------------------
def _opt_state_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from object_detection_torch2_amd import parallel
    from object_detection_torch2_amd.optim import FusedAdam
    parallel.init_from_env(backend='gloo')
    torch.manual_seed(5)
    lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    out = []
    # (a) stock torch.optim.Adam: rank 0 took three real steps (state: exp_avg, exp_avg_sq, step tensor), rank 1 none
    opt = torch.optim.Adam(lin.parameters(), lr=1e-2)
    if rank == 0:
        for k in range(3):
            opt.zero_grad()
            lin(torch.full((2, 7), float(k + 1))).sum().backward()
            opt.step()
    parallel.broadcast_module(lin)
    parallel.broadcast_optimizer_state(opt)
    digest = torch.cat([opt.state[p][k].reshape(-1) for p in lin.parameters() for k in ('exp_avg', 'exp_avg_sq')] +
                       [torch.stack([torch.as_tensor(float(opt.state[p]['step'])) for p in lin.parameters()])])
    got = [torch.empty_like(digest) for _ in range(world)]
    dist.all_gather(got, digest)
    out.append(all(torch.equal(got[0], g) for g in got) and float(opt.state[next(lin.parameters())]['step']) == 3.0
               and torch.is_tensor(opt.state[next(lin.parameters())]['step']) and digest.abs().sum().item() > 0)
    # one more identical step on both ranks must keep them identical (the materialised state is a valid Adam state)
    opt.zero_grad()
    lin(torch.ones(2, 7)).sum().backward()
    opt.step()
    w = torch.cat([p.detach().reshape(-1) for p in lin.parameters()])
    ws = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    out.append(all(torch.equal(ws[0], x) for x in ws))
    # (b) FusedAdam's layout (Python int step; step() itself needs a GPU, the state exchange does not)
    fo = FusedAdam(lin.parameters(), lr=1e-3)
    if rank == 0:
        for i, p in enumerate(lin.parameters()):
            fo.state[p] = dict(step=41, exp_avg=torch.full_like(p, 0.5 + i), exp_avg_sq=torch.full_like(p, 0.25 + i))
    fo._plans = {0: 'stale'}
    parallel.broadcast_optimizer_state(fo)
    ok = fo._plans == {}
    for i, p in enumerate(lin.parameters()):
        st = fo.state[p]
        ok = ok and st['step'] == 41 and isinstance(st['step'], int) and bool((st['exp_avg'] == 0.5 + i).all()) and bool((st['exp_avg_sq'] == 0.25 + i).all())
    out.append(ok)
    # (c) extra tensors through the hook (classifier stack of VGG16): launch() + wait_launched(), and through finish()
    hook = parallel.GradAllReduce(n_buckets=2)
    a, b, flat = torch.full((1000,), float(rank + 1)), torch.full((3, 5), float(10 * (rank + 1))), torch.full((64,), float(rank + 1))
    hook.launch(a)
    hook.wait_launched()
    hook.launch(b)
    hook.start(flat, 32, 64)
    hook.finish(flat)
    tot = sum(range(1, world + 1))
    out.append(bool((a == tot).all()) and bool((b == 10 * tot).all()) and bool((flat == tot).all()) and hook._works == [])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_optimizer_state_broadcast_when_only_rank0_has_state():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_opt_state_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert out == [True, True, True, True], (rank, out)


def test_init_from_env_sets_dmabuf_ipc_before_any_gpu_call(monkeypatch):
    """train.py under torch.distributed.run never went through bench.py's launcher, which was the only place that set
    HSA_ENABLE_IPC_MODE_LEGACY=0 (RCCL peer mappings need dmabuf IPC on this driver)."""
    import inspect
    from object_detection_torch2_amd import parallel
    src = inspect.getsource(parallel.init_from_env)
    assert src.index('HSA_ENABLE_IPC_MODE_LEGACY') < src.index('is_available') < src.index('init_process_group')
