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
    q.put((rank, lo, hi, ranges, ok_sum, ok_avg, ok_bn))
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
        assert r[4] and r[5] and r[6]


def test_shard_batch_rejects_ragged():
    from object_detection_torch2_amd import parallel
    with pytest.raises(ValueError):
        parallel.shard_batch(65, 0, 2)
