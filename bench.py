#!/usr/bin/env python3
"""Headline benchmark: images/sec of one SSD300 train step (zero_grad + forward + MultiBox loss + backward + Adam),
batch 32 per GPU, fp32, synthetic 300x300x3 batches (BASELINE.json metric; workload = configs[2], at N>1 the weak-
scaled data-parallel form with an RCCL gradient all-reduce).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Either torch.distributed.run starts the ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment), or -- when WORLD_SIZE is not set -- this process starts N fresh child processes itself before it touches the
GPU (never by replacing itself), waits for them and exits non-zero if any of them failed.

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around the launches of the dominant kernel
over extra steps run right behind the timed region (the events cost host time the bf16 step cannot hide); `cpu_baseline` times the CPU oracle (a port of the reference path) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE_TRAIN = 72.238     # fwd 62.747 + wgrad 7.114 + dgrad 2.376 (VGG frozen)
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def host_cores():
    from object_detection_torch2_amd.parallel import host_cores as hc
    return hc()


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(batch=32, warmup=3, steps=5, budget_s=120.0):
    """the oracle (a port of the reference path on stock torch CPU ops + numpy) on this host's cores, batch 32,
    fp32, 3 warm-up + 5 timed iterations of the train step (zero_grad + forward + loss + backward + Adam) and of the inference
    step (forward + decode + score + NMS on void-biased logits).  Bounded: each leg stops warming up / timing early when its
    share of `budget_s` is spent (slow hosts), and says in `sample` how many iterations it really ran."""
    import torch
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import synth
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = O.init_state(seed=1)
    df = O.default_bboxes()
    ims = torch.from_numpy(synth.images(100, batch))
    tg = synth.targets(200, batch, g_max=8)

    def timed(fn, budget):
        t_start = time.perf_counter()
        w = 0
        while w < warmup and (w == 0 or time.perf_counter() - t_start < 0.3 * budget):
            fn(); w += 1
        t0 = time.perf_counter()
        k = 0
        while k < steps and (k == 0 or time.perf_counter() - t_start < budget):
            fn(); k += 1
        return w, k, time.perf_counter() - t0

    st = {}
    wt, kt, dt_t = timed(lambda: O.train_step(sd, st, ims, tg, df), budget_s * 0.55)

    def infer():
        with torch.no_grad():
            out = O.forward(sd, ims, training=True).numpy()        # the reference never calls eval() (Q3)
        out[:, :, 4] += 3.0                                        # void-biased logits, ~1.3 k NMS candidates per image
        out[:, :, :4] = O.calc_coordicate(out, df)
        out[:, :, 4:] = O.calc_score(out)
        O.non_maximum_suppression(out)

    wi, ki, dt_i = timed(infer, budget_s * 0.45)
    return dict(value=round(batch * kt / dt_t, 3), unit='images/sec', cores=cores, kind='port',
                inference_value=round(batch * ki / dt_i, 3), cpu=cpu_model(),
                sample=f'train step: {wt} warm-up + {kt} timed at batch {batch}; inference step (forward + decode + score + NMS, '
                       f'~1.3k candidates/image): {wi} warm-up + {ki} timed at batch {batch}; fp32, torch {torch.__version__} CPU ops, '
                       f'{cores} threads; BASELINE.md §3 asks 3 + 5, fewer only where the {budget_s:.0f} s budget ran out')


def step_gflop(eng, imsize):
    """Direct-convolution GFLOP per image (forward, frozen-backbone train step) from the engine's layer table, the way
    SURVEY.md 8(d) counts them: 2*Ho*Wo*Cout*Cin*k^2 per conv; + wgrad of every trainable conv; + dgrad of every trainable
    conv whose input carries a gradient (not the first trainable conv, not the head on the frozen conv_4_3)."""
    fwd = wgrad = dgrad = 0.0
    geo = eng._geometry(imsize, imsize)
    for i, (s, g) in enumerate(zip(eng.stages, geo)):
        f = 2.0 * g['ho'] * g['wo'] * s.conv.cout * s.conv.cin * s.conv.k ** 2
        fwd += f
        if not s.conv.frozen:
            wgrad += f
            dgrad += f if i > eng.tail_start else 0.0
        if s.head is not None:
            fh = 2.0 * g['ho'] * g['wo'] * s.head.cout * s.head.cin * 9
            fwd += fh; wgrad += fh
            dgrad += fh if i >= eng.tail_start else 0.0
    return fwd / 1e9, (fwd + wgrad + dgrad) / 1e9


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes started from this process, which has not touched the
    GPU and never does (parallel.launch_local_ranks)."""
    from object_detection_torch2_amd.parallel import launch_local_ranks
    return launch_local_ranks(n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])


def dry_run(args):
    """Plumbing rehearsal without a GPU (tests/test_dist_cpu.py): rendezvous, the bench's barrier / max-over-ranks timing and the
    bucketed all-reduce of a gradient-sized flat buffer over gloo -- no kernels, so its JSON line is labelled and is no measurement."""
    import torch
    import torch.distributed as dist
    from object_detection_torch2_amd import parallel
    rank, world, _ = parallel.init_from_env(backend='gloo')
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if os.environ.get('SSD_BENCH_TEST_FAIL_RANK') == str(rank):
        raise SystemExit(7)
    hook = parallel.GradAllReduce(n_buckets=2)
    flat = torch.full((1 << 16,), float(rank + 1))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        flat.fill_(float(rank + 1))
        hook.start(flat, 1 << 15, flat.numel())
        hook.finish(flat)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ok = bool((flat == world * (world + 1) / 2).all())
    if rank == 0:
        print(json.dumps(dict(metric='DRY RUN (no GPU work): launcher + collective plumbing only', value=None, n_gpus=world, ranks=world,
                              steps=args.steps, ms_per_step=round(dt / max(1, args.steps) * 1e3, 3), allreduce_ok=ok, dry_run=True)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)       # warm-up 10 steps, time >= 50 steps (0.75 s on one MI355X)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--imsize', type=int, choices=[300, 512], default=300,
                    help='300 = the reference network (headline); 512 = SSD512 extension (BASELINE configs[4] is its bf16 8-GPU form; run with --batch 16)')
    ap.add_argument('--dtype', choices=['fp32', 'bf16'], default='fp32',
                    help='fp32 = the headline (BASELINE configs[2]); bf16 = conv forward/dgrad on the bf16 MFMA with fp32 accumulate and fp32 master\n'
                         'state (the single-GPU form of configs[3]); labelled, never the headline')
    ap.add_argument('--finetune-backbone', action='store_true',
                    help='extension (SURVEY 8(f)-4): un-freeze VGG16-bn (187.9 GFLOP/img instead of 72.2); never the headline')
    ap.add_argument('--mode', choices=['train', 'infer'], default='train',
                    help='train (headline, BASELINE configs[2]) or infer (configs[1]: forward + decode + score + NMS)')
    ap.add_argument('--graph', action='store_true',
                    help='replay the captured step (graph_step.py: host enqueue 3.6 -> 0.6 ms per step) instead of enqueueing every launch from '
                         'Python; opt-in because the eager engine overlaps better on the device (9.7 vs 10.4 ms per fp32 step, DESIGN.md 7a)')
    ap.add_argument('--allreduce', choices=['allreduce', 'rs_ag'], default=os.environ.get('SSD_ALLREDUCE', 'allreduce'),
                    help='N > 1: gradient exchange as all-reduce buckets or as reduce-scatter + all-gather (SURVEY 8(e))')
    ap.add_argument('--grad-dtype', choices=['fp32', 'bf16'], default=os.environ.get('SSD_GRAD_DTYPE', 'fp32'),
                    help='N > 1: payload of the gradient exchange (bf16 halves the bytes; the sum is taken in bf16)')
    ap.add_argument('--dry-run', action='store_true', help='no GPU: rehearse the launcher, rendezvous and all-reduce over gloo (CPU tests)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    if args.dry_run:
        return dry_run(args)

    import numpy as np  # noqa: F401
    import torch
    from object_detection_torch2_amd import parallel, synth
    from object_detection_torch2_amd.model import SSD
    from object_detection_torch2_amd.optim import FusedAdam
    import torch.distributed as dist

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    torch.manual_seed(1)
    net = SSD(num_classes=21, imsize=args.imsize).to(dev)   # random-init weights (no pretrained weights offline)
    if args.finetune_backbone:
        net.finetune_backbone()
    net.set_compute_dtype(args.dtype)
    if world > 1:
        parallel.broadcast_module(net)      # replicas start from rank 0's weights whatever each process's RNG state was
    df = net.default_bboxes.to(dev)
    world_n = parallel.attach(net, mode=args.allreduce, payload=args.grad_dtype) if world > 1 else 1
    opt = FusedAdam(net.train_params(), lr=1e-3, weight_decay=5e-4, grad_scale=1.0 / world_n, on_step=net.hip_weights_changed)
    B = args.batch
    ims = torch.from_numpy(synth.images(100 + rank, B, args.imsize)).to(dev)       # this rank's shard of the global batch
    tgs = torch.from_numpy(synth.targets(200 + rank, B, g_max=8)).to(dev)
    running = torch.zeros((), device=dev)
    torch.cuda.synchronize()
    resident = torch.cuda.current_stream().record_event()   # the synthetic batch is resident in HBM from here on (bench contract):
    # passing this event lets the frozen backbone of step i+1 start while the tail backward / Adam of step i still run

    from object_detection_torch2_amd import utils as hip_utils
    void_bias = torch.zeros(1, 1, 25, device=dev)
    void_bias[0, 0, 4] = 3.0           # void-biased logits, ~1.3k NMS candidates per image (trained-net regime)

    def infer_step():
        with torch.no_grad():
            out = net(ims, ready_event=resident)
            out += void_bias
            hip_utils.decode_score_(out, df)
            hip_utils.non_maximum_suppression(out)
        running.add_(out[0, 0, 0])

    def eager_step():
        if args.mode == 'infer':
            return infer_step()
        opt.zero_grad()
        out = net(ims, ready_event=resident)
        loss = net.loss(out, tgs, df)
        loss.backward()
        opt.step()
        running.add_(loss.detach())

    runner, step_mode = None, 'eager: every launch enqueued from Python'
    use_graph = (args.mode == 'train' and (args.graph or os.environ.get('SSD_GRAPH') == '1') and not args.finetune_backbone
                 and os.environ.get('SSD_SINGLE_STREAM') != '1')

    def step():
        if runner is None:
            return eager_step()
        runner.feed(ims, resident)             # frozen backbone of the NEXT step on its own stream ...
        running.add_(runner.step(tgs))         # ... beside this step's tail forward + loss + backward + Adam

    # the caller's stream carries the tail (small dependent launches); run it above the backbone stream's priority
    main_stream = torch.cuda.Stream(device=dev, priority=-1) if os.environ.get('SSD_STREAM_PRIO', '1') == '1' else torch.cuda.current_stream()
    main_stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(main_stream)
    eng = net._engine()
    if use_graph:
        # a few eager steps first (Adam moments exist, allocator warm), then the captured step; a capture that fails for any reason
        # leaves the eager path (and says so in the line)
        for _ in range(min(2, args.warmup)):
            eager_step()
        try:
            from object_detection_torch2_amd.graph_step import GraphedTrainStep
            runner = GraphedTrainStep(net, opt, df, ims.shape, tgs.shape)
            runner.feed(ims, resident)         # prime the pipeline: from here on every step() pairs one backbone pass with one tail
            step_mode = ('HIP graphs: backbone(i+1) [1 graph, own stream] beside tail(i) = heads + trainable layers + loss + backward + Adam '
                         f'[{sum(1 for it in runner._tail_items[0] if not isinstance(it, tuple))} graph segment(s)'
                         + (', RCCL collectives eager between them]' if world > 1 else ']'))
        except Exception as e:   # noqa: BLE001
            runner = None
            step_mode = f'eager (graph capture failed: {type(e).__name__}: {e})'
            print(f'bench.py: {step_mode}', file=sys.stderr)
    pre = min(2, args.warmup) if use_graph else 0      # the eager steps in front of the capture count as warm-up steps
    for _ in range(args.warmup - pre):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_enqueue_ms = (time.perf_counter() - t0) / max(1, args.steps) * 1e3      # host time to enqueue a step (no device wait inside the loop)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = running.item() / max(1, args.steps + args.warmup)
    if runner is not None:
        running.add_(runner.step(tgs))     # the batch fed ahead of the last timed step
        runner.drain()
        graph_runner, runner = runner, None    # the roofline and communication legs below bracket single launches: eager steps
    if not args.no_roofline:
        # The roofline leg runs on EXTRA steps right behind the timed region, never inside it: creating and recording an event pair
        # per large launch costs the host ~1 ms per step, which the bf16 step (4.8 ms of device time for 3.5 ms of enqueueing)
        # cannot hide (5.7 ms with the events inside the timed region).  First five more steps with the three streams running as in
        # the timed region, then the serialised steps below.
        eng.prof = []
        for _ in range(5):
            step()
        torch.cuda.synchronize()

    def aggregate(records):
        agg = {}
        for r in records:
            a = agg.setdefault(r['kernel'], dict(ms=0.0, flops=0.0, n=0, layers=set(), direct=0.0, padded=0.0, peak=r.get('peak_tflops', FP32_MFMA_PEAK_TFLOPS)))
            # `flops` = the ALGORITHMIC flops of the launch (for a Winograd-domain GEMM: 2 * planes * tiles * Cin * Cout with the real tile and
            # channel counts); `padded` = what the launch executes including tile / channel padding (det_4_3: 100 -> 128 channels)
            a['ms'] += r['e0'].elapsed_time(r['e1']); a['flops'] += r.get('flops_unpadded', r['flops']); a['padded'] += r['flops']
            a['n'] += 1; a['layers'].add(r['layer'])
            a['direct'] += r.get('direct_flops', r['flops'])
        return agg

    roof = None
    if eng.prof:
        agg = aggregate(eng.prof)
        name, a = max(agg.items(), key=lambda kv: kv[1]['ms'])
        ach = a['flops'] / (a['ms'] * 1e-3) / 1e12
        traffic, traffic_source = None, None   # HBM-side bytes per launch: NOT measured in this run (PMC passes need rocprofv3 around the
        for pmc_file in ('r04_pmc_summary.json', 'r03_pmc_summary.json'):     # process): read from the committed summary of scripts/collect_pmc.sh
            try:
                pmc = json.load(open(os.path.join(ROOT, 'profiles', pmc_file)))
                traffic = round(pmc[name]['hbm_bytes_per_launch'])
                traffic_source = (f'profiles/{pmc_file} (separate rocprofv3 --pmc passes of this workload: FETCH_SIZE x2 per the gfx950 '
                                  'correction + WRITE_SIZE; committed file, not collected by this run)')
                break
            except (OSError, KeyError, ValueError):
                pass
        notes = []
        if a['direct'] > a['flops'] * 1.01:
            notes.append('achieved/frac count the MFMA flops this kernel executes; the Winograd-domain GEMMs stand for '
                         f"{a['direct'] / a['flops']:.2f}x as many direct-convolution flops")
        peak = a['peak']          # fp32 MFMA peak, or the dense bf16 MFMA peak for the bf16 patch convolution (MI355X_MICROARCH.md)
        roof = dict(bound='mfma', kernel=name, achieved=round(ach, 2), peak=peak, unit='TFLOP/s',
                    frac=round(ach / peak, 4), traffic=traffic, traffic_source=traffic_source, launches=a['n'],
                    algorithmic_flop_per_launch=round(a['flops'] / a['n']), executed_flop_per_launch_incl_padding=round(a['padded'] / a['n']),
                    avg_launch_ms=round(a['ms'] / a['n'], 4), layers=sorted(a['layers']),
                    conv_fwd_tflops_launches_over_10_gflop=round(sum(v['flops'] for v in agg.values()) / (sum(v['ms'] for v in agg.values()) * 1e-3) / 1e12, 2))
        if not eng.single_stream:
            # Run like the timed region, the backbone stream shares the chip with two other streams, so an event pair on it measures a
            # time-share (1.27 ms per launch for a kernel whose own duration, rocprofv3 --kernel-trace of this same command,
            # profiles/r02_o_*, is 1.11 ms).  So the figure that rocprofv3 can confirm is taken over three more steps with the
            # streams serialised (same launches, same grid, same order per stream); the figures of the five pipelined extra steps
            # stay in the line as `timed_region` (they are taken behind it, run the same way).
            frozen_grid = eng._wf_blocks_env
            eng._wf_blocks_env = str(eng._wf_blocks)      # same grid as in the timed region (224 of 256 CUs beside the tail's streams)
            eng.single_stream = True
            step()                           # one unrecorded step to leave the pipelined regime
            torch.cuda.synchronize()
            eng.prof = []
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            ex = aggregate(eng.prof).get(name)
            eng.single_stream = False
            eng._wf_blocks_env = frozen_grid
            if ex:
                ach_x = ex['flops'] / (ex['ms'] * 1e-3) / 1e12
                roof['timed_region'] = dict(achieved=roof['achieved'], frac=roof['frac'], avg_launch_ms=roof['avg_launch_ms'], launches=roof['launches'],
                                            what='HIP events around the launches of 5 extra steps run like the timed region (three streams sharing the chip: a time-share, not the kernel alone)')
                roof.update(achieved=round(ach_x, 2), frac=round(ach_x / peak, 4), avg_launch_ms=round(ex['ms'] / ex['n'], 4), launches=ex['n'],
                            measured='HIP events around the same launches over 3 extra steps (after one unrecorded step) right after the timed region, streams serialised '
                                     '(agrees with the per-dispatch durations rocprofv3 reports for this command)')
        if a['direct'] > a['flops'] * 1.01:
            # SURVEY 8(d) counts a layer's work as direct-convolution flops; the same launches priced that way (kept beside the
            # executed-flop figure above, which is the one to judge the kernel's use of the matrix pipe by)
            dq = roof['achieved'] * a['direct'] / a['flops']
            roof['direct_equivalent'] = dict(achieved=round(dq, 2), frac=round(dq / peak, 4),
                                             algorithmic_flop_per_launch=round(a['direct'] / a['n']))
        if name.startswith('wino_fused8_kernel') and getattr(eng, '_wf_blocks', 0):
            # the persistent backbone kernels take 224 of the 256 CUs while the tail's streams run beside them (0.15-0.2 ms per step);
            # `frac` stays priced against the whole chip's peak, this field prices the same launches against the CUs they occupy
            roof['cus_used'] = eng._wf_blocks
            roof['frac_of_cus_used'] = round(roof['frac'] * 256.0 / eng._wf_blocks, 4)
            notes.append(f'the kernel is launched on {eng._wf_blocks} of the 256 CUs (the rest is left to the concurrent tail streams)')
        if name.startswith('gemm_planes_queue_kernel') and getattr(eng, '_pq_cus', 0):
            # opt-in SSD_PQ_CUS: the persistent work-queue GEMM on fewer CUs beside the tail's streams; `frac` stays priced against the
            # whole chip's peak, this field prices the same launches against the CUs they occupy
            roof['cus_used'] = eng._pq_cus
            roof['frac_of_cus_used'] = round(roof['frac'] * float(eng._cus) / eng._pq_cus, 4)
            notes.append(f'SSD_PQ_CUS: the kernel is launched on {eng._pq_cus} of the {eng._cus} CUs (the rest is left to the concurrent tail streams)')
        roof['note'] = '; '.join(notes) if notes else None
    eng.prof = None

    comm = None
    if world > 1 and args.mode == 'train':
        # (a) the step's collective alone: the same bucket pattern on a scratch buffer of the gradient's size, nothing else running;
        # (b) what the step pays for it: the same steps with the hook detached (weights diverge across ranks -- after the measurement)
        hook, flat = eng.dist_hook, eng._flat['buf']
        scratch = torch.zeros_like(flat)
        cut = eng.grad_cut()
        reps = 10
        for _ in range(2):
            hook.start(scratch, cut, scratch.numel()); hook.finish(scratch)
        dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hook.start(scratch, cut, scratch.numel()); hook.finish(scratch)
        e1.record()
        torch.cuda.synchronize()
        alone_ms = e0.elapsed_time(e1) / reps
        hook_desc = hook.describe()
        eng.dist_hook = None
        k = max(1, min(args.steps, 10))
        step()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([(time.perf_counter() - t1) / k * 1e3, alone_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        eng.dist_hook = hook
        comm = dict(alone_ms=round(t[1].item(), 3), step_ms_without=round(t[0].item(), 3), bytes=flat.numel() * 4, describe=hook_desc)

    gf_train, gf_fwd = GFLOP_PER_IMAGE_TRAIN, 62.747
    if args.imsize != 300 or args.finetune_backbone:
        gf_fwd, gf_train = step_gflop(eng, args.imsize)
    if rank == 0:
        value = world * B * args.steps / dt
        metric = 'images/sec SSD300 train step, batch 32/GPU' if args.mode == 'train' else \
            'images/sec SSD300 inference (forward + decode + score + NMS), batch 32/GPU'
        if B != 32 and args.imsize == 300:
            metric = metric.replace('batch 32/GPU', f'batch {B}/GPU (not the BASELINE batch of 32)')
        line = dict(metric=metric, value=round(value, 2), unit='images/sec', n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True,
                    scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                    config=dict(workload=f'SSD300 training batch={B}/GPU, fp32, IoU-match thresh=0.25, hard-neg 3:1, Adam '
                                         '(BASELINE configs[2]' + (f'; data-parallel, {"RCCL" if dist.get_backend() == "nccl" else dist.get_backend()} grad all-reduce)' if world > 1 else ')'),
                                global_batch=world * B, image='300x300x3', priors=8732, classes=21, parallelism=f'dp{world}',
                                bn='train-mode batch statistics (reference never calls eval())', backbone='frozen VGG16-bn',
                                conv_algorithms=('fp32 MFMA throughout; 3x3/s1/p1 layers: fused Winograd F(2x2,3x3) (conv_1_2 .. conv_3_1), three-pass '
                                                 'Winograd F(4x4,3x3) (conv_3_2 .. conv_6_1, heads on >= 16-px maps: forward, weight and data gradients)'
                                                 if eng._wino_mode == 'f4' else
                                                 'fp32 MFMA throughout; 3x3/s1/p1 layers: Winograd F(2x2,3x3) everywhere (SSD_WINOGRAD=f2)')
                                                 if args.dtype == 'fp32' else 'see precision'),
                    direct_equiv_vs_fp32_peak=round(value / world * (gf_train if args.mode == 'train' else gf_fwd)
                                                    / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4),
                    direct_equiv_note=('NOT a roofline fraction: direct-convolution flops of SURVEY 8(d) per second over the fp32 MFMA peak; it exceeds 1 because the '
                                       'Winograd layers execute 2.25x (F(2x2)) to 4x (F(4x4)) fewer MFMA flops than that count; `roofline` prices the flops a kernel executes'),
                    step_mode=step_mode, host_enqueue_ms_per_step=round(host_enqueue_ms, 3),
                    mean_loss=round(final_loss, 4), ranks=(dist.get_world_size() if world > 1 else 1))
        if comm is not None:
            # comm_ms_per_step: the gradient all-reduce of one step by itself (HIP events, max over ranks); the step hides most of
            # it behind the last weight gradients and the next step's backbone -- comm_exposed_ms_per_step is what is left
            line['comm_ms_per_step'] = comm['alone_ms']
            line['comm_exposed_ms_per_step'] = round(dt / args.steps * 1e3 - comm['step_ms_without'], 3)
            d = comm['describe']
            line['rccl_ranks_seen'] = d['ranks_seen']           # from the communicator, not from the command line
            line['comm'] = dict(collective=('all-reduce(sum)' if d['mode'] == 'allreduce' else 'reduce-scatter + all-gather (in place)')
                                + f" of the flat gradient buffer, {d['payload']} payload, {d['buckets_early']} early bucket(s) + 1 late",
                                bytes=comm['bytes'], backend=d['backend'], mode=d['mode'], payload=d['payload'],
                                per_collective=[dict(kind=k, elements=n, wire_bytes_per_rank=b) for k, n, b in d['collectives_last_step']],
                                ms_per_step_without_allreduce=comm['step_ms_without'])
        else:
            line['comm_ms_per_step'] = 0.0
        if roof is not None:
            line['roofline'] = roof
        if args.imsize != 300:      # extension variant, never the headline: say so in every field a reader may look at
            P = net.topo.num_priors()
            line['metric'] = f'images/sec SSD{args.imsize} {"train step" if args.mode == "train" else "inference"}, batch {B}/GPU (extension variant, not the BASELINE metric)'
            line['config'].update(workload=f'SSD{args.imsize} {args.mode} batch={B}/GPU, {args.dtype} ({P} priors; BASELINE configs[4] names its bf16 8-GPU form)',
                                  image=f'{args.imsize}x{args.imsize}x3', priors=P, gflop_per_image=round(gf_train if args.mode == 'train' else gf_fwd, 3))
        if args.dtype == 'bf16':
            line['dtype'] = 'bf16'
            line['config']['workload'] = line['config']['workload'].replace(', fp32,', ', bf16 fwd/bwd + fp32 master,').replace('BASELINE configs[2]', 'BASELINE configs[3] on one GPU')
            line['metric'] += ' (bf16 MFMA compute with fp32 accumulate / fp32 master state: BASELINE configs[3] form, not the fp32 headline)'
            line['config']['precision'] = ('frozen backbone: bf16 activations in HBM (zero-haloed), bf16 patch convolution with fp32 accumulate; trainable tail: '
                                           'conv forward, data and weight gradients with bf16 operands (rounded at LDS staging) and fp32 accumulate; '
                                           'master weights, BN statistics, gradients in HBM, loss, Adam: fp32')
        if args.finetune_backbone:
            line['metric'] += ' (extension: backbone un-frozen, not the BASELINE metric)'
            line['config'].update(backbone='VGG16-bn un-frozen (full fine-tuning)', gflop_per_image=round(gf_train, 3))
        if args.imsize == 300 and args.mode == 'infer':
            line['config']['workload'] = 'SSD300 inference batch=32, fp32, NMS on 8732 priors x 20 classes (BASELINE configs[1])'
        if world == 1 and not args.no_cpu_baseline and args.mode == 'train' and args.imsize == 300 and not args.finetune_backbone and args.dtype == 'fp32':
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
