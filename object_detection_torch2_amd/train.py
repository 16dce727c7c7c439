"""Training entry point with the reference's flags (src/train.py:15-27) on the MI355X path.

    python -m object_detection_torch2_amd.train --batch_size 32 --epochs 1 [--steps_per_epoch 50]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m object_detection_torch2_amd.train ...

The loop body is the reference's (train.py:103-125): zero_grad -> forward -> loss -> backward -> step, ExponentialLR per
epoch, weights.pth / params.json when the epoch loss improves.  Data: synthetic batches by default (no dataset ships with
this image); with --data_dirs the reference's pipeline (train.py:32-66) on the drop-in `dataset` / `augmentation` modules:
PIL decode + resize in DataLoader workers, colour jitter / flip / erasing in one device pass per batch.  Other deliberate
differences: torch.autograd.set_detect_anomaly is off, the optimizer is the fused HIP Adam, and under torchrun each rank
takes a contiguous shard of the global batch (synthetic) or a DistributedSampler shard (VOC) and gradients are
all-reduced with RCCL.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

from . import augmentation as aug, parallel, synth
from .dataset import PascalVOCDataset
from .utils import collate_fn
from .model import SSD
from .optim import FusedAdam


def build_model(device, world, weights_path=None, weights_path_vgg16=None, num_classes=20 + 1, imsize=300, dtype='fp32'):
    """train.py:76-83 + the data-parallel start condition: every rank holds rank 0's parameters and buffers.  SSD() draws its
    initial weights from the process's own (unseeded) RNG, so without the broadcast N ranks would train N different models.
    imsize 300 / 512 selects the prior layout (BASELINE configs[4]), dtype 'fp32' / 'bf16' the arithmetic of the convolutions
    (configs[3]: bf16 forward / backward on fp32 master parameters)."""
    net = SSD(num_classes=num_classes, weights_path=weights_path, weights_path_vgg16=weights_path_vgg16, imsize=imsize)
    net.set_compute_dtype(dtype)
    net.to(device)
    if world > 1:
        parallel.broadcast_module(net)
    return net


def end_of_epoch(net, rank, world, epoch, local_loss, min_loss, lr, weights_path, params_path, device=None):
    """train.py:141-152 under data parallelism.  `local_loss` is this rank's shard mean; the decision "improved?" is taken from
    the mean over ranks, which every rank computes identically -- so the collective inside (averaging the BatchNorm running
    statistics for the checkpoint) is entered by all ranks or by none.  Returns (global epoch loss, new min_loss)."""
    loss = parallel.global_mean(local_loss, device=device) if world > 1 else float(local_loss)
    if (min_loss is None) or (loss < min_loss):
        if world > 1:
            parallel.sync_bn_stats(net)
        if rank == 0:
            weights_path.parent.mkdir(parents=True, exist_ok=True)
            torch.save(net.state_dict(), weights_path)
            with open(params_path, 'w') as f:
                json.dump({'min_loss': loss, 'lr': lr, 'last_epoch': epoch}, f, indent=4)
        min_loss = loss
    return loss, min_loss


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--purpose', type=str, default='detection')
    parser.add_argument('--imsize', type=int, default=300)
    parser.add_argument('--batch_size', type=int, default=4)
    parser.add_argument('--epochs', type=int, default=1)
    parser.add_argument('--lr', type=float, default=0.001)
    parser.add_argument('--weight_decay', type=float, default=0.0005)
    parser.add_argument('--gamma', type=float, default=0.95)
    parser.add_argument('--num_workers', type=int, default=8)
    parser.add_argument('--result_dir', type=str, default='./result')
    parser.add_argument('--weights', type=str, default='weights.pth')
    parser.add_argument('--params', type=str, default='params.json')
    parser.add_argument('--synthetic', action='store_true', default=True, help='synthetic 300x300x3 batches (only data source here)')
    parser.add_argument('--steps_per_epoch', type=int, default=20, help='synthetic source only')
    parser.add_argument('--data_dirs', type=str, nargs='*', default=None, help='VOC roots, e.g. /work/data/VOCdevkit/VOC2007 (train.py:43)')
    parser.add_argument('--data_list', type=str, default='trainval.txt')
    parser.add_argument('--dtype', type=str, default='fp32', choices=('fp32', 'bf16'),
                        help='arithmetic of the convolutions: fp32 (the reference) or bf16 forward / backward on fp32 master parameters')
    parser.add_argument('--device-count', dest='device_count', type=int, default=None,
                        help='GPUs of this node to train on: without a launcher the script starts one rank per GPU itself '
                             '(under torch.distributed.run the launcher decides and this must match WORLD_SIZE)')
    parser.add_argument('--seed', type=int, default=None, help='seed of the initial weights (the reference draws them unseeded)')
    parser.add_argument('--graph_max_gt', type=int, default=64, help='--graph: ground-truth rows per image the captured step is built for')
    parser.add_argument('--graph', action='store_true',
                        help='replay the train step as one HIP graph (tail of step i beside the frozen backbone of step i+1)')
    args = parser.parse_args(argv)
    if args.purpose not in ('detection', 'classification'):
        raise SystemExit(f'--purpose must be detection or classification, got {args.purpose!r}')
    classify = args.purpose == 'classification'
    if args.device_count is not None and args.device_count > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher: start one fresh rank process per GPU from here, before this process touches the GPU (it never does)
        raise SystemExit(parallel.launch_local_ranks(args.device_count, [sys.executable, '-m', 'object_detection_torch2_amd.train']
                                                     + (sys.argv[1:] if argv is None else list(argv))))

    rank, world, local = parallel.init_from_env()
    if args.device_count is not None and args.device_count != world:
        raise SystemExit(f'--device-count {args.device_count} but WORLD_SIZE={world}')
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    weights_path = Path(args.result_dir) / args.purpose / args.weights
    params_path = Path(args.result_dir) / args.purpose / args.params

    if classify:
        # train.py:69-74.  The reference cannot run this branch (VGG16 has no train_params(), its loss has the wrong sign, its
        # Linear(512*7*7) does not fit the default --imsize 300: DESIGN.md); here it trains: NLL loss, --imsize 184..215.
        from .model import VGG16
        net = VGG16(num_classes=20, weights_path=weights_path, transfer_learning=True).unfreeze_features()   # 20-way stack, backbone trains
        net.to(device)
        if world > 1:
            parallel.broadcast_module(net)
            net.dropout_seed = rank          # one seed on every rank would drop the same units in every shard of the global batch (correlated masks)
        defaults = None
    else:
        net = build_model(device, world, weights_path, Path(args.result_dir) / 'classification' / args.weights, imsize=args.imsize,
                          dtype=args.dtype)
        defaults = net.default_bboxes.to(device)

    if params_path.exists():
        print('Params loaded.')
        with open(params_path, 'r') as f:
            params = json.load(f)
        min_loss, lr, start_epoch = params['min_loss'], params['lr'], params['last_epoch']
    else:
        min_loss, lr, start_epoch = None, args.lr, 0

    world_n = parallel.attach(net) if world > 1 else 1
    optimizer = FusedAdam(net.train_params(), lr=lr, weight_decay=args.weight_decay, grad_scale=1.0 / world_n,
                          on_step=net.hip_weights_changed)
    scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=args.gamma)
    lo, hi = parallel.shard_batch(args.batch_size * world, rank, world)      # --batch_size is per GPU

    loader = sampler = None
    if args.data_dirs:
        pipeline = aug.Compose([aug.RandomColorJitter(p=0.5), aug.RandomFlip(p=0.5), aug.ToTensor(), aug.RandomErasing(p=0.5, max_iter=3)])
        data = PascalVOCDataset(purpose=args.purpose, data_dirs=args.data_dirs, data_list_file_name=args.data_list, imsize=args.imsize,
                                transform=pipeline)
        sampler = torch.utils.data.distributed.DistributedSampler(data, world, rank, shuffle=True) if world > 1 else None
        loader = torch.utils.data.DataLoader(data, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler,
                                             num_workers=args.num_workers, collate_fn=collate_fn, drop_last=world > 1)

    def batches(epoch):
        if loader is not None:
            if sampler is not None:
                sampler.set_epoch(epoch)
            for images, gts in loader:
                yield images.to(device), gts.to(device)          # DeferredBatch.to = upload uint8 + device augmentation pass
            return
        for i in range(args.steps_per_epoch):
            seed = epoch * 100003 + i
            ims = torch.from_numpy(synth.images(seed, args.batch_size * world, args.imsize)[lo:hi]).to(device)
            if classify:
                labels = (synth.uniform(seed, 7, args.batch_size * world)[lo:hi] * 20).astype('int64')
                yield ims, torch.nn.functional.one_hot(torch.from_numpy(labels), 20).float().to(device)
            else:
                yield ims, torch.from_numpy(synth.targets(seed, args.batch_size * world)[lo:hi]).to(device)

    runner = None
    for epoch in range(1 + start_epoch, args.epochs + start_epoch + 1):
        running = torch.zeros((), device=device)
        t0 = time.perf_counter()
        n_steps = 0
        # one batch of lookahead: batch i+1 is uploaded (and augmented) between the loss and the backward of step i, and its
        # "ready" event lets the frozen backbone of step i+1 run next to step i's tail backward and optimizer (engine.py)
        source = batches(epoch)

        def fetch():
            item = next(source, None)
            return None if item is None else (item[0], item[1], torch.cuda.current_stream().record_event())

        cur = fetch()
        graphed = args.graph and not classify
        if graphed and cur is not None and runner is None:
            # one captured step for the whole run: batch shape of the first batch, ground truth padded to --graph_max_gt rows (zero rows
            # never match, ssd.py:231-250, so padding changes nothing); a ragged last batch falls back to the eager step below
            from .graph_step import GraphedTrainStep
            runner = GraphedTrainStep(net, optimizer, defaults, cur[0].shape, (cur[0].shape[0], max(args.graph_max_gt, cur[1].shape[1]), cur[1].shape[2]))
        fits = lambda item: item is not None and tuple(item[0].shape) == tuple(runner._x.shape) and item[1].shape[1] <= runner.G
        if graphed and fits(cur):
            runner.feed(cur[0], cur[2])
        while cur is not None:
            ims, gts, ready = cur
            n_steps += 1
            if graphed and fits(cur):
                cur = fetch()
                if fits(cur):
                    runner.feed(cur[0], cur[2])       # backbone of the next batch on its own stream, beside this batch's tail
                running += runner.step(gts)
                if not fits(cur):
                    runner.drain()                    # end of the epoch or an eager step next
                continue
            optimizer.zero_grad()
            if classify:
                outputs = net(ims)
                loss = net.loss(outputs=outputs, targets=gts, nll=True)
            else:
                outputs = net(ims, ready_event=ready)
                loss = net.loss(outputs=outputs, targets=gts, default_bboxes=defaults)
            cur = fetch()
            loss.backward()
            optimizer.step()
            running += loss.detach()
            if graphed and fits(cur):
                runner.feed(cur[0], cur[2])
        local_loss = running.item() / max(n_steps, 1)
        dt = time.perf_counter() - t0
        # every rank: same global loss, same branch (the checkpoint branch holds a collective)
        running_loss, min_loss = end_of_epoch(net, rank, world, epoch, local_loss, min_loss, scheduler.get_last_lr()[0],
                                              weights_path, params_path, device=device)
        if rank == 0:
            print(f'[Epoch {epoch}/{args.epochs + start_epoch}] loss: {round(running_loss, 5)} lr: {scheduler.get_last_lr()[0]:.6f} '
                  f'({n_steps * args.batch_size * world / dt:.1f} img/s incl. data loading)')
        scheduler.step()
    if rank == 0:
        print('Finished Training')


if __name__ == '__main__':
    main()
