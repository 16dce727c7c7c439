"""Evaluation entry point with the reference's flags (src/evaluate.py:75-81): forward -> decode -> score -> NMS ->
IoU vs GT + per-class greedy TP assignment (evaluate.py:128-151: one device pass, `ssdhip_eval_match`) -> AP on the host.

`calc_average_precision(..., reference_compatible=True)` reproduces the reference's AP exactly, including its quirk:
`torch.sort(result, dim=0)` sorts the TP column and the score column independently (evaluate.py:55), i.e. true positives
are ranked first regardless of score (README's mAP 0.314 is this number).  reference_compatible=False
ranks detections by score (standard VOC AP, all-point interpolation).  Data are synthetic batches (no VOC data here);
the runtime block of the report comes from rocm-smi instead of nvidia-smi (evaluate.py:164).
"""
import argparse
from datetime import date
from pathlib import Path
from subprocess import check_output

import torch

from . import synth
from .model import SSD
from .utils import LabelMap, calc_iou, decode_score_, non_maximum_suppression

def get_order(t: torch.Tensor, class_id: int) -> torch.Tensor:
    """evaluate.py:31-42: indices of rows with a positive score for class_id, by descending score."""
    vals, indices = torch.sort(t[:, 5 + class_id], descending=True)
    return indices[vals > 0.]


def assign_true_positives(iou_one_class: torch.Tensor) -> torch.Tensor:
    """evaluate.py:146-148: each prediction (rows, score order) claims its best-IoU GT if IoU > 0.5; only the first
    claimant of a GT is a true positive.  (X, G) -> (X, 1) of {0., 1.}."""
    g = iou_one_class.shape[1]
    best = iou_one_class.max(dim=1).indices
    valid = torch.eye(g, device=iou_one_class.device)[best] * (iou_one_class > 0.5)
    return ((valid.cumsum(dim=0) == valid) * valid).sum(dim=1, keepdim=True)


def calc_average_precision(result: torch.Tensor, count, reference_compatible: bool = True) -> torch.Tensor:
    """AP of one class from rows [is_tp, score] and the number of GT boxes (evaluate.py:45-67).
    Precision envelope: p*(k) = max_{j >= k} p(j); AP = sum_k p*(k) * (r(k) - r(k-1)), plus the reference's closing
    segment from the last recall to 1 at precision 0."""
    import numpy as np
    rows = result.detach().cpu().numpy().astype(np.float64)
    if reference_compatible:
        hits = np.sort(rows[:, 0])[::-1]            # the reference sorts each column on its own (Q11): TPs come first
    else:
        hits = rows[np.argsort(-rows[:, 1], kind='stable'), 0]
    tp = np.cumsum(hits == 1.0)
    fp = np.cumsum(hits == 0.0)
    precision = tp / np.maximum(tp + fp, 1)
    recall = tp / float(count)
    envelope = np.maximum.accumulate(precision[::-1])[::-1]
    steps = np.diff(np.concatenate([[0.0], recall]))
    return torch.tensor(float(np.sum(envelope * steps)), dtype=torch.float32)


def match_true_positives(outputs: torch.Tensor, gts: torch.Tensor, iou_thresh: float = 0.5, gt_count: torch.Tensor = None):
    """evaluate.py:132-151 for a whole batch in ONE device pass (`ssdhip_eval_match`): outputs (N,P,4+C) decoded, scored and
    NMS-filtered, gts (N,G,4+C) -> (cls (N,P) int32: class of the row's positive score or -1, tp (N,P) uint8, score (N,P)).
    gt_count (C-1,) int32 on the device, optional: += ground-truth boxes per class."""
    from . import _lib
    _lib.require_gpu(outputs, gts)
    outputs, gts = _lib.f32c(outputs), _lib.f32c(gts)
    if outputs.dim() != 3 or gts.dim() != 3 or outputs.shape[0] != gts.shape[0] or outputs.shape[2] != gts.shape[2] or outputs.shape[2] < 6:
        raise ValueError(f'match_true_positives: outputs {tuple(outputs.shape)} and gts {tuple(gts.shape)} must be (N, *, 4 + C), C >= 2')
    N, P, row = outputs.shape
    cls = torch.empty(N, P, dtype=torch.int32, device=outputs.device)
    tp = torch.empty(N, P, dtype=torch.uint8, device=outputs.device)
    score = torch.empty(N, P, dtype=torch.float32, device=outputs.device)
    _lib.check(_lib.load().ssdhip_eval_match(_lib.ptr(outputs), _lib.ptr(gts), N, P, gts.shape[1], row - 4, float(iou_thresh), _lib.ptr(cls),
                                             _lib.ptr(tp), _lib.ptr(score), _lib.ptr(gt_count) if gt_count is not None else None,
                                             _lib.stream()), 'eval_match')
    return cls, tp, score


def evaluate_batches(net, defaults, batches, num_classes=20):
    """Accumulate per-class [is_tp, score] rows and GT counts over (images, gts) device batches (evaluate.py:117-151): forward,
    decode, score, NMS and the true-positive assignment run on the device; what comes back per batch is (class, tp, score) per
    prior, grouped per class on the host like the reference's result_correct / result_count."""
    import numpy as np
    result_correct = {c: [] for c in range(num_classes)}
    gt_count = None
    with torch.no_grad():
        for images, gts in batches:
            outputs = net(images)
            decode_score_(outputs, defaults)
            outputs = non_maximum_suppression(outputs=outputs)
            if gt_count is None:
                gt_count = torch.zeros(outputs.shape[2] - 5, dtype=torch.int32, device=outputs.device)
            cls, tp, score = (t.cpu().numpy() for t in match_true_positives(outputs, gts, 0.5, gt_count))
            for n in range(cls.shape[0]):                        # image order, then class, then descending score: the reference's row order
                for c in np.unique(cls[n][cls[n] >= 0]):
                    if c >= num_classes:
                        continue
                    rows = np.nonzero(cls[n] == c)[0]
                    rows = rows[np.argsort(-score[n, rows], kind='stable')]
                    result_correct[int(c)].append(torch.from_numpy(np.stack([tp[n, rows].astype(np.float32), score[n, rows]], axis=1)))
    counts = gt_count.cpu().tolist() if gt_count is not None else [0] * num_classes
    return result_correct, {c: int(counts[c]) if c < len(counts) else 0 for c in range(num_classes)}


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--imsize', type=int, default=300)
    parser.add_argument('--batch_size', type=int, default=2)
    parser.add_argument('--num_workers', type=int, default=8)
    parser.add_argument('--result_dir', type=str, default='./result')
    parser.add_argument('--weights', type=str, default='weights.pth')
    parser.add_argument('--dtype', type=str, default='fp32', choices=('fp32', 'bf16'), help='arithmetic of the convolutions')
    parser.add_argument('--batches', type=int, default=2, help='synthetic source only')
    parser.add_argument('--data_dirs', type=str, nargs='*', default=None, help='VOC roots (evaluate.py:96: VOC2007 test)')
    parser.add_argument('--data_list', type=str, default='test.txt')
    args = parser.parse_args(argv)

    out_dir = Path(args.result_dir) / 'detection'
    out_dir.mkdir(parents=True, exist_ok=True)
    device = torch.device('cuda:0')
    net = SSD(num_classes=20 + 1, weights_path=out_dir / args.weights,
              weights_path_vgg16=Path(args.result_dir) / 'classification' / args.weights, imsize=args.imsize)
    net.set_compute_dtype(args.dtype)
    net.to(device)
    defaults = net.default_bboxes.to(device)
    labelmap = LabelMap('PascalVOC')

    def batches():
        if args.data_dirs:                                # evaluate.py:91-102: ToTensor only, batches through collate_fn
            from . import augmentation as aug
            from .dataset import PascalVOCDataset
            from .utils import collate_fn
            data = PascalVOCDataset('detection', args.data_dirs, args.data_list, args.imsize, transform=aug.Compose([aug.ToTensor()]))
            for images, gts in torch.utils.data.DataLoader(data, batch_size=args.batch_size, num_workers=args.num_workers, collate_fn=collate_fn):
                yield images.to(device), gts.to(device)
            return
        for b in range(args.batches):
            yield (torch.from_numpy(synth.images(3000 + b, args.batch_size, args.imsize)).to(device),
                   torch.from_numpy(synth.targets(4000 + b, args.batch_size)).to(device))

    correct, count = evaluate_batches(net, defaults, batches())
    lines = [f'# Evaluation report {date.today().isoformat()}', '', '## Runtime', '```']
    try:
        lines.append(check_output(['rocm-smi', '--showproductname', '--showmeminfo', 'vram']).decode().strip())
    except Exception as e:   # noqa: BLE001  (rocm-smi missing or not permitted)
        lines.append(f'rocm-smi unavailable: {e}')
    lines += ['```', '', '## Config', '', '|item|value|', '|-|-|'] + [f'|{k}|{v}|' for k, v in vars(args).items()]
    lines += ['', '## Scores', '', '|label|AP, reference definition (TPs ranked first)|AP, score-ranked (VOC)|', '|-|-|-|']
    aps = []
    for c in range(20):
        if correct[c] and count[c] > 0:
            rows = torch.cat(correct[c]).cpu()
            pair = (calc_average_precision(rows, count[c], True).item(), calc_average_precision(rows, count[c], False).item())
        else:
            pair = (0.0, 0.0)
        aps.append(pair)
        lines.append(f'|{labelmap.id2name(c)}|{pair[0]:.3f}|{pair[1]:.3f}|')
    mean = (sum(a for a, _ in aps) / 20, sum(b for _, b in aps) / 20)
    lines.append(f'|**mean**|**{mean[0]:.3f}**|**{mean[1]:.3f}**|')
    report = out_dir / f'report_{date.today().isoformat()}.md'
    report.write_text('\n'.join(lines) + '\n')
    print(lines[-1])
    print('Finished Evaluate')


if __name__ == '__main__':
    main()
