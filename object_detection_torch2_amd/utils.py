"""Drop-in for the reference's `utils` module (src/utils.py): same names (incl. the `coordicate` spelling), argument
order, return shapes and the in-place mutation of `outputs` by NMS -- executed by libssdhip.so on MI355X."""
import json
from pathlib import Path

import torch
from torch.nn.utils.rnn import pad_sequence

from . import _lib, torch_ops


def collate_fn(batch):
    """utils.py:8-16: stack images, zero-pad GT rows to Gmax (host-side data contract, no kernel involved).  Images that
    come out of the augmentation pipeline as `Deferred` (uint8 + pending device work) are stacked into a `DeferredBatch`,
    whose `.to(device)` yields the augmented fp32 batch."""
    from .augmentation import Deferred, DeferredBatch
    if isinstance(batch[0][0], Deferred):
        images = DeferredBatch.stack([im for im, _ in batch])
    else:
        images = torch.stack([im for im, _ in batch], dim=0)
    gts = pad_sequence([gt for _, gt in batch], batch_first=True)
    return images, gts


def _decode_score_copy(pr: torch.Tensor, df: torch.Tensor) -> torch.Tensor:
    _lib.require_gpu(pr, df)
    if pr.dim() != 3 or pr.shape[2] < 5:
        raise ValueError(f'expected (N, P, 4 + C) predictions, got {tuple(pr.shape)}')
    out = _lib.f32c(pr).clone()
    decode_score_(out, df)
    return out


def decode_score_(outputs: torch.Tensor, df: torch.Tensor) -> torch.Tensor:
    """Fused utils.calc_coordicate + utils.calc_score, in place on a contiguous fp32 (N,P,4+C) tensor
    (inference.py:67-68 / evaluate.py:129-130 as one pass)."""
    _lib.require_gpu(outputs, df)
    if not (outputs.is_contiguous() and outputs.dtype == torch.float32):
        raise ValueError('decode_score_: outputs must be contiguous fp32')
    N, P, row = outputs.shape
    df = _lib.f32c(df)
    if tuple(df.shape) != (P, 4):
        raise ValueError(f'default boxes must be ({P}, 4), got {tuple(df.shape)}')
    torch_ops.load().decode_score_(outputs, df)
    return outputs


def calc_coordicate(pr: torch.Tensor, df: torch.Tensor) -> torch.Tensor:
    """utils.py:19-40: (N,P,>=4),(P,4) -> (N,P,4) decoded (cx,cy,w,h)."""
    if pr.shape[2] == 4:   # location-only input: give the fused kernel dummy class columns
        pr = torch.cat([pr, pr.new_zeros(pr.shape[0], pr.shape[1], 2)], dim=2)
    return _decode_score_copy(pr, df)[:, :, :4]


def calc_score(pr: torch.Tensor) -> torch.Tensor:
    """utils.py:43-55: (N,P,4+C) -> (N,P,C): softmax * onehot(argmax) (exactly one non-zero per row)."""
    df = torch.zeros(pr.shape[1], 4, device=pr.device)
    df[:, 2:] = 1
    return _decode_score_copy(pr, df)[:, :, 4:]


def calc_iou(t: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """utils.py:58-77: (N,T,>=4),(N,S,>=4) -> (N,T,S) centre-form IoU."""
    _lib.require_gpu(t, s)
    t = _lib.f32c(t)
    s = _lib.f32c(s)
    if t.dim() != 3 or s.dim() != 3 or t.shape[0] != s.shape[0] or t.shape[2] < 4 or s.shape[2] < 4:
        raise ValueError(f'calc_iou: bad shapes {tuple(t.shape)} {tuple(s.shape)}')
    N, T, rt = t.shape
    S, rs = s.shape[1], s.shape[2]
    return torch_ops.load().calc_iou(t, s)


_nms_ws = {}


def non_maximum_suppression(outputs: torch.Tensor, iou_thresh: float = 0.5, return_keep: bool = False, variant: str = 'auto',
                            per_class: bool = False) -> torch.Tensor:
    """utils.py:80-116: greedy class-agnostic NMS; zeroes outputs[:, :, 4:] of suppressed / void rows IN PLACE and
    returns `outputs` (the same object), like the reference.
    variant: 'lds' (one workgroup per image, everything in LDS), 'ws' (bit-matrix over the whole chip + serial scan,
    cached workspace; P <= 32768) or 'auto' (= ws).
    per_class (extension, default off = the reference's class-agnostic rule): boxes only suppress boxes of the same class."""
    _lib.require_gpu(outputs)
    if outputs.dim() != 3 or outputs.shape[2] < 6:
        raise ValueError(f'expected (N, P, 4 + C) with C >= 2, got {tuple(outputs.shape)}')
    work = outputs if (outputs.is_contiguous() and outputs.dtype == torch.float32) else _lib.f32c(outputs).clone()
    N, P, row = work.shape
    ops = torch_ops.load()
    if variant in ('ws', 'auto'):
        key = (N, P, str(work.device))
        ws = _nms_ws.get(key)
        if ws is None:
            ws = torch.empty(ops.nms_ws_bytes(N, P), dtype=torch.uint8, device=work.device)
            _nms_ws.clear()
            _nms_ws[key] = ws
        keep = ops.nms_(work, float(iou_thresh), ws, bool(per_class))
    else:
        if per_class:
            raise ValueError("per_class NMS runs on the workspace variant (variant='ws' or 'auto')")
        keep = ops.nms_(work, float(iou_thresh), None, False)
    if work is not outputs:
        outputs.copy_(work)
    return (outputs, keep) if return_keep else outputs


class LabelMap:
    """utils.py:119-137 (static name <-> id lookup)."""

    def __init__(self, ds_name):
        self.ds_name = ds_name
        with open(Path(__file__).parent / 'labelmap.json', 'r') as f:
            self.labels = json.load(f)[ds_name]

    def __len__(self):
        return len(self.labels)

    def name2id(self, name):
        return self.labels.index(name)

    def id2name(self, id):
        return self.labels[id]
