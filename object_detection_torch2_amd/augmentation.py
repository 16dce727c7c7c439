"""Drop-in for the reference's `augmentation` package (src/augmentation/{compose,to_tensor,random}.py): same class names,
constructor arguments and `(img, gt) -> (img, gt)` call protocol, so `train.py:32-36`'s pipeline reads the same:

    aug.Compose([aug.RandomColorJitter(p=0.5), aug.RandomFlip(p=0.5), aug.ToTensor(), aug.RandomErasing(p=0.5, max_iter=3)])

MI355X design: the DataLoader workers only DRAW the random parameters (and flip the GT, a handful of floats); the image
stays a uint8 HWC array inside a `Deferred`.  `utils.collate_fn` stacks a batch of them into a `DeferredBatch`, and its
`.to(device)` -- the call the training loop already makes (train.py:106) -- uploads 1 byte per channel instead of 4 and
runs the whole pipeline as one HIP pass over the batch (`ssdhip_augment_batch`, csrc/augment.hip).

Pixel arithmetic = the backend the reference really runs: RandomColorJitter comes BEFORE ToTensor (train.py:32-36), so
torchvision's ColorJitter works on the PIL image (PIL.ImageEnhance + an HSV round trip, uint8 after every operation).
Pillow is importable in the build container: the device pass is pinned bit for bit by PIL's own outputs (tests/golden G12).
The SAMPLING rules restate torchvision's `ColorJitter.get_params` / `RandomErasing.get_params` (the reference subclasses
them); torchvision itself is not installed here, so the order / number of random draws follows its published source, not a
run of the library ("parity unpinned" for the draws; the pixels are pinned).  Random draws use torch's global generator
like the reference (torch.manual_seed controls them)."""
import math

import numpy as np
import torch

from . import _lib

MAX_ERASE = 3


def row_floats():
    return 24            # == ssdhip_augment_row_floats(); asserted against the library in DeferredBatch.to


class Deferred:
    """uint8 (H,W,3) pixels + the parameter row describing the pixel work still to do on the device."""
    __slots__ = ('u8', 'row')

    def __init__(self, img):
        arr = np.asarray(img.convert('RGB') if hasattr(img, 'convert') else img, dtype=np.uint8)
        if arr.ndim != 3 or arr.shape[2] != 3:
            raise ValueError(f'expected an RGB image (H,W,3), got {arr.shape}')
        self.u8 = arr
        self.row = np.zeros(row_floats(), dtype=np.float32)

    @property
    def size(self):      # (H, W)
        return self.u8.shape[0], self.u8.shape[1]


def _deferred(img):
    return img if isinstance(img, Deferred) else Deferred(img)


class Compose:
    """augmentation/compose.py:4-17."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img, gt):
        for t in self.transforms:
            img, gt = t(img, gt)
        return img, gt


class ToTensor:
    """augmentation/to_tensor.py:4-16.  uint8 -> [0,1] fp32 CHW happens inside the device pass; here it only fixes the
    image as a Deferred (so a pipeline of ToTensor alone, like aug_val at train.py:38-39, still yields device work)."""

    def __call__(self, img, gt):
        return _deferred(img), gt


def _range(value, center, lo_bound=0.0, hi_bound=float('inf')):
    if isinstance(value, (tuple, list)):
        lo, hi = float(value[0]), float(value[1])
    else:
        if value < 0:
            raise ValueError('jitter strength must be non-negative')
        lo, hi = center - float(value), center + float(value)
        lo = max(lo, lo_bound) if center == 1.0 else lo
    if not (lo_bound <= lo <= hi <= hi_bound):
        raise ValueError(f'jitter range [{lo}, {hi}] outside [{lo_bound}, {hi_bound}]')
    return lo, hi


def _uniform(lo, hi):
    return float(torch.empty(1).uniform_(lo, hi))


class RandomColorJitter:
    """augmentation/random.py:6-15: with probability p, brightness / contrast / saturation / hue in a random order, each
    factor uniform in its range (brightness, contrast, saturation: [max(0, 1-v), 1+v]; hue: [-v, v], v <= 0.5)."""

    def __init__(self, p: float = 0.5, brightness=0.5, contrast=0.5, saturation=0.5, hue=0.5):
        self.p = p
        self.ranges = (_range(brightness, 1.0), _range(contrast, 1.0), _range(saturation, 1.0), _range(hue, 0.0, -0.5, 0.5))

    def __call__(self, img, gt):
        img = _deferred(img)
        if torch.rand(1) < self.p:
            order = torch.randperm(4)
            img.row[0] = 4
            img.row[1:5] = order.numpy()
            for op, (lo, hi) in enumerate(self.ranges):
                f = _uniform(lo, hi)
                # hue travels as torchvision's uint8 shift of PIL's H channel: np.uint8(hue_factor * 255), computed in double,
                # truncated toward zero, wrapped (adjust_hue: "uint8 addition take cares of rotation across boundaries")
                img.row[5 + op] = f if op < 3 else float(int(f * 255) & 255)
        return img, gt


class RandomFlip:
    """augmentation/random.py:18-32: with probability p mirror the image left-right and the GT centres (`cx -> 1 - cx`)."""

    def __init__(self, p: float = 0.5):
        self.p = p

    def __call__(self, img, gt):
        img = _deferred(img)
        if torch.rand(1) < self.p:
            img.row[9] = 1.0 - img.row[9]
            if gt.dim() == 2:                   # detection rows [cx, cy, ...]; a classification target (one-hot vector) has no geometry
                gt[:, 0] = 1 - gt[:, 0]        # (the reference indexes gt[:, 0] unconditionally and fails on it: random.py:31)
        return img, gt


class RandomErasing:
    """augmentation/random.py:35-42: 1..max_iter rounds; each round erases, with probability p, one zero-filled box whose
    area fraction is uniform in `scale` and aspect ratio log-uniform in `ratio` (ten attempts to fit, as torchvision)."""

    def __init__(self, p=0.5, scale=(0.01, 0.04), ratio=(0.5, 2.0), max_iter=1):
        if not 1 <= max_iter <= MAX_ERASE:
            raise ValueError(f'max_iter must be in 1..{MAX_ERASE} (the device pass holds {MAX_ERASE} boxes per image)')
        self.p, self.scale, self.ratio, self.max_iter = p, scale, ratio, max_iter

    def _box(self, H, W):
        area = H * W
        log_lo, log_hi = math.log(self.ratio[0]), math.log(self.ratio[1])
        for _ in range(10):
            target = area * _uniform(self.scale[0], self.scale[1])
            aspect = math.exp(_uniform(log_lo, log_hi))
            h, w = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
            if not (h < H and w < W):
                continue
            top = int(torch.randint(0, H - h + 1, size=(1,)))
            left = int(torch.randint(0, W - w + 1, size=(1,)))
            return top, left, h, w
        return None

    def __call__(self, img, gt):
        img = _deferred(img)
        rounds = int(torch.randint(low=1, high=self.max_iter + 1, size=(1,)))
        for _ in range(rounds):
            if torch.rand(1) < self.p:
                box = self._box(*img.size)
                k = int(img.row[10])
                if box is not None and k < MAX_ERASE:
                    img.row[11 + 4 * k: 15 + 4 * k] = box
                    img.row[10] = k + 1
        return img, gt


class DeferredBatch:
    """What `utils.collate_fn` returns for a batch of Deferred images: `.to(device)` gives the augmented (N,3,H,W) fp32
    tensor (upload + one device pass), which is what `net(images)` takes."""

    def __init__(self, u8: torch.Tensor, rows: torch.Tensor):
        self.u8, self.rows = u8, rows            # (N,H,W,3) uint8, (N,24) fp32

    @classmethod
    def stack(cls, items):
        shapes = {d.u8.shape for d in items}
        if len(shapes) != 1:
            raise ValueError(f'images of one batch must share a size, got {sorted(shapes)}')
        return cls(torch.from_numpy(np.stack([d.u8 for d in items])), torch.from_numpy(np.stack([d.row for d in items])))

    def __len__(self):
        return self.u8.shape[0]

    @property
    def shape(self):
        n, h, w, _ = self.u8.shape
        return torch.Size((n, 3, h, w))

    def pin_memory(self):
        return DeferredBatch(self.u8.pin_memory(), self.rows.pin_memory())

    def to(self, device, non_blocking=False):
        u8 = self.u8.to(device, non_blocking=non_blocking).contiguous()
        rows = self.rows.to(device, non_blocking=non_blocking).contiguous()
        return augment_batch(u8, rows)


_ws = {}


def augment_batch(u8: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
    """(N,H,W,3) uint8 + (N,24) fp32 parameter rows, both on the GPU -> (N,3,H,W) fp32 in [0,1]."""
    _lib.require_gpu(u8, rows)
    lib = _lib.load()
    if u8.dtype != torch.uint8 or u8.dim() != 4 or u8.shape[3] != 3 or not u8.is_contiguous():
        raise ValueError(f'augment_batch: expected contiguous uint8 (N,H,W,3), got {u8.dtype} {tuple(u8.shape)}')
    N, H, W, _ = u8.shape
    if rows.dtype != torch.float32 or tuple(rows.shape) != (N, lib.ssdhip_augment_row_floats()) or not rows.is_contiguous():
        raise ValueError(f'augment_batch: expected fp32 parameter rows ({N}, {lib.ssdhip_augment_row_floats()}), got {tuple(rows.shape)}')
    key = (N, str(u8.device))
    ws = _ws.get(key)
    if ws is None:
        ws = torch.empty(lib.ssdhip_augment_ws_bytes(N), dtype=torch.uint8, device=u8.device)
        _ws.clear()
        _ws[key] = ws
    out = torch.empty(N, 3, H, W, dtype=torch.float32, device=u8.device)
    _lib.check(lib.ssdhip_augment_batch(_lib.ptr(u8), _lib.ptr(rows), N, H, W, _lib.ptr(out), _lib.ptr(ws), _lib.stream()), 'augment_batch')
    return out
