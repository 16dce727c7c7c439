"""torch.ops.ssdhip.*: the TORCH_LIBRARY face of libssdhip.so (csrc/torch_ops.cpp).

The reference's boundary is its Python surface (`model.SSD.loss`, `utils.calc_coordicate / calc_score / calc_iou /
non_maximum_suppression`, utils.py:19-116, ssd.py:181-229); those shims call the operators registered here, each a thin C++
wrapper over one extern "C" launcher.  The layer-by-layer engine (model/engine.py) binds the same C ABI with ctypes: it hands the
launchers raw workspace pointers and offsets into preallocated buffers that have no tensor form.

There is no fallback: a missing library raises RuntimeError."""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libssdhip_torch.so')
_ops = None


def load():
    global _ops
    if _ops is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'libssdhip_torch.so not found at {LIB_PATH}: build it with `python __graft_entry__.py` '
                               '(make -C object_detection_torch2_amd/csrc torch_ops).  There is no CPU fallback for the SSD hot path.')
        torch.ops.load_library(LIB_PATH)
        _ops = torch.ops.ssdhip
        if _ops.version() < 100:
            raise RuntimeError('libssdhip_torch.so: unexpected library version')
    return _ops
