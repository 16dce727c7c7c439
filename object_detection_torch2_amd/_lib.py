"""ctypes binding of libssdhip.so (include/ssdhip.h).

The product path has NO CPU fallback: if the HIP library is missing or a tensor is not on the GPU, the call fails
loudly.  torch is plumbing only (device memory + streams); every kernel on the path lives in csrc/*.hip.
"""
import ctypes
import os
from ctypes import c_int, c_int32, c_int64, c_float, c_double, c_void_p, c_char_p, POINTER, Structure

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libssdhip.so')
if os.environ.get('SSDHIP_PROBES') == '1':
    # measurement scripts only (scripts/*_probe.py, *_bench.py): the probe build of the same sources (`make -C csrc probes`), which
    # also holds the A/B and timing-only ablation kernels the product library refuses (tile codes >= 1,000,000)
    LIB_PATH = os.path.join(_HERE, 'lib', os.environ.get('SSDHIP_PROBE_LIB', 'libssdhip_probes.so'))      # SSDHIP_PROBE_LIB: an A/B build of the probe library

OK, EINVAL, ELAUNCH, ENOSUP = 0, -1, -2, -3


class ConvParams(Structure):
    _fields_ = [('x', c_void_p), ('w', c_void_p), ('bias', c_void_p), ('in_scale', c_void_p), ('in_shift', c_void_p),
                ('y', c_void_p), ('stats', c_void_p), ('y_batch_stride', c_int64),
                ('N', c_int32), ('Hi', c_int32), ('Wi', c_int32), ('Cin', c_int32),
                ('Ho', c_int32), ('Wo', c_int32), ('Cout', c_int32),
                ('R', c_int32), ('S', c_int32), ('stride', c_int32), ('pad', c_int32),
                ('KP', c_int32), ('mode', c_int32), ('accumulate', c_int32), ('tile', c_int32),
                ('rows_per_img', c_int32), ('w_img_stride', c_int64), ('compute', c_int32), ('ws', c_void_p), ('ws_bytes', c_int64), ('max_cus', c_int32)]


class WgradParams(Structure):
    _fields_ = [('x', c_void_p), ('dy', c_void_p), ('in_scale', c_void_p), ('in_shift', c_void_p),
                ('dw_oihw', c_void_p), ('dbias', c_void_p), ('ws', c_void_p), ('dy_batch_stride', c_int64),
                ('N', c_int32), ('Hi', c_int32), ('Wi', c_int32), ('Cin', c_int32), ('Ho', c_int32), ('Wo', c_int32),
                ('Cout', c_int32), ('R', c_int32), ('S', c_int32), ('stride', c_int32), ('pad', c_int32), ('dy_row', c_int32),
                ('dw_cin', c_int32), ('wino_v', c_void_p), ('compute', c_int32)]


# every symbol include/ssdhip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    'ssdhip_version': (c_int, []),
    'ssdhip_last_error': (c_char_p, []),
    'ssdhip_cu_count': (c_int, []),
    'ssdhip_default_boxes': (c_int, [c_void_p, POINTER(c_int32), c_int, c_double, c_double, c_void_p]),
    'ssdhip_normalize_nchw_to_nhwc4': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'ssdhip_pack_conv_weight': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_pack_conv_weights': (c_int, [c_void_p, c_int, c_int64, c_void_p]),
    'ssdhip_conv2d_nhwc': (c_int, [POINTER(ConvParams), c_void_p]),
    'ssdhip_conv2d_block_m': (c_int, [POINTER(ConvParams)]),
    'ssdhip_conv2d_ws_bytes': (c_int64, [POINTER(ConvParams)]),
    'ssdhip_conv2d_tile': (c_int, [POINTER(ConvParams)]),
    'ssdhip_winograd_tiles': (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    'ssdhip_winograd_weight': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_winograd_input': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_winograd_output': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_winograd_output_head': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_winograd4_tiles': (c_int, [c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    'ssdhip_winograd4_weight': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_winograd4_input': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_winograd4_stat_rows': (c_int, [c_int, c_int, c_int, c_int]),
    'ssdhip_winograd4_output': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_winograd4_weight_adjoint': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_winograd4_output_head': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_bn_finalize': (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ssdhip_bn_fold_eval': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    'ssdhip_bn_relu_pool': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_multibox_loss_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ssdhip_multibox_loss_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ssdhip_match_mask': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'ssdhip_decode_score': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'ssdhip_calc_iou': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'ssdhip_eval_match': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    'ssdhip_nms': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'ssdhip_nms_ws_bytes': (c_int64, [c_int, c_int]),
    'ssdhip_nms_ws': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    'ssdhip_nms_ws_per_class': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    'ssdhip_conv2d_wgrad_ws_bytes': (c_int64, [POINTER(WgradParams)]),
    'ssdhip_conv2d_wgrad': (c_int, [POINTER(WgradParams), c_void_p]),
    'ssdhip_conv2d_wgrad_winograd_ws_bytes': (c_int64, [POINTER(WgradParams)]),
    'ssdhip_conv2d_wgrad_winograd': (c_int, [POINTER(WgradParams), c_void_p]),
    'ssdhip_conv2d_wgrad_winograd4_ws_bytes': (c_int64, [POINTER(WgradParams)]),
    'ssdhip_conv2d_wgrad_winograd4': (c_int, [POINTER(WgradParams), c_void_p]),
    'ssdhip_bn_relu_bwd_reduce': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                          c_void_p, c_int, c_void_p]),
    'ssdhip_bn_relu_bwd_finalize': (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ssdhip_bn_relu_bwd_apply': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                         c_void_p, c_void_p]),
    'ssdhip_scatter_head_grad': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_scatter_head_grads': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    'ssdhip_adam_multi_tensor': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'ssdhip_adam_tick': (c_int, [c_void_p, c_void_p, c_void_p]),
    'ssdhip_conv_rgb3x3_stat_rows': (c_int, [c_int, c_int, c_int]),
    'ssdhip_conv_rgb3x3': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_conv_c64_weight_floats': (c_int64, [c_int]),
    'ssdhip_conv_c64_pack': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'ssdhip_conv_c64_stat_rows': (c_int, [c_int]),
    'ssdhip_conv_c64': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_wino_fused_weight_floats': (c_int64, [c_int, c_int]),
    'ssdhip_wino_fused_pack': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'ssdhip_wino_fused_stat_rows': (c_int, [c_int, c_int, c_int]),
    'ssdhip_wino_fused_geo_bytes': (c_int64, [c_int, c_int, c_int]),
    'ssdhip_wino_fused_geo': (c_int, [c_int, c_int, c_int, c_void_p, c_void_p]),
    'ssdhip_wino_fused_conv': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_void_p]),
    'ssdhip_wino_fused_conv_pool': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                            c_int, c_int, c_int, c_void_p]),
    'ssdhip_wino_fused_conv_pool_sel': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                        c_int, c_int, c_int, c_void_p]),
    'ssdhip_bn_relu_select': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'ssdhip_conv_bf16_weight_bytes': (c_int64, [c_int, c_int]),
    'ssdhip_conv_bf16_pack': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'ssdhip_conv_bf16_act_bytes': (c_int64, [c_int, c_int, c_int, c_int]),
    'ssdhip_conv_bf16_stat_rows': (c_int, [c_int, c_int, c_int]),
    'ssdhip_conv_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_bn_relu_to_bf16': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_bn_relu_pool_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_nhwc_flatten': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_relu_dropout': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, ctypes.c_uint32, c_int, c_void_p]),
    'ssdhip_log_softmax_dot_fwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    'ssdhip_log_softmax_dot_bwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ssdhip_wino4f_tile_blocks': (c_int, [c_int, c_int, c_int]),
    'ssdhip_wino4f_v_floats': (c_int64, [c_int, c_int, c_int, c_int]),
    'ssdhip_wino4f_pack': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'ssdhip_wino4f_input': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_wino4f_gemm': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_wino4f_gemm_pool': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ssdhip_augment_row_floats': (c_int, []),
    'ssdhip_augment_ws_bytes': (c_int64, [c_int]),
    'ssdhip_augment_batch': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def load():
    """Load libssdhip.so once.  Raises RuntimeError (never falls back) when it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'libssdhip.so not found at {LIB_PATH}: build it with `python __graft_entry__.py` '
                           f'(hipcc --offload-arch=gfx950).  There is no CPU fallback for the SSD hot path.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the build lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    """Turn a negative return code into the reference's exception style (torch raises ValueError / RuntimeError)."""
    if rc >= 0:
        return rc
    msg = load().ssdhip_last_error().decode()
    if rc == EINVAL:
        raise ValueError(f'{what}: {msg}')
    raise RuntimeError(f'{what}: {msg} (code {rc})')


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    # the raw handle of torch's current stream on the current device; torch.cuda.current_stream() builds a Stream object and resolves the
    # device index on every call (~8 us of host time, ~110 calls per training step)
    return c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('object_detection_torch2_amd runs on MI355X only: tensor is on %s. '
                               'There is no CPU fallback (the CPU restatement lives in oracle/ and is test-only).' % t.device)


def f32c(t):
    """Contiguous fp32 view/copy of t (NCHW/row-major edge tensors)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
