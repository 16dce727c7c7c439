"""CPU: host logic, module surface and the C-ABI library (load + every declared symbol, no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from object_detection_torch2_amd import _lib, synth
from object_detection_torch2_amd.model import SSD, topology as T
from oracle import ssd_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'ssdhip.h')).read()
    declared = set(re.findall(r'\b(ssdhip_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations found'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in include/ssdhip.h but not exported'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().ssdhip_version() >= 100


def test_struct_layout_matches_header(tmp_path):
    """The ctypes mirrors must have the C layout include/ssdhip.h declares: ask the C compiler for size and field offsets."""
    import subprocess
    fields = {'ssdhip_conv_params': (_lib.ConvParams, ['x', 'y_batch_stride', 'N', 'KP', 'tile', 'rows_per_img', 'w_img_stride', 'compute', 'ws_bytes', 'max_cus']),
              'ssdhip_wgrad_params': (_lib.WgradParams, ['x', 'ws', 'dy_batch_stride', 'N', 'pad', 'dy_row', 'dw_cin', 'wino_v'])}
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/ssdhip.h"', 'int main(void) {']
    for t, (_, names) in fields.items():
        prog.append(f'printf("%zu", sizeof({t}));')
        prog += [f'printf(" %zu", offsetof({t}, {n}));' for n in names]
        prog.append('printf("\\n");')
    prog.append('return 0; }')
    (tmp_path / 'layout.c').write_text('\n'.join(prog))
    subprocess.run(['gcc', str(tmp_path / 'layout.c'), '-o', str(tmp_path / 'layout')], check=True)
    lines = subprocess.run([str(tmp_path / 'layout')], check=True, capture_output=True, text=True).stdout.strip().split('\n')
    for line, (t, (cls, names)) in zip(lines, fields.items()):
        want = [int(v) for v in line.split()]
        got = [ctypes.sizeof(cls)] + [getattr(cls, n).offset for n in names]
        assert got == want, (t, got, want)


def test_module_surface_matches_reference_contract():
    net = SSD(num_classes=21)
    sd = net.state_dict()
    ref = O.init_state(1)
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 173
    assert all(sd[k].shape == ref[k].shape for k in sd)
    assert [n for n, _ in net.named_train_params()] == O.trainable_param_names()
    assert sum(p.numel() for p in net.train_params()) == 11578222
    assert list(net.detectors.keys()) == ['det_4_3', 'det_7_1', 'det_8_2', 'det_9_2', 'det_10_2', 'det_11_2']
    assert 'pool_5' not in net.features and len(net.features) == 73
    frozen = [k for k, p in net.named_parameters() if not p.requires_grad]
    assert len(frozen) == 52 and all(k.startswith('features.') for k in frozen)
    net.load_state_dict(ref)                                    # weights.pth interchange
    assert torch.equal(net.features['conv_6_1'].weight, ref['features.conv_6_1.weight'])


def test_initialize_weights_statistics():
    """A18 (ssd.py:135-158): extras + detectors kaiming_normal_(fan_out, relu) -> std = sqrt(2 / (Cout k^2)), conv bias 0,
    BatchNorm gamma 1 / beta 0 (conv_6_1 std 0.01473 probed on the reference)."""
    torch.manual_seed(0)
    net = SSD(num_classes=21)
    w = net.features['conv_6_1'].weight
    assert abs(w.std().item() - 0.01473) < 2e-4 and abs(w.mean().item()) < 2e-5          # 4.7 M samples
    for name, mod in list(net.features.items()) + list(net.detectors.items()):
        if isinstance(mod, torch.nn.Conv2d):
            cout, _, k, _ = mod.weight.shape
            expect = (2.0 / (cout * k * k)) ** 0.5
            n = mod.weight.numel()
            assert abs(mod.weight.std().item() / expect - 1) < 4.0 / n ** 0.5 + 1e-3, name   # ~5 sigma of the std estimator
            assert torch.count_nonzero(mod.bias).item() == 0, name
        elif isinstance(mod, torch.nn.BatchNorm2d):
            assert bool((mod.weight == 1).all()) and bool((mod.bias == 0).all()), name
            assert bool((mod.running_mean == 0).all()) and bool((mod.running_var == 1).all()) and int(mod.num_batches_tracked) == 0


def test_no_cpu_fallback():
    net = SSD(num_classes=21)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net(torch.zeros(2, 3, 300, 300))
    from object_detection_torch2_amd import utils
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        utils.non_maximum_suppression(torch.zeros(1, 10, 25))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net.loss(torch.zeros(1, 8732, 25), torch.zeros(1, 2, 25), torch.zeros(8732, 4))


def test_topology_tables():
    assert T.num_priors() == 8732
    plan = T.feature_plan()
    assert [l['name'] for l in plan] == [l['name'] for l in O.feature_layers()]
    assert sum(l['kind'] == 'conv' for l in plan) == 23 and sum(l['kind'] == 'pool' for l in plan) == 4


def test_synth_streams_are_reproducible():
    a, b = synth.images(5, 1, 8), synth.images(5, 1, 8)
    assert np.array_equal(a, b) and a.min() >= 0 and a.max() < 1
    t = synth.targets(7, 4)
    assert t.shape == (4, 8, 25)
    rows = t[t[..., 2] > 0]
    assert np.allclose(rows[:, 4:].sum(axis=1), 1) and (rows[:, 4] == 0).all()
    assert (rows[:, 0] - rows[:, 2] / 2 >= -1e-6).all() and (rows[:, 0] + rows[:, 2] / 2 <= 1 + 1e-6).all()
    z = synth.pseudo_normal(3, 9, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    # pinned words: the streams must never change silently (fixtures depend on them)
    assert synth.hash_u32(1, 2, 3).tolist() == synth.hash_u32(1, 2, 3).tolist()


def test_average_precision_matches_reference_and_voc(golden):
    """AP bookkeeping (evaluate.py:31-67): reference-compatible mode reproduces the reference incl. its independent-column
    sort (Q11: TPs ranked first); score-ranked mode is standard VOC AP and is never larger."""
    from object_detection_torch2_amd import evaluate as E
    fx = golden('g10_eval.npz')
    rng = synth.uniform(77, 5, 4000)
    k = 0
    for case, n in enumerate((1, 7, 40, 200)):
        tp = (rng[k:k + n] > 0.6).astype(np.float32); k += n
        sc = rng[k:k + n].astype(np.float32); k += n
        res = torch.from_numpy(np.stack([tp, sc], 1))
        cnt = int(fx[f'ap_cnt_{case}'])
        ref = E.calc_average_precision(res.clone(), cnt, True).item()
        voc = E.calc_average_precision(res.clone(), cnt, False).item()
        assert abs(ref - float(fx[f'ap_ref_{case}'])) < 1e-7 and abs(voc - float(fx[f'ap_voc_{case}'])) < 1e-7
        assert voc <= ref + 1e-7
    t = torch.from_numpy(synth.uniform(78, 6, 50 * 25).astype(np.float32).reshape(50, 25)); t[::3, 5 + 4] = 0
    assert np.array_equal(E.get_order(t, 4).numpy(), fx['order_c4'])
    # TP assignment: two predictions claim GT 0, only the first (higher score) is a TP; the third matches nothing
    iou = torch.tensor([[0.9, 0.1], [0.8, 0.2], [0.3, 0.4]])
    assert E.assign_true_positives(iou).view(-1).tolist() == [1.0, 0.0, 0.0]
    # the oracle's restatement of the evaluation loop (evaluate.py:31-42, 134-151) agrees with the reference-pinned pieces
    assert np.array_equal(O.get_order(t.numpy(), 4), fx['order_c4'])
    out = np.zeros((5, 25), np.float32); gt = np.zeros((3, 25), np.float32)
    gt[0, :4] = [0.5, 0.5, 0.2, 0.2]; gt[0, 5 + 2] = 1          # class 2
    gt[1, :4] = [0.2, 0.2, 0.1, 0.1]; gt[1, 5 + 7] = 1          # class 7, nobody claims it
    out[0, :4] = [0.5, 0.5, 0.2, 0.2]; out[0, 5 + 2] = 0.6       # exact hit, but lower score than row 3
    out[3, :4] = [0.51, 0.5, 0.2, 0.2]; out[3, 5 + 2] = 0.9      # first claimant of GT 0 -> TP
    out[1, :4] = [0.9, 0.9, 0.1, 0.1]; out[1, 5 + 2] = 0.8       # class 2, far away -> FP
    out[2, :4] = [0.5, 0.5, 0.2, 0.2]; out[2, 5 + 3] = 0.7       # right place, wrong class, no GT of class 3 -> FP
    rows, counts = O.eval_image(out, gt)
    assert counts[2] == 1 and counts[7] == 1 and counts[3] == 0 and sorted(rows) == [2, 3]
    assert rows[2].tolist() == [[1.0, np.float32(0.9)], [0.0, np.float32(0.8)], [0.0, np.float32(0.6)]]
    assert rows[3].tolist() == [[0.0, np.float32(0.7)]]
    # ... and with the torch form of the same expression used by the host module
    iou2 = torch.from_numpy(O.calc_iou(out[None], gt[None])[0])
    po, go = E.get_order(torch.from_numpy(out), 2), E.get_order(torch.from_numpy(gt), 2)
    assert E.assign_true_positives(iou2[po][:, go]).view(-1).tolist() == rows[2][:, 0].tolist()


def test_plain_c_program_links_against_the_library(tmp_path):
    """The boundary is a C ABI: a C99 translation unit that includes include/ssdhip.h links against libssdhip.so and calls it
    (version query and an argument error with its message: no GPU needed)."""
    import subprocess
    src = tmp_path / 'client.c'
    src.write_text(f'''
#include <stdio.h>
#include <string.h>
#include "{ROOT}/include/ssdhip.h"
int main(void) {{
    ssdhip_conv_params p;
    memset(&p, 0, sizeof p);
    int rc = ssdhip_conv2d_nhwc(&p, NULL);              /* null pointers: must be rejected before any launch */
    printf("%d %d %s\\n", ssdhip_version(), rc, ssdhip_last_error());
    return 0;
}}
''')
    exe = tmp_path / 'client'
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', str(src), '-o', str(exe), f'-L{libdir}', '-lssdhip', f'-Wl,-rpath,{libdir}'], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split(maxsplit=2)
    assert int(out[0]) >= 100 and int(out[1]) < 0 and 'null pointer' in out[2]


def test_product_library_refuses_probe_tile_codes():
    """Tile codes >= 1,000,000 select A/B and timing-only ablation kernels (some return wrong results by design).  They are compiled
    into libssdhip_probes.so only; the product library must reject them as invalid arguments (checked before any launch, so fake
    device pointers are safe here)."""
    import ctypes
    lib = _lib.load()
    p = _lib.ConvParams()
    p.x, p.w, p.y = 0x1000, 0x2000, 0x3000
    p.N, p.Hi, p.Wi, p.Cin, p.Ho, p.Wo, p.Cout = 16, 1, 128, 256, 1, 128, 256
    p.R, p.S, p.stride, p.pad, p.KP, p.mode, p.accumulate, p.compute = 1, 1, 1, 0, 256, 0, 0, 0
    p.y_batch_stride, p.rows_per_img, p.w_img_stride = 128 * 256, 128, 256 * 256
    for code in (1128128, 6128128, 7128128, 8128128, 9128128, 41128128):
        p.tile = code
        assert lib.ssdhip_conv2d_nhwc(ctypes.byref(p), None) == _lib.EINVAL
        assert b'tile' in lib.ssdhip_last_error()



def test_winograd4_matrices_are_exact():
    """The F(4x4,3x3) matrices of csrc/winograd4.hip (copies in ops.WINOGRAD4_*) satisfy the correlation identity in exact rational
    arithmetic: y_i = sum_k g_k d_{i+k} == sum_p AT[i][p] (G g)[p] (BT d)[p] for every monomial, i.e. sum_p AT[i][p] G[p][k] BT[p][j] =
    [j == i + k]; and the float constants in the kernel source are these rationals."""
    import os
    import re
    from fractions import Fraction as Fr
    from object_detection_torch2_amd import ops
    BT = [[Fr(v) for v in r] for r in ops.WINOGRAD4_BT]
    G = [[Fr(v) for v in r] for r in ops.WINOGRAD4_G]
    AT = [[Fr(v) for v in r] for r in ops.WINOGRAD4_AT]
    assert (len(BT), len(BT[0]), len(G), len(G[0]), len(AT), len(AT[0])) == (6, 6, 6, 3, 4, 6)
    for i in range(4):
        for k in range(3):
            for j in range(6):
                assert sum(AT[i][p] * G[p][k] * BT[p][j] for p in range(6)) == (1 if j == i + k else 0), (i, k, j)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'object_detection_torch2_amd', 'csrc', 'winograd4.hip')).read()
    consts = dict(re.findall(r'(kC\w+) = ([0-9.f /]+?)[,;]', src))
    want = {'kC97_36': Fr(97, 36), 'kC4_9': Fr(4, 9), 'kC9_4': Fr(9, 4), 'kC3_2': Fr(3, 2), 'kC2_3': Fr(2, 3), 'kC27_8': Fr(27, 8), 'kC8_27': Fr(8, 27)}
    assert set(consts) == set(want), consts
    for name, expr in consts.items():
        parts = [Fr(t.strip().rstrip('f')) for t in expr.split('/')]
        val = parts[0] / parts[1] if len(parts) == 2 else parts[0]
        assert val == want[name], (name, expr)
    used = {abs(v) for r in BT + AT for v in r} - {Fr(0), Fr(1)}
    assert used == set(want.values())
    for row in re.findall(r'\{(-?[0-9.]+ / [0-9.]+|[-0-9.]+), (-?[0-9.]+ / [0-9.]+|[-0-9.]+), (-?[0-9.]+ / [0-9.]+|[-0-9.]+)\}', src):
        pass
    grows = re.search(r'const double G\[6\]\[3\] = \{(.*?)\};', src, re.S).group(1)
    vals = [Fr(a.replace('.0', '')) / Fr(b.replace('.0', '')) if b else Fr(a.replace('.0', '')) for a, b in re.findall(r'(-?[0-9.]+)(?: / ([0-9.]+))?', grows)]
    assert vals == [v for r in G for v in r]


ASAN_CLIENT = r"""
#include <stdio.h>
#include <string.h>
#include <stdint.h>
#include "%(root)s/include/ssdhip.h"
/* every call below must be rejected by the host-side argument checks BEFORE any launch (there is no GPU here): null pointers, zero and
   negative extents, channel counts the kernels do not tile, overflowing sizes.  Run under AddressSanitizer + UBSan. */
static int bad = 0;
#define REJECT(expr) do { int rc_ = (int)(expr); if (rc_ >= 0) { printf("NOT REJECTED: %%s -> %%d\n", #expr, rc_); bad++; } \
                          else if (!ssdhip_last_error() || !ssdhip_last_error()[0]) { printf("NO MESSAGE: %%s\n", #expr); bad++; } } while (0)
int main(void) {
    float* F = (float*)0x1000;                 /* fake device pointers: never dereferenced on the host */
    ssdhip_conv_params p; memset(&p, 0, sizeof p);
    REJECT(ssdhip_conv2d_nhwc(&p, NULL));
    REJECT(ssdhip_conv2d_nhwc(NULL, NULL));
    p.x = F; p.w = F; p.y = F; p.N = 2; p.Hi = 8; p.Wi = 8; p.Cin = 48; p.Ho = 8; p.Wo = 8; p.Cout = 64; p.R = 3; p.S = 3; p.stride = 1; p.pad = 1; p.KP = 448;
    REJECT(ssdhip_conv2d_nhwc(&p, NULL));                       /* Cin not 4 / multiple of 32 */
    p.Cin = 64; p.KP = 576; p.N = -1;
    REJECT(ssdhip_conv2d_nhwc(&p, NULL));
    p.N = 2; p.tile = 96096;
    REJECT(ssdhip_conv2d_nhwc(&p, NULL));
    p.tile = 0; p.stride = 0;
    REJECT(ssdhip_conv2d_nhwc(&p, NULL));
    p.stride = 1; p.N = 1 << 30; p.Hi = 1 << 15; p.Wi = 1 << 15; p.Ho = 1 << 15; p.Wo = 1 << 15;      /* M overflows 32 bits */
    REJECT(ssdhip_conv2d_nhwc(&p, NULL));
    REJECT(ssdhip_normalize_nchw_to_nhwc4(NULL, F, 1, 8, 8, NULL));
    REJECT(ssdhip_normalize_nchw_to_nhwc4(F, F, 0, 8, 8, NULL));
    REJECT(ssdhip_pack_conv_weight(F, F, 64, 64, 3, 3, 100, 64, 576, 0, NULL));
    REJECT(ssdhip_pack_conv_weight(NULL, F, 64, 64, 3, 3, 128, 64, 576, 0, NULL));
    int T = 0, Tp = 0;
    REJECT(ssdhip_winograd_tiles(0, 8, 8, &T, &Tp));
    REJECT(ssdhip_winograd4_tiles(1, 8, 8, 100, &T, &Tp));
    REJECT(ssdhip_winograd_input(F, NULL, NULL, F, 1, 8, 8, 30, NULL));
    REJECT(ssdhip_winograd4_input(F, NULL, NULL, F, 1, 8, 8, 64, 3, NULL));      /* Tp smaller than the tile count */
    REJECT(ssdhip_winograd4_weight(F, NULL, 64, 64, 128, 64, NULL));
    REJECT(ssdhip_winograd4_output(F, NULL, NULL, NULL, 0, 1, 8, 8, 64, 128, NULL));
    int32_t maps[3] = {38, 38, 4};
    REJECT(ssdhip_default_boxes(NULL, maps, 1, 0.2, 0.9, NULL));
    REJECT(ssdhip_default_boxes(F, maps, 0, 0.2, 0.9, NULL));
    REJECT(ssdhip_multibox_loss_fwd(F, F, F, 0, 8732, 8, 21, 1.0f, 0.25f, F, (int32_t*)F, F, F, NULL));
    REJECT(ssdhip_multibox_loss_fwd(F, F, NULL, 2, 8732, 8, 21, 1.0f, 0.25f, F, (int32_t*)F, F, F, NULL));
    REJECT(ssdhip_decode_score(NULL, F, 1, 8732, 21, NULL));
    REJECT(ssdhip_nms(F, 1, 200000, 21, 0.5f, NULL, NULL));
    REJECT(ssdhip_nms_ws(F, 1, 8732, 21, 0.5f, NULL, NULL, NULL));
    REJECT(ssdhip_adam_multi_tensor(NULL, NULL, NULL, 1, 8192, NULL, NULL));
    REJECT(ssdhip_bn_finalize(F, 0, 64, 100, F, F, NULL, NULL, 0.1f, 1e-5f, F, F, F, F, NULL, NULL));
    REJECT(ssdhip_bn_relu_pool(F, F, F, F, 1, 8, 8, 30, 1, 0, NULL));
    REJECT(ssdhip_conv_rgb3x3(F, F, F, NULL, NULL, 1, 8, 8, 64, NULL));
    REJECT(ssdhip_wino_fused_conv(F, F, F, NULL, NULL, F, NULL, NULL, 1, 8, 8, 24, 64, 0, NULL));
    REJECT(ssdhip_conv_bf16(F, F, F, F, 0, NULL, 1, 8, 8, 24, 64, 0, NULL));
    ssdhip_wgrad_params g; memset(&g, 0, sizeof g);
    REJECT(ssdhip_conv2d_wgrad(&g, NULL));
    REJECT(ssdhip_conv2d_wgrad_winograd4(&g, NULL));
    /* sizing queries on valid shapes: pure host arithmetic */
    memset(&p, 0, sizeof p);
    p.x = F; p.w = F; p.y = F; p.N = 32; p.Hi = 3; p.Wi = 3; p.Cin = 256; p.Ho = 1; p.Wo = 1; p.Cout = 256; p.R = 3; p.S = 3; p.stride = 1; p.pad = 0; p.KP = 2304;
    p.y_batch_stride = 256;
    long long ws = (long long)ssdhip_conv2d_ws_bytes(&p);
    int bm = ssdhip_conv2d_block_m(&p), tile = ssdhip_conv2d_tile(&p);
    printf("version %%d ws %%lld bm %%d tile %%d stat_rows %%d geo %%lld bad %%d\n", ssdhip_version(), ws, bm, tile,
           ssdhip_winograd4_stat_rows(32, 38, 38, 512), (long long)ssdhip_wino_fused_geo_bytes(32, 150, 150), bad);
    return bad ? 1 : 0;
}
"""


def test_asan_build_of_the_host_side_rejects_bad_arguments(tmp_path):
    """SURVEY.md section 5 (sanitizers): `make asan` = the library's host side under AddressSanitizer + UndefinedBehaviorSanitizer (device
    code uninstrumented: GPU ASan needs xnack+, which the pool refuses).  A C client drives the argument-validation paths of the
    launchers -- null pointers, zero / negative / overflowing extents, untiled channel counts -- and two sizing queries; any
    out-of-bounds access, signed overflow or misaligned access on those paths aborts the client."""
    import glob
    import subprocess
    csrc = os.path.join(ROOT, 'object_detection_torch2_amd', 'csrc')
    subprocess.run(['make', '-C', csrc, 'asan', '-j8'], check=True, capture_output=True)
    libdir = os.path.dirname(_lib.LIB_PATH)
    rt = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
    assert rt, 'clang AddressSanitizer runtime not found'
    src = tmp_path / 'asan_client.c'
    src.write_text(ASAN_CLIENT % dict(root=ROOT))
    exe = tmp_path / 'asan_client'
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-g', str(src), '-o', str(exe), f'-L{libdir}', '-lssdhip_asan',
                    f'-Wl,-rpath,{libdir}', f'-Wl,-rpath,{os.path.dirname(rt[0])}'], check=True)
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:halt_on_error=1',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'bad 0' in r.stdout and 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stdout + r.stderr


def test_torch_library_ops_are_registered():
    """TORCH_LIBRARY(ssdhip) (csrc/torch_ops.cpp): the wrapper library loads without a GPU, registers the minimum operator
    set of 8(b) with tensor-typed schemas, reports the C library's version and refuses CPU tensors (no fallback)."""
    import torch
    from object_detection_torch2_amd import torch_ops
    S = torch_ops.load()
    assert S.version() == _lib.load().ssdhip_version()
    want = ['default_boxes', 'normalize_nchw_to_nhwc4', 'pack_conv_weight', 'conv2d_nhwc', 'conv2d_wgrad', 'bn_finalize', 'bn_relu_pool',
            'multibox_loss_fwd', 'multibox_loss_bwd', 'decode_score_', 'nms_', 'nms_ws_bytes', 'calc_iou', 'adam_multi_tensor_']
    for name in want:
        schema = str(getattr(S, name).default._schema)
        assert schema.startswith(f'ssdhip::{name}('), schema
    assert 'Tensor(a!) outputs' in str(S.nms_.default._schema)                      # in-place like utils.py:114
    assert S.nms_ws_bytes(32, 8732) == _lib.load().ssdhip_nms_ws_bytes(32, 8732)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        S.decode_score_(torch.zeros(1, 10, 25), torch.zeros(10, 4))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        S.multibox_loss_fwd(torch.zeros(1, 10, 25), torch.zeros(1, 2, 25), torch.zeros(10, 4), 1.0, 0.25)
