"""GPU parity of the device augmentation pass: bit-exact against the oracle's restatement of the PIL backend the
reference's pipeline runs (train.py:32-36: colour jitter before ToTensor), itself pinned by PIL's own outputs (G12: see
tests/test_data_cpu.py and csrc/augment.hip) and against the G12 bytes directly; plus the end-to-end data path
dataset -> DataLoader(collate_fn) -> .to(device) -> net."""
import pathlib
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import voc_fixture  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rows(seed, N, H, W):
    """Random but valid parameter rows, all features on for most images."""
    from object_detection_torch2_amd import synth
    u = synth.uniform(seed, 3, N * 40).reshape(N, 40)
    rows = np.zeros((N, 24), dtype=np.float32)
    for n in range(N):
        if u[n, 0] < 0.8:
            rows[n, 0] = 4
            rows[n, 1:5] = np.argsort(u[n, 1:5])
            rows[n, 5:8] = 0.5 + u[n, 5:8]
            rows[n, 8] = int((u[n, 8] - 0.5) * 255) & 255             # hue: torchvision's uint8 shift of PIL's H channel
        rows[n, 9] = u[n, 9] < 0.5
        ne = int(u[n, 10] * 4)
        rows[n, 10] = ne
        for k in range(ne):
            h, w = 1 + int(u[n, 11 + 4 * k] * (H // 3)), 1 + int(u[n, 12 + 4 * k] * (W // 3))
            rows[n, 11 + 4 * k: 15 + 4 * k] = [int(u[n, 13 + 4 * k] * (H - h)), int(u[n, 14 + 4 * k] * (W - w)), h, w]
    return rows


@pytest.mark.parametrize('N,H,W', [(1, 7, 5), (3, 64, 48), (8, 300, 300)])
def test_augment_batch_matches_oracle(N, H, W):
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import synth, augmentation as aug
    u8 = (synth.uniform(100 + N, 1, N * H * W * 3) * 256).astype(np.uint8).reshape(N, H, W, 3)
    u8[0, : H // 2, : W // 2] = 77                                            # a gray patch: the max == min branch of the hue op
    rows = _rows(200 + N, N, H, W)
    rows[0, 0] = 4; rows[0, 1:5] = [3, 1, 0, 2]; rows[0, 5:9] = [1.3, 0.6, 1.4, 94]
    got = aug.augment_batch(torch.from_numpy(u8).to(DEV), torch.from_numpy(rows).to(DEV)).cpu().numpy()
    want = O.augment_batch(u8, rows)
    assert got.shape == (N, 3, H, W) and got.dtype == np.float32
    assert np.array_equal(got, want)                                          # integer pipeline + one IEEE division: every bit


def test_augment_batch_matches_pil_golden(golden):
    """The device pass against PIL's own bytes (G12), every case, both images."""
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import augmentation as aug
    fx = golden('g12_augment_pil.npz')
    for i in range(2):
        im = fx[f'img_{i}']
        nc = int(fx['n_cases'])
        rows = np.zeros((nc, 24), dtype=np.float32)
        for ci in range(nc):
            ops = fx[f'ops_{ci}']
            # a row always carries 4 ops: pad single-op cases with identities (factor 1 is exact for the three blends; the hue
            # identity is NOT exact, so a case without hue gets a zero-length chain instead: n_ops = len(ops))
            rows[ci, 0] = len(ops)
            rows[ci, 1:1 + len(ops)] = ops[:, 0]
            rows[ci, 5:8] = 1
            for op, prm in ops:
                rows[ci, 5 + int(op)] = prm if op < 3 else O.hue_shift(prm)
        u8 = np.repeat(im[None], nc, axis=0)
        got = aug.augment_batch(torch.from_numpy(u8).to(DEV), torch.from_numpy(rows).to(DEV)).cpu().numpy()
        for ci in range(nc):
            want = (fx[f'out_{ci}_{i}'].astype(np.float32) / np.float32(255)).transpose(2, 0, 1)
            assert np.array_equal(got[ci], want), (ci, i)


def test_augment_rejects_bad_arguments():
    from object_detection_torch2_amd import augmentation as aug
    u8 = torch.zeros(2, 8, 8, 3, dtype=torch.uint8, device=DEV)
    with pytest.raises(ValueError):
        aug.augment_batch(u8, torch.zeros(2, 23, device=DEV))
    with pytest.raises(ValueError):
        aug.augment_batch(u8.float(), torch.zeros(2, 24, device=DEV))
    with pytest.raises(ValueError):
        aug.augment_batch(u8[:, :, :, :2], torch.zeros(2, 24, device=DEV))
    # garbage parameter rows must not fault: op ids / counts are clamped on the device
    out = aug.augment_batch(u8, torch.full((2, 24), 1e6, device=DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()


def test_data_path_end_to_end(tmp_path):
    """train.py:32-66,103-120 with the drop-in modules: the loop's `images.to(device)` is where the device pass runs."""
    from oracle import ssd_oracle as O
    from object_detection_torch2_amd import augmentation as aug, dataset as ds, utils
    from object_detection_torch2_amd.model import SSD
    root = voc_fixture.make_tree(tmp_path)
    pipe = aug.Compose([aug.RandomColorJitter(p=0.5), aug.RandomFlip(p=0.5), aug.ToTensor(), aug.RandomErasing(p=0.5, max_iter=3)])
    data = ds.PascalVOCDataset('detection', [str(root)], 'trainval.txt', 300, transform=pipe)
    torch.manual_seed(3)
    loader = torch.utils.data.DataLoader(data, batch_size=4, shuffle=False, num_workers=0, collate_fn=utils.collate_fn)
    images, gts = next(iter(loader))
    want = O.augment_batch(images.u8.numpy(), images.rows.numpy())
    ims = images.to(DEV)
    assert isinstance(ims, torch.Tensor) and tuple(ims.shape) == (4, 3, 300, 300) and ims.dtype == torch.float32
    assert np.array_equal(ims.cpu().numpy(), want)                          # pinned arithmetic: every bit
    net = SSD(21)
    net.load_state_dict(O.init_state(1))
    net.to(DEV)
    out = net(ims)
    loss = net.loss(out, gts.to(DEV), net.default_bboxes.to(DEV))
    assert tuple(out.shape) == (4, 8732, 25) and torch.isfinite(loss)


def test_train_entry_point_on_voc_tree(tmp_path, monkeypatch, capsys):
    """`python -m object_detection_torch2_amd.train --data_dirs ...` (train.py:32-66 pipeline) for one epoch of the tiny tree."""
    from object_detection_torch2_amd import train
    root = voc_fixture.make_tree(tmp_path / 'voc')
    monkeypatch.setattr(sys, 'argv', ['train', '--data_dirs', str(root), '--batch_size', '2', '--epochs', '1', '--num_workers', '0',
                                      '--result_dir', str(tmp_path / 'result')])
    torch.manual_seed(0)
    train.main()
    out = capsys.readouterr().out
    assert 'Finished Training' in out and '[Epoch 1/1] loss:' in out
    assert (tmp_path / 'result' / 'detection' / 'weights.pth').exists()
    sd = torch.load(tmp_path / 'result' / 'detection' / 'weights.pth', map_location='cpu', weights_only=True)
    assert len(sd) == 173 and all(torch.isfinite(v.float()).all() for v in sd.values())


def test_evaluate_and_inference_entry_points(tmp_path, monkeypatch, capsys):
    """evaluate.py on the tiny VOC tree (evaluate.py:90-132 pipeline: ToTensor only) and inference.py on synthetic batches."""
    from object_detection_torch2_amd import evaluate, inference
    root = voc_fixture.make_tree(tmp_path / 'voc', list_name='test.txt')
    monkeypatch.setattr(sys, 'argv', ['evaluate', '--data_dirs', str(root), '--batch_size', '2', '--num_workers', '0',
                                      '--result_dir', str(tmp_path / 'result')])
    evaluate.main()
    out = capsys.readouterr().out
    assert 'Finished Evaluate' in out
    reports = list((tmp_path / 'result' / 'detection').glob('report_*.md'))
    assert len(reports) == 1 and '|**mean**|' in reports[0].read_text()
    monkeypatch.setattr(sys, 'argv', ['inference', '--batch_size', '2', '--result_dir', str(tmp_path / 'result')])
    inference.main()
    assert 'Finished' in capsys.readouterr().out
    # inference.py:25-69 over the dataset (PascalVOCDataset + ToTensor + collate_fn), annotated images like inference.py:101
    monkeypatch.setattr(sys, 'argv', ['inference', '--data_dirs', str(root), '--batch_size', '2', '--num_workers', '0', '--save_png',
                                      '--result_dir', str(tmp_path / 'result')])
    inference.main()
    out = capsys.readouterr().out
    n_images = len((root / 'ImageSets' / 'Main' / 'test.txt').read_text().split())
    assert out.count('detections, top:') == n_images and 'Finished Inference' in out
    pngs = sorted((tmp_path / 'result' / 'detection').glob('0*.png'))
    assert len(pngs) == n_images
    from PIL import Image
    assert Image.open(pngs[0]).size == (300, 300)
