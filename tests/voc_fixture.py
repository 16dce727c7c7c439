"""A tiny Pascal-VOC-shaped directory tree written with PIL (test data: our own synthetic pixels and boxes, no reference
files).  Used by the CPU data-path tests and by tests/golden/make_golden.py (G11), which runs the reference's
dataset.PascalVOCDataset on the same tree."""
import numpy as np

from object_detection_torch2_amd import synth

NAMES = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog', 'horse',
         'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']
# (file id, width, height, [(class name, xmin, ymin, xmax, ymax), ...])
ITEMS = [('000001', 353, 500, [('dog', 48, 240, 195, 371), ('person', 8, 12, 352, 498)]),
         ('000002', 500, 375, [('car', 139, 200, 207, 301)]),
         ('000003', 320, 240, [('cat', 1, 1, 319, 239), ('chair', 100, 50, 180, 170), ('sofa', 20, 30, 300, 200)]),
         ('000004', 64, 48, [])]


def make_tree(root, list_name='trainval.txt'):
    from PIL import Image
    for sub in ('ImageSets/Main', 'JPEGImages', 'Annotations'):
        (root / sub).mkdir(parents=True, exist_ok=True)
    for k, (name, w, h, objs) in enumerate(ITEMS):
        u = synth.uniform(900 + k, 0, h * w * 3).reshape(h, w, 3)
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([xx / w, yy / h, (xx + yy) / (w + h)], axis=2)            # smooth gradients + a little noise: JPEG-friendly
        px = np.clip((0.8 * base + 0.2 * u) * 255.0, 0, 255).astype(np.uint8)
        Image.fromarray(px, 'RGB').save(root / 'JPEGImages' / f'{name}.jpg', quality=90)
        body = ''.join(f'<object><name>{c}</name><bndbox><xmin>{a}</xmin><ymin>{b}</ymin><xmax>{cc}</xmax><ymax>{d}</ymax></bndbox></object>'
                       for c, a, b, cc, d in objs)
        (root / 'Annotations' / f'{name}.xml').write_text(
            f'<annotation><filename>{name}.jpg</filename><size><width>{w}</width><height>{h}</height><depth>3</depth></size>{body}</annotation>')
    (root / 'ImageSets' / 'Main' / list_name).write_text(''.join(f'{name}\n' for name, *_ in ITEMS))
    return root
