"""Inference entry point with the reference's flags (src/inference.py:14-20): forward -> decode -> score -> NMS
(inference.py:65-69) over a dataset (inference.py:25-40: PascalVOCDataset + ToTensor through collate_fn, with --data_dirs)
or over synthetic batches (default: no dataset ships with this image).  It prints the kept detections; with --save_png it
also writes the annotated images like inference.py:71-101 (PIL only: the reference's seaborn 'hls' palette is rebuilt with
colorsys).  Like the reference, BatchNorm stays in train mode (no net.eval() anywhere)."""
import argparse
import colorsys
from pathlib import Path

import torch

from . import synth
from .model import SSD
from .utils import LabelMap, decode_score_, non_maximum_suppression


def detections(outputs: torch.Tensor):
    """Per image the rows NMS kept with a non-void arg-max class: (row indices, class ids 1..C-1, scores) on the CPU
    (inference.py:73-79: `torch.max(bbox_confs[i], dim=1)`, rows whose class is 0 = void are skipped)."""
    confs, ids = outputs[:, :, 4:].max(dim=2)
    keep = (ids > 0) & (confs > 0)
    confs, ids, keep = confs.cpu(), ids.cpu(), keep.cpu()
    out = []
    for i in range(outputs.shape[0]):
        rows = torch.nonzero(keep[i]).flatten()
        rows = rows[torch.argsort(confs[i, rows], descending=True)]
        out.append((rows, ids[i, rows], confs[i, rows]))
    return out


def draw(image_chw: torch.Tensor, boxes: torch.Tensor, ids, scores, labelmap, imsize: int, num_colors: int):
    """inference.py:72-99: one rectangle + label per kept detection on the input image."""
    from PIL import Image, ImageDraw
    image = Image.fromarray((image_chw.permute(1, 2, 0).cpu().numpy() * 255).astype('uint8'))
    pen = ImageDraw.Draw(image)
    for loc, cid, score in zip(boxes, ids, scores):
        cx, cy, w, h = [v.item() for v in loc * imsize]
        xmin, ymin, xmax, ymax = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2
        left_top = (max(xmin, 0), max(ymin, 0))
        right_bottom = (min(xmax, imsize), min(ymax, imsize))
        if right_bottom[0] < left_top[0] or right_bottom[1] < left_top[1]:
            continue
        text = f' {labelmap.id2name(int(cid) - 1)} {round(float(score), 3)}'
        text_loc = (max(xmin, 0), max(ymin, 0) - 11)
        color = tuple(int(c * 255) for c in colorsys.hls_to_rgb(int(cid) / num_colors, 0.6, 0.65))
        pen.rectangle(left_top + right_bottom, outline=color)
        pen.rectangle(text_loc + (text_loc[0] + len(text) * 6, max(ymin, 0)), fill=color, outline=color)
        pen.text(text_loc, text, fill=(0, 0, 0))
    return image


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--imsize', type=int, default=300)
    parser.add_argument('--batch_size', type=int, default=2)
    parser.add_argument('--num_workers', type=int, default=8)
    parser.add_argument('--result_dir', type=str, default='./result')
    parser.add_argument('--weights', type=str, default='weights.pth')
    parser.add_argument('--dtype', type=str, default='fp32', choices=('fp32', 'bf16'), help='arithmetic of the convolutions')
    parser.add_argument('--batches', type=int, default=1, help='synthetic source only')
    parser.add_argument('--data_dirs', type=str, nargs='*', default=None, help='VOC roots (inference.py:31: /work/data/VOCdevkit/VOC2007)')
    parser.add_argument('--data_list', type=str, default='test.txt')
    parser.add_argument('--save_png', action='store_true', help='write <result_dir>/detection/NNNNNN.png like inference.py:101')
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
        if args.data_dirs:                                # inference.py:25-40: ToTensor only, batches through collate_fn
            from . import augmentation as aug
            from .dataset import PascalVOCDataset
            from .utils import collate_fn
            data = PascalVOCDataset('detection', args.data_dirs, args.data_list, args.imsize, transform=aug.Compose([aug.ToTensor()]))
            for images, _ in torch.utils.data.DataLoader(data, batch_size=args.batch_size, num_workers=args.num_workers, collate_fn=collate_fn):
                yield images.to(device)
            return
        for b in range(args.batches):
            yield torch.from_numpy(synth.images(1000 + b, args.batch_size, args.imsize)).to(device)

    n = 1
    with torch.no_grad():
        for images in batches():
            outputs = net(images)
            decode_score_(outputs, defaults)                      # inference.py:67-68
            outputs = non_maximum_suppression(outputs=outputs)    # inference.py:69
            for i, (rows, ids, scores) in enumerate(detections(outputs)):
                top = [f'{labelmap.id2name(int(c) - 1)} {float(s):.3f}' for c, s in zip(ids[:5], scores[:5])]
                print(f'image {n}: {len(rows)} detections, top: {top}')
                if args.save_png:
                    draw(images[i], outputs[i, rows.to(device), :4].cpu(), ids, scores, labelmap, args.imsize, 21).save(out_dir / f'{n:06}.png')
                n += 1
    print('Finished Inference')


if __name__ == '__main__':
    main()
