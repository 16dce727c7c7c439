"""Inference entry point with the reference's flags (src/inference.py:14-20): forward -> decode -> score -> NMS
(inference.py:65-69) on synthetic batches; prints the kept detections instead of drawing them (PIL/seaborn drawing,
inference.py:71-101, is visualisation and outside the path)."""
import argparse
from pathlib import Path

import torch

from . import synth
from .model import SSD
from .utils import LabelMap, decode_score_, non_maximum_suppression


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--imsize', type=int, default=300)
    parser.add_argument('--batch_size', type=int, default=2)
    parser.add_argument('--num_workers', type=int, default=8)
    parser.add_argument('--result_dir', type=str, default='./result')
    parser.add_argument('--weights', type=str, default='weights.pth')
    parser.add_argument('--batches', type=int, default=1)
    args = parser.parse_args()

    device = torch.device('cuda:0')
    net = SSD(num_classes=20 + 1, weights_path=Path(args.result_dir) / 'detection' / args.weights,
              weights_path_vgg16=Path(args.result_dir) / 'classification' / args.weights)
    net.to(device)
    defaults = net.default_bboxes.to(device)
    labelmap = LabelMap('PascalVOC')
    with torch.no_grad():          # like the reference, BatchNorm stays in train mode (no net.eval() anywhere)
        for b in range(args.batches):
            images = torch.from_numpy(synth.images(1000 + b, args.batch_size, args.imsize)).to(device)
            outputs = net(images)
            decode_score_(outputs, defaults)                      # inference.py:67-68
            outputs = non_maximum_suppression(outputs=outputs)    # inference.py:69
            confs, ids = outputs[:, :, 4:].max(dim=2)
            for i in range(len(images)):
                keep = (ids[i] > 0) & (confs[i] > 0)
                top = torch.argsort(confs[i] * keep, descending=True)[:5]
                dets = [f'{labelmap.id2name(ids[i, j].item() - 1)} {confs[i, j].item():.3f}' for j in top if keep[j]]
                print(f'image {b * args.batch_size + i}: {int(keep.sum())} detections, top: {dets}')
    print('Finished Inference')


if __name__ == '__main__':
    main()
