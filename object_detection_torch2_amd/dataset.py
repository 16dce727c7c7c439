"""Drop-in for the reference's `dataset` module (src/dataset.py): `PascalVOCDataset(purpose, data_dirs,
data_list_file_name, imsize, transform)` and `Purpose`, same list files / directory layout / ground-truth contract.

Decode + resize stay on the host with PIL exactly like the reference (dataset.py:43,49: `Image.open(...).resize((S, S))`,
PIL's default filter); everything after that -- colour jitter, flip, uint8 -> float, erasing -- is deferred to one device
pass per batch (augmentation.py).  Ground truth rows are `[cx, cy, w, h, onehot(C + 1)]` with column 4 = void and the
label's one-hot at column `4 + 1 + id` (dataset.py:92-116)."""
import enum
import xml.etree.ElementTree as ET
from pathlib import Path

import torch
from torch.utils.data import Dataset

from .utils import LabelMap


class Purpose(enum.Enum):
    CLASSIFICATION = 'classification'
    DETECTION = 'detection'

    @classmethod
    def get_all(cls):
        return [m.value for m in cls]


def _int(node, tag):
    return int(node.find(tag).text)


def _corners(obj):
    box = obj.find('bndbox')
    return _int(box, 'xmin'), _int(box, 'ymin'), _int(box, 'xmax'), _int(box, 'ymax')


class PascalVOCDataset(Dataset):
    def __init__(self, purpose: str, data_dirs, data_list_file_name: str, imsize: int, transform=None):
        if purpose not in Purpose.get_all():
            raise ValueError(f'purpose "{purpose}" is invalid')      # dataset.py:26-27
        self.purpose, self.imsize, self.transform = purpose, imsize, transform
        self.labelmap = LabelMap('PascalVOC')
        self.num_classes = len(self.labelmap)
        self.data_list = self._get_list(data_dirs, data_list_file_name)

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, i):
        from PIL import Image
        size = (self.imsize, self.imsize)
        if self.purpose == Purpose.CLASSIFICATION.value:
            label, corners, image_path = self.data_list[i]
            image = Image.open(image_path).crop(corners).resize(size)
            gt = torch.zeros(self.num_classes)
            gt[label] = 1.0
        else:
            image_path, anno_path = self.data_list[i]
            image = Image.open(image_path).resize(size)
            gt = self._get_gt(anno_path)
        if self.transform:
            image, gt = self.transform(image, gt)
        return image, gt

    def _get_list(self, data_dirs, data_list_file_name):
        """dataset.py:56-90: detection -> [[image_path, annotation_path]]; classification -> one [id, box, image_path] per object."""
        dirs = [data_dirs] if isinstance(data_dirs, (str, Path)) else list(data_dirs)
        items = []
        for d in map(Path, dirs):
            with open(d / 'ImageSets' / 'Main' / data_list_file_name, 'r') as f:
                ids = f.read().split('\n')[:-1]          # the list files end with a newline (dataset.py:77)
            for name in ids:
                image_path, anno_path = d / 'JPEGImages' / f'{name}.jpg', d / 'Annotations' / f'{name}.xml'
                if self.purpose == Purpose.DETECTION.value:
                    items.append([image_path, anno_path])
                else:
                    for obj in ET.parse(anno_path).getroot().iter('object'):
                        items.append([self.labelmap.name2id(obj.find('name').text), _corners(obj), image_path])
        return items

    def _get_gt(self, anno_path) -> torch.Tensor:
        """(G, 4 + C + 1) rows, centre form normalised by the annotated image size."""
        root = ET.parse(anno_path).getroot()
        size = root.find('size')
        width, height = _int(size, 'width'), _int(size, 'height')
        objects = list(root.iter('object'))
        gt = torch.zeros(len(objects), 4 + self.num_classes + 1)
        for row, obj in zip(gt, objects):
            xmin, ymin, xmax, ymax = _corners(obj)
            row[0], row[1] = (xmin + xmax) / 2 / width, (ymin + ymax) / 2 / height
            row[2], row[3] = (xmax - xmin) / width, (ymax - ymin) / height
            row[4 + 1 + self.labelmap.name2id(obj.find('name').text)] = 1.0
        return gt
