from .ssd import SSD  # noqa: F401  (reference: src/model/__init__.py:3-4 exports VGG16 and SSD)
from .vgg16 import VGG16  # noqa: F401  (classification pre-training; not on the detection hot path)
