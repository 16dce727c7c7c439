from .ssd import SSD  # noqa: F401  (reference: src/model/__init__.py:3-4 exports VGG16 and SSD; only SSD is on the hot path)
