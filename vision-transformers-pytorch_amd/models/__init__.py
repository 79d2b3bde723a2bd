"""MI355X-native drop-in for the transformer part of the reference's ``models`` package.

The reference's models/__init__.py:1-7 re-exports eight names; the ones on the ViT / Swin training hot path are provided
here (SwinTransformer, dino) plus VisionTransformer, the PyramidVisionTransformer of models/pvt.py (SURVEY.md section 8
row F1; models/__init__.py:3) and the TwinsSVT of models/twins.py (the row after F1-F4; the reference does not re-export
it -- ``from models.twins import TwinsSVT`` works in both) and, since round 5, the HaloTransformer of models/halo_transformer.py
(models/__init__.py:1; the row SURVEY.md section 8 ranks last).  The convolutional families are outside this build's scope
(SURVEY.md section 8).
"""
from .halo_transformer import HaloTransformer
from .pvt import PyramidVisionTransformer
from .swin_transformer import SwinTransformer
from .twins import TwinsSVT
from .vit import VisionTransformer, dino

__all__ = ["SwinTransformer", "VisionTransformer", "dino", "PyramidVisionTransformer", "TwinsSVT", "HaloTransformer"]
