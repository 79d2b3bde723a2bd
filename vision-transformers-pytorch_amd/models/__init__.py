"""MI355X-native drop-in for the transformer part of the reference's ``models`` package.

The reference's models/__init__.py:1-7 re-exports eight names; the ones on the ViT / Swin training hot
path are provided here (SwinTransformer, dino) plus VisionTransformer and -- SURVEY.md section 8 row F1 -- the
PyramidVisionTransformer of models/pvt.py (the reference does not re-export it either; import models.pvt).  The
convolutional families and the Halo / Twins models are outside this build's scope.
"""
from .pvt import PyramidVisionTransformer
from .swin_transformer import SwinTransformer
from .vit import VisionTransformer, dino

__all__ = ["SwinTransformer", "VisionTransformer", "dino", "PyramidVisionTransformer"]
