"""MI355X-native drop-in for the transformer part of the reference's ``models`` package.

The reference's models/__init__.py:1-7 re-exports eight names; the ones on the ViT / Swin training hot
path are provided here (SwinTransformer, dino) plus VisionTransformer.  The convolutional families and
the Halo / PVT / Twins models are outside this build's scope.
"""
from .swin_transformer import SwinTransformer
from .vit import VisionTransformer, dino

__all__ = ["SwinTransformer", "VisionTransformer", "dino"]
