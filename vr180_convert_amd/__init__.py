"""MI355X-native drop-in for the hot path of 34j/vr180-convert: ``apply`` / ``apply_lr`` /
``get_map`` and the Transformer classes (reference src/vr180_convert/__init__.py:1-32)."""
__version__ = "0.1.0"

from .chain import (
    DenormalizeTransformer,
    EquirectangularEncoder,
    Euclidean3DRotator,
    Euclidean3DTransformer,
    FisheyeDecoder,
    FisheyeEncoder,
    MultiTransformer,
    NormalizeTransformer,
    PolarRollTransformer,
    TransformerBase,
    ZoomTransformer,
)
from .remapper import (anaglyph_tensors, apply, apply_lr, apply_lr_tensors, auto_radius_tensor, get_map, remap_tensors,
                       remap_tensors_auto)
from .sharding import remap_sharded

__all__ = [
    "TransformerBase",
    "ZoomTransformer",
    "MultiTransformer",
    "NormalizeTransformer",
    "PolarRollTransformer",
    "DenormalizeTransformer",
    "FisheyeDecoder",
    "FisheyeEncoder",
    "EquirectangularEncoder",
    "Euclidean3DRotator",
    "Euclidean3DTransformer",
    "apply",
    "apply_lr",
    "get_map",
    # additions of this engine (device-resident entry points)
    "apply_lr_tensors",
    "anaglyph_tensors",
    "remap_tensors",
    "remap_tensors_auto",
    "auto_radius_tensor",
    "remap_sharded",
]
