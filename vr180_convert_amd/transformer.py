"""Module-name parity with the reference: ``from vr180_convert.transformer import ...`` becomes
``from vr180_convert_amd.transformer import ...`` (reference tests/test_remapper.py:25-30,
cli.py:20).  The implementation lives in :mod:`vr180_convert_amd.chain`."""
from .chain import (  # noqa: F401
    DenormalizeTransformer,
    EquirectangularDecoder,
    EquirectangularEncoder,
    Euclidean3DRotator,
    Euclidean3DTransformer,
    FisheyeDecoder,
    FisheyeEncoder,
    InverseTransformer,
    MultiTransformer,
    NormalizeTransformer,
    NotLowerable,
    PolarRollTransformer,
    PolynomialScaler,
    RectilinearDecoder,
    TransformerBase,
    ZoomTransformer,
    equidistant_from_3d,
    equidistant_to_3d,
    get_radius,
)
