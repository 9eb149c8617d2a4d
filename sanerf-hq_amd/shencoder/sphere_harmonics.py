"""Module path of the reference (shencoder/sphere_harmonics.py); the implementation lives in sanerf_hq_amd.ops."""
from ..ops import SHEncoder, _sh_encoder, sh_encode  # noqa: F401
