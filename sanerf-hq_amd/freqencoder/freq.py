"""Module path of the reference (freqencoder/freq.py); the implementation lives in sanerf_hq_amd.ops."""
from ..ops import FreqEncoder, _freq_encoder, freq_encode  # noqa: F401
