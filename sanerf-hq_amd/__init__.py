"""sanerf-hq_amd — MI355X (gfx950) implementation of SANeRF-HQ's volumetric-rendering hot path.

Layout
  csrc/            hand-written HIP kernels + the C ABI (libsanerf_hip.so, include/sanerf_hip.h)
  _lib.py          ctypes binding (the only native boundary; no CPU fallback)
  ops.py          grid / SH / frequency encoders (autograd Functions + modules) over the C ABI
  gridencoder/ shencoder/ freqencoder/ raymarching/ encoding.py activation.py
                   the reference's operator modules, same names and call signatures
  nerf/            NeRFRenderer / NeRFNetwork / get_rays with the reference's contracts
  dist.py          ray-tile sharding of one image over the GPUs of a node + RCCL all-gather
  synth.py         deterministic synthetic inputs for bench / smoke / tests

The directory name carries a hyphen, so it is imported as `sanerf_hq_amd` through the shim
at the repository root (sanerf_hq_amd.py).  `install_dropin()` additionally registers the
operator modules under the reference's top-level names, so the reference's own
`nerf/network.py` (`from encoding import get_encoder`, ...) runs on these kernels unmodified.
"""
import sys

from . import _lib, synth  # noqa: F401

__all__ = ["build", "install_dropin", "native_library_path"]


def build(force: bool = False) -> str:
    return _lib.build(force)


def native_library_path() -> str:
    return _lib.LIB_PATH


def install_dropin() -> None:
    """Make `import gridencoder / shencoder / freqencoder / raymarching / encoding / activation`
    resolve to this package (what the reference's encoding.py:60-74 and nerf/network.py:5-7 import)."""
    from . import activation, encoding, freqencoder, gridencoder, raymarching, shencoder
    for name, mod in (("gridencoder", gridencoder), ("shencoder", shencoder), ("freqencoder", freqencoder),
                      ("raymarching", raymarching), ("encoding", encoding), ("activation", activation)):
        sys.modules[name] = mod
