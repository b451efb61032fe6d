"""Import shim: makes the hyphenated package directory ``phi-3-vision-mlx_amd/``
importable as ``phi_3_vision_mlx_amd`` and re-exports the reference's public
API (`load/generate/choose/constrain/benchmark`, reference
phi_3_vision_mlx.py:1178-1487) so that
``from phi_3_vision_mlx_amd import generate`` is the drop-in for
``from phi_3_vision_mlx import generate``.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "phi-3-vision-mlx_amd")
__path__ = [_PKG_DIR]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__

from phi_3_vision_mlx_amd.api import (  # noqa: E402,F401
    ID_ASS, ID_EOS, LogitStopper, Streamer, TokenStopper, benchmark, choose,
    constrain, generate, load,
)
