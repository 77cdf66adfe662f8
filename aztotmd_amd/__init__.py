"""aztotmd_amd - MI355X-native (gfx950, fp64) implementation of the azTotMD per-step hot path.

cell-list build / counting sort  ->  pair VdW + short-range Coulomb  ->  velocity Verlet (+ radiative thermostat)

The compute lives in `libaztot.so` (hand-written HIP kernels + C++ host side, built from `csrc/`) behind the
C ABI declared in `include/aztot.h`.  This package is only a thin ctypes front-end plus the generator of the
benchmark inputs; there is no CPU fallback: every compute entry point fails loudly without the HIP library
and a GPU.
"""
from .api import Engine, Model, AztotError, library_path, build_library  # noqa: F401
