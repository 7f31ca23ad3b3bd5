"""MI355X-native ray/scene intersection path behind a C ABI (include/mrt.h).

``binding`` wraps libmrt.so with ctypes; ``scene_io`` reads and writes the flattened scene
container.  The compute path is HIP only -- importing this package never falls back to a CPU
implementation.
"""
from . import binding, multigpu, scene_io  # noqa: F401
from .binding import Camera, MrtError, Raytracer, MODE_PACKET, MODE_RECURSIVE, MODE_BVH_DEPTH  # noqa: F401
