"""nanorepeat_amd -- MI355X-native implementation of NanoRepeat's per-read repeat-size
estimation hot path (1D round 3 and the 2D joint grid), behind a C ABI
(include/nanorepeat_amd.h) loaded with ctypes.  HIP/gfx950 only; no CPU fallback."""
__version__ = "0.1.0"
