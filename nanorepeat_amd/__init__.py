"""nanorepeat_amd -- MI355X-native implementation of NanoRepeat's per-read repeat-size
estimation hot path (1D round 3 and the 2D joint grid), behind a C ABI
(include/nanorepeat_amd.h) loaded with ctypes.  HIP/gfx950 only; no CPU fallback."""
__version__ = "0.1.0"

import os as _os

# The library's default number of HIP hardware queues (csrc/nra_host.cpp, g_hw_queues_default), set here as well
# so that it also holds when something imported after this package -- torch in dist.py -- starts the HIP runtime
# before the library is loaded.  Only a default: a value already in the environment stays.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
