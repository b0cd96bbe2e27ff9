"""nanorepeat_amd -- MI355X-native implementation of NanoRepeat's per-read repeat-size
estimation hot path (1D round 3 and the 2D joint grid), behind a C ABI
(include/nanorepeat_amd.h) loaded with ctypes.  HIP/gfx950 only; no CPU fallback."""
__version__ = "0.1.0"

# What a host process may want in its environment BEFORE anything starts the HIP runtime.  Neither the package
# nor the shared library touches the environment on its own: an entry point (bench.py, a CLI) opts in.
RECOMMENDED_ENV = {
    # a batch runs every rows-per-lane bucket's kernels on a stream of its own; with HIP's default of 4 hardware
    # queues short kernels and copies wait behind another stream's long sweeps (config 3: 25.6 -> 19.2 ms per run)
    "GPU_MAX_HW_QUEUES": "8",
}


def apply_recommended_env():
    """Opt-in, for entry points: sets RECOMMENDED_ENV where the variable is not set yet.  Only effective before the
    HIP runtime starts (before torch.cuda / the first nra_* call); call it from the main thread at start-up."""
    import os
    for key, value in RECOMMENDED_ENV.items():
        os.environ.setdefault(key, value)
