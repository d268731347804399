"""nndepth_amd — MI355X (gfx950) native stereo-disparity hot path for nndepth's RAFT-Stereo family.

Everything computational lives in libnndepth_amd.so (hand-written HIP, C-ABI in
include/nndepth_amd.h); this package is the Python-side mirror of the reference's seams.
Importing it requires the built library — there is no fallback path.
"""
from . import weightgen  # noqa: F401  (pure numpy/torch, needs no GPU)

__all__ = ["weightgen"]


def __getattr__(name):
    # lazy: `import nndepth_amd` must not dlopen on tooling that only wants weightgen
    if name in ("ops", "cost_volume", "blocks", "upsample", "encoder", "raft_stereo", "_lib"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
