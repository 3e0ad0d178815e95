"""MI355X-native ambient-occlusion bake path (drop-in for BakeEngine::Bake of epicabsol/aobaker).

The product is the HIP library `aobaker_amd/csrc/libaobaker_hip.so` behind the C ABI of
`include/aobaker.h`; this package is the thin host side over it (ctypes, no torch types).
Importing the package does not load the library; the first call does, and raises if it has
not been built — there is no CPU fallback.
"""
from . import capi                                   # noqa: F401
from .bake_engine import BakeEngine, Baker           # noqa: F401
from .scene import (BakeObject, BakeTexture, FlatScene, MaterialObject, RenderMesh, Scene,   # noqa: F401
                    Section)

__all__ = ["capi", "BakeEngine", "Baker", "Scene", "BakeObject", "Section", "RenderMesh",
           "MaterialObject", "BakeTexture", "FlatScene"]
