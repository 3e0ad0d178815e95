"""ctypes mirror of include/aobaker.h and loader of the HIP library.

The library is the product: there is no CPU fallback.  `load()` raises if
`libaobaker_hip.so` has not been built (`python __graft_entry__.py` or
`make -C aobaker_amd/csrc`), and every wrapper raises `AobError` with the
library's own `aob_last_error()` text when a call fails.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

ABI_VERSION = 7

# ---- enums (include/aobaker.h) ------------------------------------------------
AOB_OK = 0
SAMPLER_REFERENCE_COMPAT = 0
SAMPLER_COSINE = 1
ACCUMULATE_QUANTISED = 0
ACCUMULATE_FLOAT = 1
FP16_OFF = 0
FP16_RTZ = 1
FP16_RNE = 2
BLEED_REFERENCE = 0
BLEED_NEAREST = 1
RASTER_REFERENCE = 0
RASTER_CONSERVATIVE = 1
LAYERS_LAST = 0
LAYERS_ALL = 1
MESH_HIDDEN_FROM_SELF = 1
MESH_HIDDEN_FROM_OTHERS = 2
INVALID_ADDR = 0xFFFFFFFF
UINT64_MAX = 0xFFFFFFFFFFFFFFFF


class AobOptions(C.Structure):
    _fields_ = [
        ("sample_count", C.c_int32),
        ("sampler", C.c_int32),
        ("seed", C.c_uint32),
        ("use_any_hit", C.c_int32),
        ("falloff_distance", C.c_float),
        ("ray_tmax", C.c_float),
        ("ray_bias", C.c_float),
        ("bleed_distance", C.c_int32),
        ("accumulate", C.c_int32),
        ("normal_fp16", C.c_int32),
        ("ray_chunk_size", C.c_int32),
        ("clamp_tmax_to_falloff", C.c_int32),
        ("bvh_width", C.c_int32),
        ("bvh_quantised", C.c_int32),
        ("bvh_builder", C.c_int32),
        ("bleed_mode", C.c_int32),
        ("material_ranges", C.c_int32),
        ("morton_adaptive", C.c_int32),
        ("raster_mode", C.c_int32),
        ("raster_layers", C.c_int32),
        ("trace_refill_idle", C.c_int32),
        ("trace_leaf_wait", C.c_int32),
        ("trace_parts", C.c_int32),
        ("trace_part_min", C.c_int32),
        ("trace_fine_div", C.c_int32),
        ("trace_bands", C.c_int32),
        ("trace_scalar_min", C.c_int32),
        ("trace_cut", C.c_int32),
        ("trace_horizon", C.c_int32),
    ]


class AobMeshDesc(C.Structure):
    _fields_ = [
        ("positions", C.c_void_p),
        ("normals", C.c_void_p),
        ("uvs", C.c_void_p),
        ("position_stride", C.c_uint32),
        ("normal_stride", C.c_uint32),
        ("uv_stride", C.c_uint32),
        ("vertex_count", C.c_uint32),
        ("indices", C.c_void_p),
        ("index_count", C.c_uint32),
        ("material", C.c_int32),
        ("model", C.c_float * 16),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class AobMaterialDesc(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("channels", C.c_int32),
        ("bytes_per_channel", C.c_int32),
        ("data", C.c_void_p),
        ("ao_max_range", C.c_float),
        ("reserved", C.c_int32),
    ]


class AobSceneDesc(C.Structure):
    _fields_ = [
        ("meshes", C.POINTER(AobMeshDesc)),
        ("mesh_count", C.c_uint32),
        ("materials", C.POINTER(AobMaterialDesc)),
        ("material_count", C.c_uint32),
    ]


class AobDeviceConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32 * 6)]


class AobStats(C.Structure):
    _fields_ = [
        ("triangle_count", C.c_uint64),
        ("node_count", C.c_uint64),
        ("surfel_count", C.c_uint64),
        ("ray_count", C.c_uint64),
        ("bvh_depth", C.c_uint32),
        ("trace_cuts", C.c_uint32),
        ("ms_upload", C.c_float),
        ("ms_bvh_build", C.c_float),
        ("ms_rasterize", C.c_float),
        ("ms_trace", C.c_float),
        ("ms_resolve", C.c_float),
        ("ms_download", C.c_float),
        ("ms_total_device", C.c_float),
        ("reserved1", C.c_float),
        ("rays_cast", C.c_uint64),
        ("trace_horizon", C.c_uint32),
        ("reserved2", C.c_uint32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if not n.startswith("reserved")}


# numpy dtypes of the device records (AobSurfel 40 B, AobBvhNode 64 B, AobResult 8 B)
def surfel_dtype():
    import numpy as np
    return np.dtype([("texel_x", "<i4"), ("texel_y", "<i4"), ("position", "<f4", 3),
                     ("texcoord", "<f4", 2), ("normal", "<f4", 3)])


def node_dtype():
    import numpy as np
    return np.dtype([("lmin_or_v0", "<f4", 3), ("addr_left", "<u4"),
                     ("lmax_or_v1", "<f4", 3), ("shape_id", "<u4"),
                     ("rmin_or_v2", "<f4", 3), ("addr_right", "<u4"),
                     ("rmax", "<f4", 3), ("prim_id", "<u4")])


def node4_dtype():
    import numpy as np
    return np.dtype([("lo", "<f4", (4, 4)), ("hi", "<f4", (4, 4))])


def leaf_tri_dtype():
    import numpy as np
    return np.dtype([("v0", "<f4", 4), ("v1", "<f4", 4), ("v2", "<f4", 4)])


def result_dtype():
    import numpy as np
    return np.dtype([("total_bits", "<u4"), ("weight_bits", "<u4")])


# every symbol include/aobaker.h declares; tests/test_capi_symbols.py checks the
# built library exports exactly these
EXPORTED_SYMBOLS = [
    "aob_init", "aob_shutdown", "aob_default_options", "aob_bake", "aob_last_error",
    "aob_abi_version", "aob_resolve_bvh_builder",
    "aob_bake_create", "aob_bake_destroy", "aob_bake_build_bvh", "aob_bake_rasterize",
    "aob_bake_trace", "aob_bake_resolve", "aob_bake_download", "aob_bake_run_device",
    "aob_bake_sync", "aob_bake_get_stats", "aob_bake_clear_results",
    "aob_bake_node_count", "aob_bake_copy_nodes", "aob_bake_surfel_count",
    "aob_bake_copy_surfels", "aob_bake_copy_results", "aob_bake_results_device_ptr",
    "aob_bake_image_device_ptr", "aob_bake_stream", "aob_bake_last_trace_ms",
    "aob_bake_shard_bounds", "aob_bake_node4_count", "aob_bake_copy_nodes4", "aob_bake_copy_leaf_tris", "aob_bake_add_results", "aob_bake_add_results_from", "aob_bake_copy_tile_cuts", "aob_bake_copy_horizon", "aob_bake_horizon_counters", "aob_bake_last_stage_ms", "aob_debug_fail_tile_cuts",
    "aob_bake_copy_nodes2q", "aob_bake_quant_frame", "aob_bake_copy_nodes4q", "aob_bake_node4q_count", "aob_bake_rasterize_rows",
]


class AobError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"aobaker HIP library error {code}: {text}")
        self.code = code


_LIB = None


def library_path() -> str:
    # AOBAKER_HIP_LIBRARY: another build of the same library (kernel experiments); still no fallback
    return os.environ.get("AOBAKER_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libaobaker_hip.so")


def load() -> C.CDLL:
    """dlopen the HIP library (no GPU needed for the dlopen itself)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension is the product and there is no fallback. "
            "Build it with `python __graft_entry__.py` (or `make -C aobaker_amd/csrc`).")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 with the same
    # soname as /opt/rocm's.  Whichever is loaded first serves both, and torch cannot see the GPU
    # through the system copy, so when torch is installed (multi-GPU driver, bench) its copy is
    # pre-loaded here.  Without torch the library simply uses the system runtime.
    if "torch" not in sys.modules:
        try:
            spec = importlib.util.find_spec("torch")
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else None
            if cand and os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except (ImportError, OSError, ValueError):
            pass
    lib = C.CDLL(path)
    vp, i32, u64 = C.c_void_p, C.c_int32, C.c_uint64
    P = C.POINTER
    sig = {
        "aob_init": (C.c_int, [P(AobDeviceConfig)]),
        "aob_shutdown": (None, []),
        "aob_default_options": (None, [P(AobOptions)]),
        "aob_bake": (C.c_int, [P(AobSceneDesc), P(AobOptions), P(AobStats)]),
        "aob_last_error": (C.c_char_p, []),
        "aob_abi_version": (C.c_int, []),
        "aob_resolve_bvh_builder": (C.c_int, [P(AobOptions), u64, u64]),
        "aob_bake_create": (C.c_int, [P(AobSceneDesc), P(AobOptions), P(vp)]),
        "aob_bake_destroy": (None, [vp]),
        "aob_bake_build_bvh": (C.c_int, [vp]),
        "aob_bake_rasterize": (C.c_int, [vp, i32]),
        "aob_bake_rasterize_rows": (C.c_int, [vp, i32, i32, i32]),
        "aob_bake_trace": (C.c_int, [vp, i32, u64, u64]),
        "aob_bake_resolve": (C.c_int, [vp, i32]),
        "aob_bake_download": (C.c_int, [vp, i32, vp, C.c_size_t]),
        "aob_bake_run_device": (C.c_int, [vp]),
        "aob_bake_sync": (C.c_int, [vp]),
        "aob_bake_get_stats": (C.c_int, [vp, P(AobStats)]),
        "aob_bake_clear_results": (C.c_int, [vp, i32]),
        "aob_bake_node_count": (C.c_int, [vp, P(u64)]),
        "aob_bake_copy_nodes": (C.c_int, [vp, vp, u64]),
        "aob_bake_surfel_count": (C.c_int, [vp, i32, P(u64)]),
        "aob_bake_copy_surfels": (C.c_int, [vp, i32, vp, u64]),
        "aob_bake_copy_results": (C.c_int, [vp, i32, vp, u64]),
        "aob_bake_results_device_ptr": (C.c_int, [vp, i32, P(vp), P(u64)]),
        "aob_bake_image_device_ptr": (C.c_int, [vp, i32, P(vp), P(u64)]),
        "aob_bake_stream": (C.c_int, [vp, P(vp)]),
        "aob_bake_last_trace_ms": (C.c_int, [vp, P(C.c_float)]),
        "aob_bake_shard_bounds": (C.c_int, [vp, i32, C.c_uint32, P(u64), P(u64), P(u64), P(i32)]),
        "aob_bake_node4_count": (C.c_int, [vp, P(u64)]),
        "aob_bake_copy_nodes4": (C.c_int, [vp, vp, u64]),
        "aob_bake_copy_leaf_tris": (C.c_int, [vp, vp, u64]),
        "aob_bake_add_results": (C.c_int, [vp, C.c_int32, vp, u64]),
        "aob_bake_add_results_from": (C.c_int, [vp, C.c_int32, vp, C.c_int32]),
        "aob_bake_copy_tile_cuts": (C.c_int, [vp, C.c_int32, vp, u64, P(C.c_uint32), P(C.c_uint32)]),
        "aob_bake_copy_horizon": (C.c_int, [vp, C.c_int32, vp, u64, P(C.c_uint32), P(C.c_uint64)]),
        "aob_bake_horizon_counters": (C.c_int, [vp, vp, C.c_int]),
        "aob_bake_last_stage_ms": (C.c_int, [vp, vp]),
        "aob_debug_fail_tile_cuts": (None, [C.c_int]),
        "aob_bake_copy_nodes2q": (C.c_int, [vp, vp, u64]),
        "aob_bake_quant_frame": (C.c_int, [vp, vp]),
        "aob_bake_copy_nodes4q": (C.c_int, [vp, vp, u64]),
        "aob_bake_node4q_count": (C.c_int, [vp, P(u64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.aob_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {lib.aob_abi_version()} != {ABI_VERSION}")
    _LIB = lib
    return lib


def check(code: int) -> None:
    if code != AOB_OK:
        text = load().aob_last_error()
        raise AobError(code, text.decode("utf-8", "replace") if text else "")


def default_options() -> AobOptions:
    """The reference's compile-time constants (BakeEngine.cpp:13-18), from the library."""
    o = AobOptions()
    load().aob_default_options(C.byref(o))
    return o
