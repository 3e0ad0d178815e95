"""Runs the REFERENCE's own OpenCL kernels on the GPU (test infrastructure).

oracle/ref_kernels.mk compiles /root/reference/kernels/CL/{intersect_bvh2_lds,build_hlbvh,intersect_bvh2level_skiplinks}.cl
with the in-image ROCm clang into gfx950 code objects under oracle/_ref/ (built in the build container, where
/root/reference exists; they travel to the GPU box with the snapshot).  This module loads them with hipModuleLoad and
launches them with hipModuleLaunchKernel through oracle/libref_launch.so (oracle/ref_launch.cpp).  It is imported by
tests only; nothing under aobaker_amd/ knows it exists.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
_LIB = None

BBOX = np.dtype([("pmin", "<f4", 4), ("pmax", "<f4", 4)])                                      # common.cl:46-50
HLBVH_NODE = np.dtype([("parent", "<i4"), ("left", "<i4"), ("right", "<i4"), ("next", "<i4")])  # build_hlbvh.cl:54-60


def code_object(name: str) -> str:
    path = os.path.join(REF_DIR, name)
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: build it where /root/reference exists with `make -C oracle -f ref_kernels.mk` "
            "(python __graft_entry__.py does); the code objects travel to the GPU box with the snapshot")
    return path


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        from aobaker_amd import capi
        capi.load()          # one HIP runtime per process: the library's loader picks it
        path = os.path.join(ROOT, "oracle", "libref_launch.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: `make -C oracle libref_launch.so`")
        l = C.CDLL(path)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        l.ref_last_error.restype = C.c_char_p
        l.ref_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        l.ref_unload.argtypes = [vp]
        l.ref_alloc.argtypes = [u64, C.POINTER(vp)]
        l.ref_free.argtypes = [vp]
        l.ref_h2d.argtypes = [vp, vp, u64]
        l.ref_d2h.argtypes = [vp, vp, u64]
        l.ref_memset.argtypes = [vp, C.c_int, u64]
        l.ref_launch.argtypes = [vp, C.c_char_p, u32, u32, C.POINTER(vp), u32, C.POINTER(C.c_float)]
        _LIB = l
    return _LIB


def _check(rc):
    if rc != 0:
        raise RuntimeError("reference kernel launcher: " + lib().ref_last_error().decode())


class DeviceBuffer:
    def __init__(self, host: np.ndarray | None = None, nbytes: int | None = None, fill: int | None = None):
        self.nbytes = int(host.nbytes if host is not None else nbytes)
        self.ptr = C.c_void_p()
        _check(lib().ref_alloc(self.nbytes, C.byref(self.ptr)))
        if host is not None:
            host = np.ascontiguousarray(host)
            _check(lib().ref_h2d(self.ptr, host.ctypes.data, host.nbytes))
        elif fill is not None:
            _check(lib().ref_memset(self.ptr, fill, self.nbytes))

    def read(self, dtype, count) -> np.ndarray:
        out = np.zeros(count, dtype=dtype)
        _check(lib().ref_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().ref_free(self.ptr)
            self.ptr = C.c_void_p()


class Module:
    """One code object of oracle/_ref/."""

    def __init__(self, name: str):
        self.handle = C.c_void_p()
        _check(lib().ref_load(code_object(name).encode(), C.byref(self.handle)))
        self.last_ms = 0.0

    def close(self):
        if self.handle:
            lib().ref_unload(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    def launch(self, kernel: str, work_items: int, args, block: int = 64):
        """args: DeviceBuffer (a __global pointer) or a ctypes scalar (by value), in the kernel's parameter order."""
        holders = [C.c_void_p(a.ptr.value) if isinstance(a, DeviceBuffer) else a for a in args]
        argv = (C.c_void_p * len(holders))(*[C.addressof(h) for h in holders])
        ms = C.c_float()
        blocks = max(1, (int(work_items) + block - 1) // block)
        _check(lib().ref_launch(self.handle, kernel.encode(), blocks, block, argv, len(holders), C.byref(ms)))
        self.last_ms = ms.value


# ---- intersect_bvh2_lds.cl -------------------------------------------------------------------------------------

def intersect(module: Module, nodes: np.ndarray, rays: np.ndarray, fill_hits: np.ndarray):
    """intersect_main (intersect_bvh2_lds.cl:65-226): nodes = AobBvhNode[] (the reference's bvh_node layout), rays =
    48-byte records, returns the 32-byte Intersection records (inactive rays keep fill_hits' content)."""
    n = len(rays)
    bufs = [DeviceBuffer(nodes), DeviceBuffer(rays), DeviceBuffer(np.array([n], dtype=np.int32)),
            DeviceBuffer(nbytes=32 * 4 * max(1, n), fill=0), DeviceBuffer(fill_hits)]       # stack: STACK_SIZE 32 uints per ray
    try:
        module.launch("intersect_main", n, bufs)
        return bufs[4].read(fill_hits.dtype, len(fill_hits))
    finally:
        for b in bufs:
            b.free()


def occluded(module: Module, nodes: np.ndarray, rays: np.ndarray, fill_hits: np.ndarray):
    """occluded_main (intersect_bvh2_lds.cl:228-363): 1 (hit) / -1 (miss) per active ray."""
    n = len(rays)
    bufs = [DeviceBuffer(nodes), DeviceBuffer(rays), DeviceBuffer(np.array([n], dtype=np.int32)),
            DeviceBuffer(nbytes=32 * 4 * max(1, n), fill=0), DeviceBuffer(fill_hits)]
    try:
        module.launch("occluded_main", n, bufs)
        return bufs[4].read(np.int32, len(fill_hits))
    finally:
        for b in bufs:
            b.free()


# ---- build_hlbvh.cl ----------------------------------------------------------------------------------------------

def morton_codes(module: Module, prim_boxes: np.ndarray, scene_box: np.ndarray) -> np.ndarray:
    """calculate_morton_code_main (build_hlbvh.cl:101-125): 30-bit codes of the primitive boxes' centres."""
    n = len(prim_boxes)
    bufs = [DeviceBuffer(prim_boxes), DeviceBuffer(scene_box), DeviceBuffer(nbytes=4 * max(1, n), fill=0)]
    try:
        module.launch("calculate_morton_code_main", n, [bufs[0], C.c_int32(n), bufs[1], bufs[2]])
        return bufs[2].read(np.int32, n)
    finally:
        for b in bufs:
            b.free()


def hierarchy(module: Module, sorted_codes: np.ndarray, prim_boxes: np.ndarray, sorted_indices: np.ndarray):
    """emit_hierarchy_main (:216-261) then refit_bounds_main (:264-314): (HlbvhNode[2n-1], bbox[2n-1])."""
    n = len(sorted_codes)
    bufs = [DeviceBuffer(np.ascontiguousarray(sorted_codes, dtype=np.int32)), DeviceBuffer(prim_boxes),
            DeviceBuffer(np.ascontiguousarray(sorted_indices, dtype=np.int32)),
            DeviceBuffer(nbytes=16 * (2 * n - 1), fill=0xff), DeviceBuffer(nbytes=32 * (2 * n - 1), fill=0),
            DeviceBuffer(nbytes=4 * (2 * n - 1), fill=0)]
    try:
        module.launch("emit_hierarchy_main", n, [bufs[0], bufs[1], bufs[2], C.c_int32(n), bufs[3], bufs[4]])
        module.launch("refit_bounds_main", n, [bufs[4], C.c_int32(n), bufs[3], bufs[5]])
        return bufs[3].read(HLBVH_NODE, 2 * n - 1), bufs[4].read(BBOX, 2 * n - 1)
    finally:
        for b in bufs:
            b.free()
