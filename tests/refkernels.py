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


# ---- intersect_bvh2level_skiplinks.cl ------------------------------------------------------------------------------

SHAPE = np.dtype([("id", "<i4"), ("bvh_idx", "<i4"), ("disabled", "<u4"), ("padding", "<i4"), ("m", "<f4", (4, 4)),
                  ("velocity_linear", "<f4", 4), ("velocity_angular", "<f4", 4)])          # Shape :75-92, 112 B
FACE = np.dtype([("idx", "<i4", 3), ("shape_id", "<i4"), ("prim_id", "<i4")])              # Face :94-102, 20 B


def _skiplinks(nodes2, out, payload_of_leaf):
    """Appends the BVH2 `nodes2` (AobBvhNode[]: child boxes in the parent, leaves at [n-1, 2n-1)) to `out` in the layout
    intersect_bvh2level_skiplinks.cl walks (:41-47): one bbox per node, the left child right behind its parent, pmin.w = -1
    for an inner node or (payload << 4 | 1) for a leaf, pmax.w = the node to go to after this subtree (-1: none).
    Returns the root's index in `out`."""
    n = (len(nodes2) + 1) // 2
    leaf_base = n - 1
    base = len(out)
    if n == 1:
        v = np.stack([nodes2[0]["lmin_or_v0"], nodes2[0]["lmax_or_v1"], nodes2[0]["rmin_or_v2"]])
        out.append((v.min(axis=0), float((payload_of_leaf(0) << 4) | 1), v.max(axis=0), -1.0))
        return base
    size = np.ones(len(nodes2), dtype=np.int64)
    order, stack = [], [0]
    while stack:
        i = stack.pop()
        order.append(i)
        if i < leaf_base:
            stack += [int(nodes2[i]["addr_left"]), int(nodes2[i]["addr_right"])]
    for i in reversed(order):
        if i < leaf_base:
            size[i] = 1 + size[int(nodes2[i]["addr_left"])] + size[int(nodes2[i]["addr_right"])]
    root = nodes2[0]
    root_box = (np.minimum(root["lmin_or_v0"], root["rmin_or_v2"]), np.maximum(root["lmax_or_v1"], root["rmax"]))
    todo = [(0, root_box, -1)]                     # (BVH2 node, its box, next) — a stack that emits in preorder
    while todo:
        i, (lo, hi), nxt = todo.pop()
        at = len(out)
        if i >= leaf_base:
            out.append((lo, float((payload_of_leaf(i - leaf_base) << 4) | 1), hi, float(nxt)))
            continue
        out.append((lo, -1.0, hi, float(nxt)))
        nd = nodes2[i]
        l, r = int(nd["addr_left"]), int(nd["addr_right"])
        right_at = at + 1 + int(size[l])
        todo.append((r, (nd["rmin_or_v2"], nd["rmax"]), nxt))
        todo.append((l, (nd["lmin_or_v0"], nd["lmax_or_v1"]), right_at))       # popped first: lands at at + 1
    return base


def two_level_buffers(tlas, instances, blas_list):
    """The reference's buffers for a two-level scene made by oracle.two_level_scene: (nodes bbox[], vertices float4[],
    faces Face[], shapes Shape[], root_idx)."""
    out, verts, faces, roots = [], [], [], []
    for b in blas_list:
        n = (len(b) + 1) // 2
        leaves = b[n - 1:] if n > 1 else b
        first_face = len(faces)
        for k, lf in enumerate(leaves):
            v0 = len(verts)
            verts += [np.append(lf["lmin_or_v0"], 0), np.append(lf["lmax_or_v1"], 0), np.append(lf["rmin_or_v2"], 0)]
            faces.append(((v0, v0 + 1, v0 + 2), 0, int(lf["prim_id"])))
        roots.append(_skiplinks(b, out, lambda p, f=first_face: f + p))
    shapes = np.zeros(len(instances), dtype=SHAPE)
    for k, inst in enumerate(instances):
        shapes[k]["id"] = int(inst["shape_id"])
        shapes[k]["bvh_idx"] = roots[int(inst["blas"])]
        shapes[k]["m"][:3] = inst["world_to_object"].reshape(3, 4)
        shapes[k]["m"][3] = (0, 0, 0, 1)
    ntl = (len(tlas) + 1) // 2
    tl_leaves = tlas[ntl - 1:] if ntl > 1 else tlas
    root_idx = _skiplinks(tlas, out, lambda p: int(tl_leaves[p]["prim_id"]))
    nodes = np.zeros(len(out), dtype=BBOX)
    for k, (lo, w0, hi, w1) in enumerate(out):
        nodes[k]["pmin"][:3], nodes[k]["pmin"][3], nodes[k]["pmax"][:3], nodes[k]["pmax"][3] = lo, w0, hi, w1
    f = np.zeros(len(faces), dtype=FACE)
    for k, (idx, sid, pid) in enumerate(faces):
        f[k]["idx"], f[k]["shape_id"], f[k]["prim_id"] = idx, sid, pid
    return nodes, np.array(verts, dtype=np.float32), f, shapes, root_idx


def intersect_two_level(module: Module, nodes, vertices, faces, shapes, root_idx, rays: np.ndarray, fill_hits: np.ndarray):
    """intersect_main of intersect_bvh2level_skiplinks.cl (:134-307)."""
    n = len(rays)
    bufs = [DeviceBuffer(nodes), DeviceBuffer(vertices), DeviceBuffer(faces), DeviceBuffer(shapes), DeviceBuffer(rays),
            DeviceBuffer(np.array([n], dtype=np.int32)), DeviceBuffer(fill_hits)]
    try:
        module.launch("intersect_main", n, [bufs[0], bufs[1], bufs[2], bufs[3], C.c_int32(int(root_idx)), bufs[4], bufs[5], bufs[6]])
        return bufs[6].read(fill_hits.dtype, len(fill_hits))
    finally:
        for b in bufs:
            b.free()
