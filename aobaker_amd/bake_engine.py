"""Host-side mirror of the reference's BakeEngine namespace over the C ABI.

`BakeEngine.Init() / Dispose() / Bake(scene)` keep the names, argument meaning and
(absent) return values of the reference (BakeEngine.h:5-10); failures raise
`capi.AobError` where the reference silently ignored HRESULTs.  `Baker` wraps the
staged API (`aob_bake_*`) for tests, the benchmark and the multi-GPU driver.

Everything here calls the HIP library; nothing falls back to a CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .scene import BakeTexture, FlatScene, Scene


class Baker:
    """Staged bake of one flattened scene resident in HBM (aob_bake_* entry points)."""

    def __init__(self, flat: FlatScene, options: capi.AobOptions):
        self.lib = capi.load()
        self.flat = flat
        self.options = options
        h = C.c_void_p()
        capi.check(self.lib.aob_bake_create(C.byref(flat.desc), C.byref(options), C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            self.lib.aob_bake_destroy(self.handle)
            self.handle = None

    def __enter__(self): return self
    def __exit__(self, *a): self.close()
    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # stages -------------------------------------------------------------------
    def build_bvh(self): capi.check(self.lib.aob_bake_build_bvh(self.handle))
    def rasterize(self, material=0, rows=None):
        """rows = (begin, end): only that strip of texel rows (a rank's share of a multi-GPU bake)."""
        if rows is None:
            capi.check(self.lib.aob_bake_rasterize(self.handle, material))
        else:
            capi.check(self.lib.aob_bake_rasterize_rows(self.handle, material, int(rows[0]), int(rows[1])))
    def trace(self, material=0, begin=0, end=capi.UINT64_MAX):
        capi.check(self.lib.aob_bake_trace(self.handle, material, begin, end))
    def resolve(self, material=0): capi.check(self.lib.aob_bake_resolve(self.handle, material))
    def clear_results(self, material=0): capi.check(self.lib.aob_bake_clear_results(self.handle, material))
    def run_device(self): capi.check(self.lib.aob_bake_run_device(self.handle))
    def sync(self): capi.check(self.lib.aob_bake_sync(self.handle))

    def download(self, material=0, dst: np.ndarray | None = None):
        if dst is None:
            capi.check(self.lib.aob_bake_download(self.handle, material, None, 0))
            return None
        capi.check(self.lib.aob_bake_download(self.handle, material, dst.ctypes.data, dst.nbytes))
        return dst

    # introspection --------------------------------------------------------------
    def stats(self) -> capi.AobStats:
        s = capi.AobStats()
        capi.check(self.lib.aob_bake_get_stats(self.handle, C.byref(s)))
        return s

    def last_stage_ms(self) -> dict:
        """HIP-event times of the most recent stages (ms; None: not run): bvh_build, rasterize, trace, and of the trace's last launch tile_cuts, horizon, walk."""
        out = (C.c_float * 6)()
        capi.check(self.lib.aob_bake_last_stage_ms(self.handle, out))
        return {k: (None if out[i] < 0 else float(out[i])) for i, k in enumerate(("bvh_build", "rasterize", "trace", "tile_cuts", "horizon", "walk"))}

    def node_count(self) -> int:
        n = C.c_uint64()
        capi.check(self.lib.aob_bake_node_count(self.handle, C.byref(n)))
        return n.value

    def nodes(self) -> np.ndarray:
        n = self.node_count()
        out = np.zeros(max(n, 1), dtype=capi.node_dtype())
        capi.check(self.lib.aob_bake_copy_nodes(self.handle, out.ctypes.data, n))
        return out[:n]

    def nodes4(self) -> np.ndarray:
        n = C.c_uint64()
        capi.check(self.lib.aob_bake_node4_count(self.handle, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=capi.node4_dtype())
        capi.check(self.lib.aob_bake_copy_nodes4(self.handle, out.ctypes.data, n.value))
        return out[:n.value]

    def leaf_tris(self) -> np.ndarray:
        n = self.stats().triangle_count
        out = np.zeros(max(n, 1), dtype=capi.leaf_tri_dtype())
        capi.check(self.lib.aob_bake_copy_leaf_tris(self.handle, out.ctypes.data, n))
        return out[:n]

    def nodes2q(self):
        """(AobBvh2QNode words [n-1, 8] uint32, frame [6] float32)."""
        n = max(0, int(self.stats().triangle_count) - 1)
        out = np.zeros((max(n, 1), 8), dtype=np.uint32)
        capi.check(self.lib.aob_bake_copy_nodes2q(self.handle, out.ctypes.data, n))
        frame = np.zeros(6, dtype=np.float32)
        capi.check(self.lib.aob_bake_quant_frame(self.handle, frame.ctypes.data))
        return out[:n], frame

    def nodes4q(self) -> np.ndarray:
        n = C.c_uint64()
        capi.check(self.lib.aob_bake_node4q_count(self.handle, C.byref(n)))
        out = np.zeros((max(n.value, 1), 16), dtype=np.uint32)
        capi.check(self.lib.aob_bake_copy_nodes4q(self.handle, out.ctypes.data, n.value))
        return out[:n.value]

    def tile_cuts(self, material=0) -> np.ndarray:
        """The tile cuts of the material's most recent trace launch: uint32 [tiles, records_per_tile, 128] (ref[32], then 32 x {x, y, z}
        plane words), or an empty array if the launch walked from the root."""
        tiles, records = C.c_uint32(), C.c_uint32()
        rc = self.lib.aob_bake_copy_tile_cuts(self.handle, material, None, 0, C.byref(tiles), C.byref(records))
        if tiles.value == 0:
            capi.check(rc)
            return np.zeros((0, 0, 128), dtype=np.uint32)
        out = np.zeros((tiles.value, records.value, 128), dtype=np.uint32)
        capi.check(self.lib.aob_bake_copy_tile_cuts(self.handle, material, out.ctypes.data, out.size, C.byref(tiles), C.byref(records)))
        return out

    def horizon(self, material=0) -> np.ndarray:
        """The horizon tables of the material's most recent trace launches: uint8 [surfels, stride] in the order of surfels() — byte cy =
        leading cells of row cy of the COSINE sampler's grid that were not cast — or an empty array if the launches had none."""
        stride, n = C.c_uint32(), C.c_uint64()
        n_all = self.surfel_count(material)
        out = np.zeros((max(n_all, 1), 32), dtype=np.uint8)
        capi.check(self.lib.aob_bake_copy_horizon(self.handle, material, out.ctypes.data, out.nbytes, C.byref(stride), C.byref(n)))
        if n.value == 0:
            return np.zeros((0, 0), dtype=np.uint8)
        return out.reshape(-1)[:n.value * stride.value].reshape(n.value, stride.value).copy()

    def surfel_count(self, material=0) -> int:
        n = C.c_uint64()
        capi.check(self.lib.aob_bake_surfel_count(self.handle, material, C.byref(n)))
        return n.value

    def surfels(self, material=0) -> np.ndarray:
        n = self.surfel_count(material)
        out = np.zeros(max(n, 1), dtype=capi.surfel_dtype())
        capi.check(self.lib.aob_bake_copy_surfels(self.handle, material, out.ctypes.data, n))
        return out[:n]

    def results(self, material=0) -> np.ndarray:
        m = self.flat.material_array[material]
        cells = m.width * m.height
        out = np.zeros(cells, dtype=capi.result_dtype())
        capi.check(self.lib.aob_bake_copy_results(self.handle, material, out.ctypes.data, cells))
        return out

    def results_device_ptr(self, material=0):
        p, n = C.c_void_p(), C.c_uint64()
        capi.check(self.lib.aob_bake_results_device_ptr(self.handle, material, C.byref(p), C.byref(n)))
        return p.value, n.value

    def image_device_ptr(self, material=0):
        p, n = C.c_void_p(), C.c_uint64()
        capi.check(self.lib.aob_bake_image_device_ptr(self.handle, material, C.byref(p), C.byref(n)))
        return p.value, n.value

    def shard_bounds(self, material, parts):
        """(bounds[parts+1], [(first_cell, last_cell)] per part, overlapping): tile-row aligned balanced
        split of the trace-order surfel list; first > last marks an empty part."""
        bounds = (C.c_uint64 * (parts + 1))()
        c0, c1 = (C.c_uint64 * parts)(), (C.c_uint64 * parts)()
        ov = C.c_int32()
        capi.check(self.lib.aob_bake_shard_bounds(self.handle, material, parts, bounds, c0, c1, C.byref(ov)))
        return list(bounds), list(zip(c0, c1)), bool(ov.value)

    def stream(self):
        p = C.c_void_p()
        capi.check(self.lib.aob_bake_stream(self.handle, C.byref(p)))
        return p.value

    def last_trace_ms(self) -> float:
        ms = C.c_float()
        capi.check(self.lib.aob_bake_last_trace_ms(self.handle, C.byref(ms)))
        return ms.value


class BakeEngine:
    """namespace BakeEngine { void Init(); void Dispose(); void Bake(Scene*); } (BakeEngine.h:5-10)"""

    Options: capi.AobOptions | None = None     # the reference's constants unless overridden
    LastStats: capi.AobStats | None = None

    @staticmethod
    def Init(device: int = -1):
        cfg = capi.AobDeviceConfig()
        cfg.device = device
        capi.check(capi.load().aob_init(C.byref(cfg)))
        if BakeEngine.Options is None:
            BakeEngine.Options = capi.default_options()

    @staticmethod
    def Dispose():
        capi.load().aob_shutdown()

    @staticmethod
    def Bake(scene: Scene, texture: BakeTexture | None = None):
        """Bake(Scene*) writes every material's AO texture (BakeEngine.cpp:404-734).
        Bake(Scene*, BakeTexture*) — the overload BASELINE.json's north star names — bakes only the
        material whose GetAOTexture() is `texture`."""
        opts = BakeEngine.Options or capi.default_options()
        flat = FlatScene(scene)
        if texture is None:
            stats = capi.AobStats()
            capi.check(capi.load().aob_bake(C.byref(flat.desc), C.byref(opts), C.byref(stats)))
            BakeEngine.LastStats = stats
            for i in range(scene.GetMaterialCount()):
                scene.GetMaterial(i).GetAOTexture().RefreshPreview()     # BakeEngine.cpp:734
            return
        which = [i for i in range(scene.GetMaterialCount()) if scene.GetMaterial(i).GetAOTexture() is texture]
        if not which:
            raise ValueError("texture does not belong to any material of the scene")
        with Baker(flat, opts) as b:
            b.build_bvh()
            for i in which:
                b.rasterize(i)
                b.trace(i)
                b.resolve(i)
                b.download(i)
            BakeEngine.LastStats = b.stats()
        texture.RefreshPreview()
