"""ctypes wrapper of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (aobaker_amd/) never imports this.
Traversal, intersection and the LBVH build are pinned against the reference's own OpenCL kernels run on the GPU
(oracle/ref_kernels.mk, tests/test_gpu_reference_kernels.py); see aob_oracle.h for what is not.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from aobaker_amd import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcTraceStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("inner_visits", C.c_uint64),
                ("leaf_visits", C.c_uint64), ("hits", C.c_uint64)]


TRIANGLE = np.dtype([("v", "<f4", (3, 3)), ("shape_id", "<u4"), ("prim_id", "<u4")])


def build():
    subprocess.run(["make", "-s", "-C", _HERE, "liboracle.so"], check=True)


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    # AOB_ORACLE_LIBRARY: another build of the same source (the sanitizer build, tests/test_oracle_sanitizers.py)
    path = os.environ.get("AOB_ORACLE_LIBRARY") or os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    P, vp, u64, i32, u32, f32 = C.POINTER, C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32, C.c_float
    fp = P(f32)
    sig = {
        "orc_msvc_rand": (u32, [P(u32)]),
        "orc_ftoi": (i32, [f32]),
        "orc_f32_to_f16": (C.c_uint16, [f32, C.c_int]),
        "orc_f16_to_f32": (f32, [C.c_uint16]),
        "orc_intersect_triangle": (f32, [fp, fp, fp, fp, fp, f32]),
        "orc_safe_invdir": (None, [fp, fp]),
        "orc_intersect_bbox": (None, [fp, fp, fp, fp, f32, fp]),
        "orc_cosine_sample": (None, [u32, u32, i32, i32, fp]),
        "orc_sincos2pi": (None, [f32, fp, fp]),
        "orc_hemisphere_table": (None, [vp, u64]),
        "orc_compat_table_entries": (u64, [P(capi.AobOptions)]),
        "orc_scene_triangle_count": (u64, [P(capi.AobSceneDesc)]),
        "orc_scene_triangles": (None, [P(capi.AobSceneDesc), vp]),
        "orc_bvh_build": (u64, [vp, u64, vp, P(u32)]),
        "orc_bvh_build_ex": (u64, [vp, u64, vp, P(u32), C.c_int, C.c_int]),
        "orc_bvh4_build": (u64, [vp, u64, vp, vp]),
        "orc_bvh4_build_ex": (u64, [vp, u64, vp, vp, C.c_int]),
        "orc_bvh2q_build": (u64, [vp, u64, vp, vp]),
        "orc_bvh4q_build": (u64, [vp, u64, vp, vp]),
        "orc_rasterize": (u64, [P(capi.AobSceneDesc), i32, P(capi.AobOptions), vp, vp, u64]),
        "orc_trace": (C.c_int, [vp, u64, vp, vp, u64, u64, P(capi.AobOptions), vp, i32, i32, vp, i32,
                                P(OrcTraceStats)]),
        "orc_rasterize_ex": (u64, [P(capi.AobSceneDesc), i32, P(capi.AobOptions), vp, vp, vp, u64]),
        "orc_trace_ex": (C.c_int, [vp, u64, vp, vp, vp, vp, u32, f32, u64, u64, P(capi.AobOptions), vp, i32, i32, vp, i32,
                                   P(OrcTraceStats)]),
        "orc_resolve": (None, [vp, i32, i32, P(capi.AobOptions), i32, i32, vp]),
        "orc_bake": (C.c_int, [P(capi.AobSceneDesc), P(capi.AobOptions), i32, P(OrcTraceStats)]),
        "orc_generate_rays": (u64, [vp, vp, u64, u64, P(capi.AobOptions), vp, i32, f32, i32, vp]),
        "orc_intersect_rays": (None, [vp, u64, vp, u64, i32, vp]),
        "orc_occluded_rays": (None, [vp, u64, vp, u64, vp]),
        "orc_reference_stack_depths": (None, [vp, u64, vp, u64, i32, vp]),
        "orc_process_intersections": (None, [vp, u64, u64, vp, vp, P(capi.AobOptions), f32, i32, i32, vp]),
        "orc_morton_codes": (None, [vp, u64, C.c_int, vp, vp, vp]),
        "orc_two_level_intersect": (C.c_int, [vp, u64, vp, vp, vp, vp, u64, i32, vp, P(OrcTraceStats), P(OrcTraceStats)]),
        "orc_flat_intersect": (C.c_int, [vp, u64, vp, u64, i32, vp, P(OrcTraceStats)]),
        "orc_walk4q": (C.c_int, [vp, u64, vp, vp, vp, u64, i32, P(OrcTraceStats), P(OrcTraceStats)]),
        "orc_kary_visits": (C.c_int, [vp, u64, vp, u64, i32, i32, i32, P(OrcTraceStats), P(u64)]),
        "orc_bvh_build_sah": (u64, [vp, u64, vp, P(u32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _LIB = lib
    return lib


def _f3(a):
    return (C.c_float * 3)(*[float(x) for x in a])


def msvc_rand_sequence(n, state=1):
    lib = load()
    s = C.c_uint32(state)
    return [lib.orc_msvc_rand(C.byref(s)) for _ in range(n)]


def ftoi(v):
    return load().orc_ftoi(C.c_float(v))


def f32_to_f16(v, mode):
    return load().orc_f32_to_f16(C.c_float(v), mode)


def f16_to_f32(h):
    return load().orc_f16_to_f32(C.c_uint16(h))


def intersect_triangle(o, d, v1, v2, v3, t_max):
    return load().orc_intersect_triangle(_f3(o), _f3(d), _f3(v1), _f3(v2), _f3(v3), C.c_float(t_max))


def safe_invdir(d):
    out = (C.c_float * 3)()
    load().orc_safe_invdir(_f3(d), out)
    return np.array(out[:], dtype=np.float32)


def intersect_bbox(pmin, pmax, invdir, oxinvdir, t_max):
    out = (C.c_float * 2)()
    load().orc_intersect_bbox(_f3(pmin), _f3(pmax), _f3(invdir), _f3(oxinvdir), C.c_float(t_max), out)
    return float(out[0]), float(out[1])


def cosine_sample(seed, texel, i, spp):
    out = (C.c_float * 3)()
    load().orc_cosine_sample(seed, texel, i, spp, out)
    return np.array(out[:], dtype=np.float32)


def sincos2pi(u):
    s, c = C.c_float(), C.c_float()
    load().orc_sincos2pi(C.c_float(u), C.byref(s), C.byref(c))
    return s.value, c.value


def hemisphere_table(count):
    out = np.empty((count, 3), dtype=np.float32)
    load().orc_hemisphere_table(out.ctypes.data, count)
    return out


def compat_table(opt):
    return hemisphere_table(int(load().orc_compat_table_entries(C.byref(opt))))


def scene_triangles(desc):
    lib = load()
    n = int(lib.orc_scene_triangle_count(C.byref(desc)))
    out = np.zeros(max(n, 1), dtype=TRIANGLE)
    lib.orc_scene_triangles(C.byref(desc), out.ctypes.data)
    return out[:n]


def bvh_build(tris, builder=None, adaptive=None):
    """builder: 0 = LBVH (build_hlbvh.cl), 1 = PLOC; adaptive: extent-adaptive Morton bit order
    (AobOptions.morton_adaptive); None = the default options' values."""
    lib = load()
    if builder is None:
        builder = default_options().bvh_builder
    if builder < 0:
        builder = 1            # "by workload" resolves to PLOC below 2^28 rays (resolve_builder); a caller that means the SAH top says 2
    if adaptive is None:
        adaptive = default_options().morton_adaptive
    n = len(tris)
    tris = np.ascontiguousarray(tris)
    nodes = np.zeros(max(1, 2 * n), dtype=capi.node_dtype())
    depth = C.c_uint32()
    cnt = int(lib.orc_bvh_build_ex(tris.ctypes.data, n, nodes.ctypes.data, C.byref(depth), builder, adaptive))
    return nodes[:cnt], depth.value


def bvh4_build(nodes2, pairs=None):
    """pairs: leaf pairs take one child slot (AOB_PAIR_FLAG); None = what the default options build (quantised 4-ary: yes)."""
    lib = load()
    if pairs is None:
        d = default_options()
        pairs = d.bvh_width == 4 and d.bvh_quantised != 0
    nodes2 = np.ascontiguousarray(nodes2)
    n = (len(nodes2) + 1) // 2
    nodes4 = np.zeros(max(1, n - 1), dtype=capi.node4_dtype())
    tris = np.zeros(max(1, n), dtype=capi.leaf_tri_dtype())
    cnt = int(lib.orc_bvh4_build_ex(nodes2.ctypes.data if len(nodes2) else None, len(nodes2), nodes4.ctypes.data, tris.ctypes.data, int(bool(pairs))))
    return nodes4[:cnt], tris[:n]


def bvh2q_build(nodes2):
    """(words [n-1, 8] uint32, frame [6] float32)"""
    lib = load()
    nodes2 = np.ascontiguousarray(nodes2)
    n = (len(nodes2) + 1) // 2
    out = np.zeros((max(1, n - 1), 8), dtype=np.uint32)
    frame = np.zeros(6, dtype=np.float32)
    cnt = int(lib.orc_bvh2q_build(nodes2.ctypes.data if len(nodes2) else None, len(nodes2), out.ctypes.data, frame.ctypes.data))
    return out[:cnt], frame


def bvh4q_build(nodes4, frame):
    lib = load()
    nodes4 = np.ascontiguousarray(nodes4)
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    out = np.zeros((max(1, len(nodes4)), 16), dtype=np.uint32)
    cnt = int(lib.orc_bvh4q_build(nodes4.ctypes.data, len(nodes4), frame.ctypes.data, out.ctypes.data))
    return out[:cnt]


def rasterize(desc, material, opt):
    lib = load()
    n = int(lib.orc_rasterize(C.byref(desc), material, C.byref(opt), None, None, 0))
    surfels = np.zeros(max(n, 1), dtype=capi.surfel_dtype())
    sidx = np.zeros(max(n, 1), dtype=np.uint32)
    lib.orc_rasterize(C.byref(desc), material, C.byref(opt), surfels.ctypes.data, sidx.ctypes.data, n)
    return surfels[:n], sidx[:n]


def rasterize_ex(desc, material, opt):
    """(surfels, section_index, surfel_mesh): also the mesh (section) every surfel came from."""
    lib = load()
    n = int(lib.orc_rasterize(C.byref(desc), material, C.byref(opt), None, None, 0))
    surfels = np.zeros(max(n, 1), dtype=capi.surfel_dtype())
    sidx = np.zeros(max(n, 1), dtype=np.uint32)
    smesh = np.zeros(max(n, 1), dtype=np.uint32)
    lib.orc_rasterize_ex(C.byref(desc), material, C.byref(opt), surfels.ctypes.data, sidx.ctypes.data, smesh.ctypes.data, n)
    return surfels[:n], sidx[:n], smesh[:n]


def trace(nodes, surfels, section_index, opt, width, height, hemisphere=None, begin=0, end=None,
          threads=1, results=None, surfel_mesh=None, mesh_flags=None, ao_max_range=0.0):
    """surfel_mesh + mesh_flags: section visibility (AOB_MESH_*); ao_max_range: the material's range
    (honoured when opt.material_ranges is set)."""
    lib = load()
    if mesh_flags is not None and surfel_mesh is not None:
        return _trace_ex(nodes, surfels, section_index, opt, width, height, hemisphere, begin, end, threads, results,
                         surfel_mesh, mesh_flags, ao_max_range)
    if ao_max_range:
        return _trace_ex(nodes, surfels, section_index, opt, width, height, hemisphere, begin, end, threads, results,
                         None, None, ao_max_range)
    if end is None:
        end = len(surfels)
    if results is None:
        results = np.zeros(width * height, dtype=capi.result_dtype())
    stats = OrcTraceStats()
    nodes = np.ascontiguousarray(nodes)
    surfels = np.ascontiguousarray(surfels)
    sidx = None if section_index is None else np.ascontiguousarray(section_index, dtype=np.uint32)
    hemi = None if hemisphere is None else np.ascontiguousarray(hemisphere, dtype=np.float32)
    rc = lib.orc_trace(nodes.ctypes.data if len(nodes) else None, len(nodes),
                       surfels.ctypes.data if len(surfels) else None,
                       None if sidx is None or not len(sidx) else sidx.ctypes.data,
                       begin, end, C.byref(opt), None if hemi is None else hemi.ctypes.data,
                       width, height, results.ctypes.data, threads, C.byref(stats))
    if rc != 0:
        raise RuntimeError(f"orc_trace failed: {rc}")
    return results, stats


def _trace_ex(nodes, surfels, section_index, opt, width, height, hemisphere, begin, end, threads, results,
              surfel_mesh, mesh_flags, ao_max_range):
    lib = load()
    if end is None:
        end = len(surfels)
    if results is None:
        results = np.zeros(width * height, dtype=capi.result_dtype())
    stats = OrcTraceStats()
    nodes = np.ascontiguousarray(nodes)
    surfels = np.ascontiguousarray(surfels)
    sidx = None if section_index is None else np.ascontiguousarray(section_index, dtype=np.uint32)
    hemi = None if hemisphere is None else np.ascontiguousarray(hemisphere, dtype=np.float32)
    smesh = None if surfel_mesh is None else np.ascontiguousarray(surfel_mesh, dtype=np.uint32)
    flags = None if mesh_flags is None else np.ascontiguousarray(mesh_flags, dtype=np.uint32)
    rc = lib.orc_trace_ex(nodes.ctypes.data if len(nodes) else None, len(nodes),
                          surfels.ctypes.data if len(surfels) else None,
                          None if sidx is None or not len(sidx) else sidx.ctypes.data,
                          None if smesh is None or not len(smesh) else smesh.ctypes.data,
                          None if flags is None else flags.ctypes.data, 0 if flags is None else len(flags), float(ao_max_range),
                          begin, end, C.byref(opt), None if hemi is None else hemi.ctypes.data,
                          width, height, results.ctypes.data, threads, C.byref(stats))
    if rc != 0:
        raise RuntimeError(f"orc_trace_ex failed: {rc}")
    return results, stats


def resolve(results, width, height, opt, channels=1, bpc=1):
    out = np.zeros(width * height * channels * bpc, dtype=np.uint8)
    results = np.ascontiguousarray(results)
    load().orc_resolve(results.ctypes.data, width, height, C.byref(opt), channels, bpc, out.ctypes.data)
    return out


def bake(desc, opt, threads=1):
    stats = OrcTraceStats()
    rc = load().orc_bake(C.byref(desc), C.byref(opt), threads, C.byref(stats))
    if rc != 0:
        raise RuntimeError(f"orc_bake failed: {rc}")
    return stats


# ---- the reference's own records (tests/refkernels.py runs the reference's OpenCL kernels on them) ----------------
RAY = np.dtype([("o", "<f4", 4), ("d", "<f4", 4), ("extra", "<i4", 2), ("backface_cull", "<i4"), ("padding", "<i4")])   # 48 B
INTERSECTION = np.dtype([("shape_id", "<i4"), ("prim_id", "<i4"), ("pad", "<i4", 2), ("uvwt", "<f4", 4)])               # 32 B


def generate_rays(surfels, section_index, opt, width, hemisphere=None, t_max=None, mask=0, begin=0, end=None):
    """GenerateRaysComputeShader.hlsl:43-54: sample_count 48-byte ray records per surfel, the rays orc_trace casts."""
    if end is None:
        end = len(surfels)
    surfels = np.ascontiguousarray(surfels)
    sidx = None if section_index is None else np.ascontiguousarray(section_index, dtype=np.uint32)
    hemi = None if hemisphere is None else np.ascontiguousarray(hemisphere, dtype=np.float32)
    out = np.zeros(max(1, (end - begin) * opt.sample_count), dtype=RAY)
    n = load().orc_generate_rays(surfels.ctypes.data if len(surfels) else None,
                                 None if sidx is None or not len(sidx) else sidx.ctypes.data, begin, end, C.byref(opt),
                                 None if hemi is None else hemi.ctypes.data, width,
                                 float(opt.ray_tmax if t_max is None else t_max), int(mask), out.ctypes.data)
    return out[:n]


def intersect_rays(nodes, rays, use_mask=False, fill=None):
    """intersect_main (intersect_bvh2_lds.cl:65-226) on the CPU.  fill: initial content of the records (inactive rays
    leave theirs untouched)."""
    nodes, rays = np.ascontiguousarray(nodes), np.ascontiguousarray(rays)
    hits = np.zeros(max(1, len(rays)), dtype=INTERSECTION) if fill is None else fill.copy()
    load().orc_intersect_rays(nodes.ctypes.data if len(nodes) else None, len(nodes), rays.ctypes.data, len(rays),
                              int(bool(use_mask)), hits.ctypes.data)
    return hits[:len(rays)]


def occluded_rays(nodes, rays, fill=None):
    """occluded_main (intersect_bvh2_lds.cl:228-363) on the CPU: 1 / -1 per active ray."""
    nodes, rays = np.ascontiguousarray(nodes), np.ascontiguousarray(rays)
    hits = np.zeros(max(1, len(rays)), dtype=np.int32) if fill is None else fill.copy()
    load().orc_occluded_rays(nodes.ctypes.data if len(nodes) else None, len(nodes), rays.ctypes.data, len(rays), hits.ctypes.data)
    return hits[:len(rays)]


def reference_stack_depths(nodes, rays, any_hit=False):
    """Deferred entries each ray holds at most in the reference's walk (>= 16 spills its LDS stack, intersect_bvh2_lds.cl:144-153)."""
    nodes, rays = np.ascontiguousarray(nodes), np.ascontiguousarray(rays)
    out = np.zeros(max(1, len(rays)), dtype=np.uint32)
    load().orc_reference_stack_depths(nodes.ctypes.data if len(nodes) else None, len(nodes), rays.ctypes.data, len(rays), int(bool(any_hit)), out.ctypes.data)
    return out[:len(rays)]


def process_intersections(surfels, rays, hits, opt, width, height, falloff=None, results=None):
    """ProcessIntersectionsComputeShader.hlsl:52-92 from Intersection records of all surfels' rays."""
    surfels, rays, hits = np.ascontiguousarray(surfels), np.ascontiguousarray(rays), np.ascontiguousarray(hits)
    if results is None:
        results = np.zeros(width * height, dtype=capi.result_dtype())
    load().orc_process_intersections(surfels.ctypes.data, 0, len(surfels), rays.ctypes.data, hits.ctypes.data, C.byref(opt),
                                     float(opt.falloff_distance if falloff is None else falloff), width, height,
                                     results.ctypes.data)
    return results


def morton_codes(tris, adaptive=0):
    """(codes uint64 [n] in primitive order, primitive boxes [n, 8], scene box [8]) — boxes as the reference's bbox records."""
    tris = np.ascontiguousarray(tris)
    n = len(tris)
    codes = np.zeros(max(1, n), dtype=np.uint64)
    boxes = np.zeros((max(1, n), 8), dtype=np.float32)
    scene = np.zeros(8, dtype=np.float32)
    load().orc_morton_codes(tris.ctypes.data, n, int(adaptive), codes.ctypes.data, boxes.ctypes.data, scene.ctypes.data)
    return codes[:n], boxes[:n], scene


INSTANCE = np.dtype([("world_to_object", "<f4", 12), ("shape_id", "<u4"), ("blas", "<u4")])


def flat_intersect(nodes, rays, any_hit=False, want_hits=True):
    """traverse() over ray records with node-visit counts: (hits or None, OrcTraceStats)."""
    nodes, rays = np.ascontiguousarray(nodes), np.ascontiguousarray(rays)
    hits = np.zeros(max(1, len(rays)), dtype=INTERSECTION) if want_hits else None
    st = OrcTraceStats()
    rc = load().orc_flat_intersect(nodes.ctypes.data if len(nodes) else None, len(nodes), rays.ctypes.data, len(rays), int(bool(any_hit)),
                                   None if hits is None else hits.ctypes.data, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"orc_flat_intersect failed: {rc}")
    return (None if hits is None else hits[:len(rays)]), st


def two_level_intersect(tlas, instances, blas_list, rays, any_hit=False, want_hits=True):
    """The two-level walk (intersect_bvh2level_skiplinks.cl's structure, traverse()'s discipline):
    (hits or None, top-level stats, bottom-level stats)."""
    tlas, rays = np.ascontiguousarray(tlas), np.ascontiguousarray(rays)
    instances = np.ascontiguousarray(instances, dtype=INSTANCE)
    blas_list = [np.ascontiguousarray(b) for b in blas_list]
    ptrs = (C.c_void_p * len(blas_list))(*[b.ctypes.data for b in blas_list])
    counts = (C.c_uint64 * len(blas_list))(*[len(b) for b in blas_list])
    hits = np.zeros(max(1, len(rays)), dtype=INTERSECTION) if want_hits else None
    top, bottom = OrcTraceStats(), OrcTraceStats()
    rc = load().orc_two_level_intersect(tlas.ctypes.data if len(tlas) else None, len(tlas), instances.ctypes.data, ptrs, counts,
                                        rays.ctypes.data, len(rays), int(bool(any_hit)), None if hits is None else hits.ctypes.data,
                                        C.byref(top), C.byref(bottom))
    if rc != 0:
        raise RuntimeError(f"orc_two_level_intersect failed: {rc}")
    return (None if hits is None else hits[:len(rays)]), top, bottom


def two_level_scene(scene, builder=None, adaptive=None):
    """Top-level tree, instances and bottom-level trees of a Scene (aobaker_amd.scene): one instance per (object, section) in
    scene order — shape id = mesh index + 1, as the flat build numbers them — sections whose meshes hold the same vertex
    positions and indices share one bottom-level tree (built in OBJECT space), the top level is built over the instances'
    world boxes with the same builder.  Returns (tlas, instances, blas_list)."""
    blas_list, blas_key, instances, boxes = [], {}, [], []
    shape = 0
    for i in range(scene.GetObjectCount()):
        obj = scene.GetObject(i)
        model = np.asarray(obj.BuildTransform(), dtype=np.float64).reshape(4, 4)      # row-vector: world = [p 1] model
        inv = np.linalg.inv(model)
        for s in range(obj.GetSectionCount()):
            mesh = obj.GetSection(s).GetPreviewMesh()
            shape += 1
            v = np.ascontiguousarray(mesh.GetVertices()["Position"], dtype=np.float32)
            idx = np.ascontiguousarray(mesh.GetIndices(), dtype=np.uint32)
            key = (v.tobytes(), idx.tobytes())
            if key not in blas_key:
                tri = np.zeros(len(idx) // 3, dtype=TRIANGLE)
                tri["v"] = v[idx[:3 * (len(idx) // 3)].reshape(-1, 3)]
                tri["shape_id"] = 0
                tri["prim_id"] = np.arange(len(tri), dtype=np.uint32)
                blas_key[key] = len(blas_list)
                blas_list.append(bvh_build(tri, builder, adaptive)[0])
            world = (np.concatenate([v.astype(np.float64), np.ones((len(v), 1))], axis=1) @ model)[:, :3].astype(np.float32)
            boxes.append((world.min(axis=0), world.max(axis=0)))
            inst = np.zeros(1, dtype=INSTANCE)
            inst["world_to_object"][0] = inv.T[:3, :].reshape(-1).astype(np.float32)     # column-vector rows m0, m1, m2
            inst["shape_id"], inst["blas"] = shape, blas_key[key]
            instances.append(inst)
    fake = np.zeros(len(boxes), dtype=TRIANGLE)
    for k, (lo, hi) in enumerate(boxes):
        fake["v"][k, 0], fake["v"][k, 1], fake["v"][k, 2] = lo, hi, lo
    fake["prim_id"] = np.arange(len(boxes), dtype=np.uint32)
    tlas = bvh_build(fake, builder, adaptive)[0]
    return tlas, np.concatenate(instances), blas_list


def walk4q(nodes4q, frame, leaf_tris, rays, any_hit=True):
    """(stats of all rays, stats of the rays that hit) of traverse4q on the given arrays."""
    nodes4q = np.ascontiguousarray(nodes4q, dtype=np.uint32)
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    leaf_tris, rays = np.ascontiguousarray(leaf_tris), np.ascontiguousarray(rays)
    a, h = OrcTraceStats(), OrcTraceStats()
    rc = load().orc_walk4q(nodes4q.ctypes.data, len(nodes4q), frame.ctypes.data, leaf_tris.ctypes.data, rays.ctypes.data, len(rays),
                           int(bool(any_hit)), C.byref(a), C.byref(h))
    if rc != 0:
        raise RuntimeError(f"orc_walk4q failed: {rc}")
    return a, h


class OrcEntryStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "hits", "inner_visits", "leaf_visits", "entries_tested", "mask_bits")]


def tile_cut(nodes4q, frame, lo, hi, max_entries=32, rule=1):
    """A tile's cut of the quantised 4-ary tree (trace.hip k_tile_cuts): rows of {ref, x, y, z plane words}."""
    nodes4q = np.ascontiguousarray(nodes4q, dtype=np.uint32)
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    lo, hi = np.ascontiguousarray(lo, dtype=np.float32), np.ascontiguousarray(hi, dtype=np.float32)
    out = np.zeros((128, 4), dtype=np.uint32)
    lib = load()
    lib.orc_tile_cut.restype = C.c_int32
    m = lib.orc_tile_cut(C.c_void_p(nodes4q.ctypes.data), C.c_uint64(len(nodes4q)), C.c_void_p(frame.ctypes.data), C.c_void_p(lo.ctypes.data),
                         C.c_void_p(hi.ctypes.data), C.c_int32(max_entries), C.c_int32(rule), C.c_void_p(out.ctypes.data))
    return out[:m].copy()


def tile_cut_reach(nodes4q, frame, lo, hi, full_lo, full_hi, reach, max_entries=32, grow2=-1.0):
    """A tile's cut for rays of bounded reach (trace.hip k_tile_cuts with reach2 > 0): slots farther than `reach` from the tile's full
    box never join the list.  reach 0: orc_tile_cut's rule 1."""
    nodes4q = np.ascontiguousarray(nodes4q, dtype=np.uint32)
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    lo, hi = np.ascontiguousarray(lo, dtype=np.float32), np.ascontiguousarray(hi, dtype=np.float32)
    flo, fhi = np.ascontiguousarray(full_lo, dtype=np.float32), np.ascontiguousarray(full_hi, dtype=np.float32)
    out = np.zeros((128, 4), dtype=np.uint32)
    lib = load()
    lib.orc_tile_cut_reach.restype = C.c_int32
    m = lib.orc_tile_cut_reach(C.c_void_p(nodes4q.ctypes.data), C.c_uint64(len(nodes4q)), C.c_void_p(frame.ctypes.data), C.c_void_p(lo.ctypes.data),
                               C.c_void_p(hi.ctypes.data), C.c_void_p(flo.ctypes.data), C.c_void_p(fhi.ctypes.data), C.c_float(reach),
                               C.c_int32(max_entries), C.c_float(grow2), C.c_void_p(out.ctypes.data))
    return out[:m].copy()


def cut_walk_closest(nodes4q, frame, leaf_tris, rays, cut, order=1):
    """(stats, per-ray closest distance) of the closest-hit walk that starts below a tile's cut, the entered entries in `order`
    (1: the list's last first — what trace.hip does)."""
    nodes4q = np.ascontiguousarray(nodes4q, dtype=np.uint32)
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    leaf_tris, rays, cut = np.ascontiguousarray(leaf_tris), np.ascontiguousarray(rays), np.ascontiguousarray(cut, dtype=np.uint32)
    st = OrcEntryStats()
    t = np.zeros(max(1, len(rays)), dtype=np.float32)
    lib = load()
    lib.orc_cut_walk_closest.restype = C.c_int
    rc = lib.orc_cut_walk_closest(C.c_void_p(nodes4q.ctypes.data), C.c_uint64(len(nodes4q)), C.c_void_p(frame.ctypes.data), C.c_void_p(leaf_tris.ctypes.data),
                                  C.c_void_p(rays.ctypes.data), C.c_uint64(len(rays)), C.c_void_p(cut.ctypes.data), C.c_int32(len(cut)), C.c_int32(order),
                                  C.byref(st), C.c_void_p(t.ctypes.data))
    if rc != 0:
        raise RuntimeError(f"orc_cut_walk_closest failed: {rc}")
    return st, t[:len(rays)]


def walk4q_t(nodes4q, frame, leaf_tris, rays, any_hit=False):
    """(stats, per-ray closest distance — t_max where nothing was hit) of traverse4q from the root."""
    nodes4q = np.ascontiguousarray(nodes4q, dtype=np.uint32)
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    leaf_tris, rays = np.ascontiguousarray(leaf_tris), np.ascontiguousarray(rays)
    a = OrcTraceStats()
    t = np.zeros(max(1, len(rays)), dtype=np.float32)
    lib = load()
    lib.orc_walk4q_t.restype = C.c_int
    rc = lib.orc_walk4q_t(C.c_void_p(nodes4q.ctypes.data), C.c_uint64(len(nodes4q)), C.c_void_p(frame.ctypes.data), C.c_void_p(leaf_tris.ctypes.data),
                          C.c_void_p(rays.ctypes.data), C.c_uint64(len(rays)), C.c_int32(int(bool(any_hit))), C.byref(a), C.c_void_p(t.ctypes.data))
    if rc != 0:
        raise RuntimeError(f"orc_walk4q_t failed: {rc}")
    return a, t[:len(rays)]


def cut_walk(nodes4q, frame, leaf_tris, rays, cut, want_masks=False):
    """(stats, per-ray hit flags) of the any-hit walk that starts below a tile's cut."""
    nodes4q = np.ascontiguousarray(nodes4q, dtype=np.uint32)
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    leaf_tris, rays, cut = np.ascontiguousarray(leaf_tris), np.ascontiguousarray(rays), np.ascontiguousarray(cut, dtype=np.uint32)
    st = OrcEntryStats()
    hits = np.zeros(max(1, len(rays)), dtype=np.int32)
    masks = np.zeros(max(1, len(rays)), dtype=np.uint64)
    lib = load()
    lib.orc_cut_walk.restype = C.c_int
    rc = lib.orc_cut_walk(C.c_void_p(nodes4q.ctypes.data), C.c_uint64(len(nodes4q)), C.c_void_p(frame.ctypes.data), C.c_void_p(leaf_tris.ctypes.data),
                          C.c_void_p(rays.ctypes.data), C.c_uint64(len(rays)), C.c_void_p(cut.ctypes.data), C.c_int32(len(cut)), C.byref(st),
                          C.c_void_p(hits.ctypes.data), C.c_void_p(masks.ctypes.data))
    if rc != 0:
        raise RuntimeError(f"orc_cut_walk failed: {rc}")
    if want_masks:
        return st, hits[:len(rays)], masks[:len(rays)]
    return st, hits[:len(rays)]


def kary_visits(nodes2, rays, levels, pairs=True, any_hit=True):
    """(stats, box tests) of a 2^levels-ary collapse of the BVH2 walked on fp32 boxes."""
    nodes2, rays = np.ascontiguousarray(nodes2), np.ascontiguousarray(rays)
    st, boxes = OrcTraceStats(), C.c_uint64()
    rc = load().orc_kary_visits(nodes2.ctypes.data, len(nodes2), rays.ctypes.data, len(rays), levels, int(bool(pairs)), int(bool(any_hit)),
                                C.byref(st), C.byref(boxes))
    if rc != 0:
        raise RuntimeError(f"orc_kary_visits failed: {rc}")
    return st, boxes.value


def bvh_build_sah(tris):
    """Top-down binned-SAH BVH2 (measurement only)."""
    tris = np.ascontiguousarray(tris)
    n = len(tris)
    nodes = np.zeros(max(1, 2 * n), dtype=capi.node_dtype())
    depth = C.c_uint32()
    cnt = int(load().orc_bvh_build_sah(tris.ctypes.data, n, nodes.ctypes.data, C.byref(depth)))
    return nodes[:cnt], depth.value


def default_options():
    """The reference's constants (BakeEngine.cpp:13-18), stated independently of the library."""
    o = capi.AobOptions()
    o.sample_count = 100
    o.sampler = capi.SAMPLER_REFERENCE_COMPAT
    o.seed = 0
    o.use_any_hit = 1
    o.falloff_distance = 1.25
    o.ray_tmax = 1000000.0
    o.ray_bias = 0.001
    o.bleed_distance = 5
    o.accumulate = capi.ACCUMULATE_QUANTISED
    o.normal_fp16 = capi.FP16_RNE
    o.ray_chunk_size = 0xFFFFFF
    o.clamp_tmax_to_falloff = 0
    o.bvh_width = 4
    o.bvh_quantised = 1
    o.bvh_builder = -1         # by workload (orc_resolve_bvh_builder / aob_resolve_bvh_builder)
    o.morton_adaptive = 1
    # the schedule of the library's trace kernel (no meaning for the oracle; stated so that the two defaults compare equal)
    o.trace_refill_idle, o.trace_leaf_wait, o.trace_parts, o.trace_part_min, o.trace_fine_div, o.trace_bands, o.trace_scalar_min, o.trace_cut = 0, 0, 4, 16, 8, 1, 0, 0
    return o
