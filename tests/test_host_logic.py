"""Host-side logic above the C ABI: the scene mirror, flattening, transforms, sharding maths."""
import ctypes as C

import numpy as np
import pytest

from aobaker_amd import capi, scenes
from aobaker_amd import dist as aod
from aobaker_amd.scene import (RENDER_VERTEX, BakeObject, BakeTexture, FlatScene, MaterialObject, RenderMesh, Scene,
                               Section, make_vertices)


def test_bake_texture_contract():
    # Scene/BakeTexture.cpp:7-67: valid (channels, bytes) combinations, 0xFF fill, tight packing
    t = BakeTexture(7, 5, 4, 2)
    assert t.GetDataLength() == 7 * 5 * 4 * 2 and (t.GetData() == 0xFF).all()
    assert t.as_array().shape == (5, 7, 4) and t.as_array().dtype == np.float16
    for bad in ((3, 1), (3, 2), (5, 4), (1, 3)):
        with pytest.raises(ValueError):
            BakeTexture(4, 4, *bad)


def test_render_vertex_layout_and_flatten():
    assert RENDER_VERTEX.itemsize == 48
    assert [RENDER_VERTEX.fields[n][1] for n in ("Position", "Normal", "UV", "Color")] == [0, 12, 24, 32]
    sc = scenes.two_section_scene(16)
    flat = FlatScene(sc)
    assert flat.desc.mesh_count == 2 and flat.desc.material_count == 1
    m0 = flat.mesh_array[0]
    assert (m0.position_stride, m0.normal_stride, m0.uv_stride) == (48, 48, 48)
    assert m0.normals - m0.positions == 12 and m0.uvs - m0.positions == 24
    assert m0.vertex_count == 4 and m0.index_count == 6 and m0.material == 0
    # the descriptor aliases the scene's own arrays: the bake writes straight into BakeTexture.Data
    assert flat.material_array[0].data == sc.GetMaterial(0).GetAOTexture().GetData().ctypes.data
    assert flat.triangle_count() == 4


def test_material_association_is_by_identity():
    sc = Scene()
    a = sc.AddMaterial(MaterialObject("same", 8, 8)); b = sc.AddMaterial(MaterialObject("same", 8, 8))
    q = scenes.quad((0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0))
    o = sc.AddObject(BakeObject("O"))
    o.AddSection(Section("s", RenderMesh(make_vertices(*q[:3]), q[3]), b))
    o.AddSection(Section("t", RenderMesh(make_vertices(*q[:3]), q[3]), None))
    flat = FlatScene(sc)
    assert flat.mesh_array[0].material == 1 and flat.mesh_array[1].material == -1
    assert a is not b


def test_build_transform_matches_simplemath_convention():
    # TransformObject.cpp:16-24: Scale * Rotation(yaw=rot.y, pitch=rot.x, roll=rot.z) * Translation, row vectors
    o = BakeObject("O", position=(1, 2, 3), rotation=(0, np.pi / 2, 0), scale=(2, 2, 2))
    m = o.BuildTransform()
    p = np.float32([1, 0, 0, 1]) @ m
    # yaw of +90 degrees about Y (left-handed, row vectors) sends +X to -Z
    assert np.allclose(p[:3], [1, 2, 3 - 2], atol=1e-6)
    o = BakeObject("O", rotation=(np.pi / 2, 0, 0))
    p = np.float32([0, 0, 1, 0]) @ o.BuildTransform()
    assert np.allclose(p[:3], [0, -1, 0], atol=1e-6)       # pitch +90 sends +Z to -Y
    assert np.allclose(BakeObject("I").BuildTransform(), np.eye(4))


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 64, 4194304, 4194305):
        for world in (1, 2, 3, 4, 8):
            parts = [aod.shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1


def test_weak_scaling_atlas():
    assert aod.weak_scaling_atlas(2048, 1) == (2048, 2048)
    assert aod.weak_scaling_atlas(2048, 2) == (2048, 4096)
    assert aod.weak_scaling_atlas(2048, 4) == (4096, 4096)       # BASELINE configs[3]
    assert aod.weak_scaling_atlas(2048, 8) == (4096, 8192)
    for n in (1, 2, 3, 4, 6, 8):
        w, h = aod.weak_scaling_atlas(128, n)
        assert w * h == n * 128 * 128


def test_terrain_scene_matches_baseline_config():
    sc = scenes.terrain_scene(grid=708, atlas=64)
    mesh = sc.GetObject(0).GetSection(0).GetPreviewMesh()
    assert mesh.GetIndexCount() // 3 == 999698
    v = mesh.GetVertices()
    assert np.allclose(np.linalg.norm(v["Normal"], axis=1), 1.0, atol=1e-5)
    assert v["UV"].min() == 0.0 and v["UV"].max() == 1.0
    again = scenes.terrain_scene(grid=708, atlas=64).GetObject(0).GetSection(0).GetPreviewMesh().GetVertices()
    assert np.array_equal(v, again)                              # seeded, reproducible


def test_sphere_scene_triangle_count():
    mesh = scenes.sphere_scene(n=187, atlas=64).GetObject(0).GetSection(0).GetPreviewMesh()
    assert mesh.GetIndexCount() // 3 == 69938                    # BASELINE.md cfg2 stand-in


def test_scene_file_roundtrip(tmp_path):
    import struct
    from aobaker_amd.scenefile import write_scene
    sc = scenes.cube_scene(32)
    path = tmp_path / "cube.aobs"
    write_scene(sc, str(path))
    raw = path.read_bytes()
    assert raw[:4] == b"AOBS" and struct.unpack("<III", raw[4:16]) == (1, 1, 1)
    mesh = sc.GetObject(0).GetSection(0).GetPreviewMesh()
    assert len(raw) == 16 + 16 + 36 + 4 + 12 + 48 * mesh.GetVertexCount() + 4 * mesh.GetIndexCount()


def test_bench_helpers_and_committed_counters():
    """bench.py's roofline inputs that do not need a GPU: the committed PMC record is used only for the kernel source and
    the configuration it was measured on, and the texel subsample is a fixed pseudo-random set."""
    import json
    import os
    import types
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    opts = capi.default_options()
    opts.bvh_builder = 2                                           # what the library's default (-1: by workload) settles on for BASELINE configs[2]
    args = types.SimpleNamespace(spp=256, grid=708)
    rec, why = bench.counter_record(opts, args, 2048, 2048, 1)
    path = os.path.join(root, "profiles", "k_trace_counters.json")
    if os.path.exists(path):
        committed = json.load(open(path))
        if committed["source_sha256_16"] == bench.kernel_source_hash():
            assert rec is not None and 1e9 < rec.get("fabric_side_bytes_per_launch", rec.get("hbm_side_bytes_per_launch")) < 1e12
            roof = rec["roof"]                                     # BOTH roofs as throughput fractions, the larger one named
            assert roof["bound"] in ("vmem", "valu") and 0 < roof["frac"] <= 1.3
            assert roof["frac"] == max(roof["vmem"]["frac"], roof["valu"]["frac"])
            assert roof["vmem"]["marginal_cu_cycles_per_instruction"]["mean"] > 0
            assert rec["config"] == bench.counter_key(opts, args, 2048, 2048, 1) and rec["rays_per_launch"] == 2048 * 2048 * 256
        else:
            assert rec is None and "kernel source" in why        # a stale record is refused, not reported
    else:
        assert rec is None
    other, why = bench.counter_record(opts, types.SimpleNamespace(spp=64, grid=708), 1024, 1024, 1)
    assert other is None                                           # another configuration never inherits the figure
    requested = {"node_fetches": 6.0, "triangle_fetches": 1.9, "cut_entries_tested": 30.0, "cut_entries_entered": 2.2, "counted_on": "test"}
    footprint = {"bytes": 800e6, "of_which": {}}
    r = bench.roofline_record(opts, args, 2048, 2048, 1, (15.0, 1.8), 1 << 30, 76.0, requested, footprint)
    assert r["bytes_per_ray_algorithmic"] == round(64.0 * 16.8 + 40.0 / 256, 1)             # SURVEY §8(d) as written ...
    assert r["algorithmic_frac"] == round((1 << 30) * (64.0 * 16.8 + 40.0 / 256) / 76e-3 / 1e9 / 8000.0, 4)      # ... kept, and allowed above 1
    # the roofline fraction proper: bytes the lanes request from the vector L1 against its 64 B / clk / CU — physically bounded
    b_req = 64.0 * 6.0 + 48.0 * 1.9 + 16.0 + 48.0 + 8.0 / 256
    assert r["requested"]["bytes_per_ray"] == round(b_req, 1)
    assert r["frac"] == round(round((1 << 30) * b_req / 76e-3 / 1e9, 1) / (64.0 * 256 * 2.4), 4) and 0 < r["frac"] <= 1
    assert r["achieved"] == r["requested"]["GBps"] and r["peak"] == round(64.0 * 256 * 2.4, 1) and r["unit"] == "GB/s"
    assert all(v is None or 0 <= v <= 1 for v in r["byte_rate_fractions"].values())
    if r["fabric"]:
        assert r["refetch_ratio"] == round(r["fabric"]["bytes_per_launch"] / 800e6, 1) and r["traffic"] == r["fabric"]["bytes_per_launch"]
    surfels = np.zeros(64 * 64, dtype=capi.surfel_dtype())
    surfels["texel_x"], surfels["texel_y"] = np.tile(np.arange(64), 64), np.repeat(np.arange(64), 64)
    sub = bench._subsample(surfels, 16)
    assert 0.7 * 256 < len(sub) < 1.3 * 256 and len(bench._subsample(surfels, 1)) == len(surfels)
    again = bench._subsample(surfels[::-1], 16)                  # a fixed subset of the texels, whatever the list order
    assert set(zip(sub["texel_x"].tolist(), sub["texel_y"].tolist())) == set(zip(again["texel_x"].tolist(), again["texel_y"].tolist()))
    on_lattice = np.mean((sub["texel_x"] % 8 == 0) & (sub["texel_y"] % 8 == 0))
    assert on_lattice < 0.06                                     # not aligned with the terrain's noise lattice (1/64 expected)


def test_counter_records_and_scaling_bases_of_other_runs_are_refused(tmp_path):
    """A PMC record describes the run it was measured on: bench.py attaches it only where kernel source AND configuration match
    (a record taken on another builder says so in the line), tools/pmc_record.py copies the configuration from the profiled
    run's own JSON line and refuses to write a record without one, and a strong-scaling base from another kernel is refused."""
    import json
    import os
    import sys
    import types
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    opts = capi.default_options()
    opts.bvh_builder = 2
    args = types.SimpleNamespace(spp=256, grid=708)
    key = bench.counter_key(opts, args, 2048, 2048, 1)
    good = {"source_sha256_16": bench.kernel_source_hash(), "config": key, "fabric_side_bytes_per_launch": 4e10, "counters": {"TCP_TCC_READ_REQ_sum": 4e9}, "roof": None}
    path = tmp_path / "rec.json"
    path.write_text(json.dumps(good))
    rec, why = bench.counter_record(opts, args, 2048, 2048, 1, str(path))
    assert rec is not None and why is None
    other = dict(good, config=dict(key, bvh_builder=1))            # the same kernel, profiled with another builder
    path.write_text(json.dumps(other))
    rec, why = bench.counter_record(opts, args, 2048, 2048, 1, str(path))
    assert rec is None and why == "counters were measured on another configuration"
    line = bench.roofline_record(opts, args, 2048, 2048, 1, (14.2, 1.8), 1 << 30, 50.0)
    assert line["counter_key"] == key and line["rays_per_launch"] == 1 << 30
    path.write_text(json.dumps(dict(good, config=dict(key, trace_cut=1))))        # ... or with the tile cuts switched off
    assert bench.counter_record(opts, args, 2048, 2048, 1, str(path)) == (None, "counters were measured on another configuration")
    path.write_text(json.dumps(dict(good, source_sha256_16="0123456789abcdef")))
    rec, why = bench.counter_record(opts, args, 2048, 2048, 1, str(path))
    assert rec is None and "kernel source" in why
    # tools/pmc_record.py: the configuration comes from the profiled run's line
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import pmc_record
    finally:
        sys.path.pop(0)
    d = tmp_path / "pmc"
    d.mkdir()
    (d / "pass1.log").write_text("noise\n" + json.dumps({"metric": "AO rays/sec", "roofline": line}) + "\n")
    (d / "pass2.log").write_text(json.dumps({"metric": "AO rays/sec", "roofline": line}) + "\n")
    assert pmc_record.profiled_run(str(d)) == (key, float(1 << 30))
    (d / "pass2.log").write_text(json.dumps({"metric": "AO rays/sec", "roofline": dict(line, counter_key=dict(key, bvh_builder=1))}) + "\n")
    with pytest.raises(SystemExit):
        pmc_record.profiled_run(str(d))                            # two passes, two configurations
    (d / "pass2.log").unlink()
    (d / "pass1.log").write_text("a program that printed no bench line\n")
    with pytest.raises(SystemExit):
        pmc_record.profiled_run(str(d))
    # the strong-scaling base
    base = tmp_path / "base.json"
    base.write_text(json.dumps({"atlas": 4096, "spp": 256, "kernel_source_sha256_16": bench.kernel_source_hash(), "ms_per_step": 200.0}))
    assert bench.strong_scaling_base(4096, 256, str(base))["ms_per_step"] == 200.0
    base.write_text(json.dumps({"atlas": 4096, "spp": 256, "kernel_source_sha256_16": "0123456789abcdef", "ms_per_step": 200.0}))
    assert "refused" in bench.strong_scaling_base(4096, 256, str(base))
    assert bench.strong_scaling_base(2048, 256, str(base)) is None


def test_bench_launches_its_own_ranks_or_says_why_not():
    """`python bench.py --gpus 2` without a launcher starts its own rank processes; on a box with fewer devices (here:
    none) it says so in one JSON line and exits 0 instead of dying on a missing WORLD_SIZE."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] is None and "needs 2 devices" in line["skipped"]


def test_bench_supervisor_ends_the_other_ranks_when_one_dies():
    """bench.supervise: a rank that exits non-zero must not leave its peers waiting in a collective — the parent terminates
    them within seconds and names the rank; a clean run returns every rank's 0 and rank 0's stdout; a run past its deadline
    is ended too."""
    import sys
    import time
    import bench
    py = sys.executable
    hang = [py, "-c", "import time; print('started', flush=True); time.sleep(600)"]
    t = time.monotonic()
    codes, out0, errs, reason = bench.supervise([(hang, None), ([py, "-c", "import sys; sys.stderr.write('boom'); sys.exit(3)"], None), (hang, None)], 120.0)
    assert time.monotonic() - t < 30.0
    assert reason == "rank 1 exited with code 3" and codes[1] == 3 and codes[0] != 0 and codes[2] != 0
    assert "started" in out0 and "boom" in errs[1]
    codes, out0, errs, reason = bench.supervise([([py, "-c", "print('{\"ok\": 1}')"], None), ([py, "-c", "pass"], None)], 60.0)
    assert codes == [0, 0] and reason is None and out0.strip() == '{"ok": 1}'
    t = time.monotonic()
    codes, _, _, reason = bench.supervise([(hang, None)], 1.0)
    assert time.monotonic() - t < 20.0 and "still running" in reason and codes[0] != 0


def test_rebalance_rows_levels_the_ranks_and_matches_the_cpp_host():
    """aobaker_amd.dist.rebalance_rows == aob_rebalance_rows (MultiGpu.cpp), and re-cutting by measured cost converges."""
    import ctypes as C
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = C.CDLL(os.path.join(root, "aobaker_amd", "host", "libaobaker_host.so"))
    rng = np.random.default_rng(4)

    def cpp_rebalance(bounds, ms, height):
        n = len(ms)
        out = (C.c_int32 * (n + 1))()
        assert host.aobhost_rebalance_rows((C.c_int32 * (n + 1))(*bounds), (C.c_double * n)(*ms), n, height, 8, out) == 0
        return list(out)

    for height in (64, 1000, 2048, 4096, 8192):
        for parts in (1, 2, 3, 4, 8):
            eq = aod.equal_rows(height, parts)
            out = (C.c_int32 * (parts + 1))()
            assert host.aobhost_equal_rows(height, parts, 8, out) == 0 and list(out) == eq
            assert eq[0] == 0 and eq[-1] == height and all(a <= b for a, b in zip(eq, eq[1:])) and all(b % 8 == 0 for b in eq[:-1])
            for _ in range(5):
                ms = rng.uniform(0.0, 50.0, parts).tolist()
                got = aod.rebalance_rows(eq, ms, height)
                assert got == cpp_rebalance(eq, ms, height)
                assert got[0] == 0 and got[-1] == height and all(a <= b for a, b in zip(got, got[1:]))
    # a cost profile like the terrain's (cheap borders): measure -> re-cut converges to level ranks
    height, parts = 4096, 8
    density = 1.0 - 0.35 * np.abs(np.linspace(-1, 1, height)) ** 4          # cost per row
    cum = np.concatenate([[0.0], np.cumsum(density)])
    bounds = aod.equal_rows(height, parts)
    spread = []
    for _ in range(4):
        ms = [cum[bounds[r + 1]] - cum[bounds[r]] for r in range(parts)]
        spread.append(max(ms) / (sum(ms) / parts))
        bounds = aod.rebalance_rows(bounds, ms, height)
    assert spread[0] > 1.05 and spread[-1] < 1.02            # cuts snap to 8-row tile rows: 1.6 % of a 512-row strip


def test_row_windows_partition_the_atlas_on_tile_rows():
    for height in (8, 64, 100, 2048, 4096):
        for world in (1, 2, 3, 4, 8):
            w = [aod.row_window(height, r, world) for r in range(world)]
            assert w[0][0] == 0 and w[-1][1] == height
            assert all(a[1] == b[0] for a, b in zip(w, w[1:]))
            assert all(b % 8 == 0 for b, _ in w)
            sizes = [e - b for b, e in w]
            assert max(sizes) - min(sizes) <= 8 or height % 8
