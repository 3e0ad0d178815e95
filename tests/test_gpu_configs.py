"""BASELINE.json configs[3] and configs[4] at their full sizes, on one GPU, and the C++ multi-GPU host.

configs[3]: 999 698-triangle terrain, ONE 4096^2 atlas, 256 spp, cut into strips of texel rows as 8 GPUs would cut it
            (each strip rasterised and traced on its own, `aob_bake_rasterize_rows`): stitched == unsharded, and the
            oracle on a pseudo-random subset of the texels.  The same through the C++ host (`aobake_mgpu`: one thread per
            rank; 8 ranks rehearsed on the one device of the test box).
configs[4]: 100 rocks x 100 352 triangles + ground = 10 035 202 triangles (every copy a full shape with its own transform,
            as the reference attaches them — it never calls CreateInstance), 8192^2 atlas, closest-hit falloff scoring with
            the ray length clamped to the falloff distance: tree, surfels and a sampled trace against the oracle.
"""
import json
import os
import subprocess

import numpy as np
import pytest

from aobaker_amd import Baker, capi, scenefile, scenes
from aobaker_amd.scene import FlatScene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MGPU = os.path.join(ROOT, "aobaker_amd", "host", "aobake_mgpu")
CORES = os.cpu_count() or 1


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8).tobytes()


def cosine_options(orc, spp, **kw):
    o = orc.default_options()
    o.sample_count, o.sampler, o.accumulate, o.normal_fp16, o.seed = spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, 3
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def pseudo_random_subset(surfels, one_in):
    x, y = surfels["texel_x"].astype(np.uint32), surfels["texel_y"].astype(np.uint32)
    h = x * np.uint32(0x9E3779B1) ^ y * np.uint32(0x85EBCA77)
    h ^= h >> np.uint32(15)
    h *= np.uint32(0x2C1B3C6D)
    h ^= h >> np.uint32(13)
    return h % np.uint32(one_in) == 0


def test_config4_fixed_atlas_in_8_row_windows(engine, orc):
    """BASELINE configs[3] (the fourth of five): 4096^2 x 256 spp = 4.29 G rays."""
    from aobaker_amd import dist as aod
    W = H = 4096
    flat = FlatScene(scenes.terrain_scene(grid=708, atlas=W, seed=3, bpc=4))
    opt = cosine_options(orc, 256)
    with Baker(flat, opt) as b:
        b.build_bvh()
        b.rasterize(0)
        assert b.surfel_count(0) == W * H
        b.trace(0)
        full = b.results(0)
        ms_full = b.last_trace_ms()
        surfels = b.surfels(0)
        # 8 strips, equal and then cut where a cost-balanced run would cut them: any cut gives the same cells
        for bounds in (aod.equal_rows(H, 8), aod.rebalance_rows(aod.equal_rows(H, 8), [30, 36, 37, 37, 37, 37, 36, 30], H), [0, 8, 8, 1000, 1000, 4000, 4088, 4096, 4096]):
            stitched = np.zeros(W * H, dtype=capi.result_dtype())
            total = 0
            for r in range(8):
                b.rasterize(0, (bounds[r], bounds[r + 1]))
                n = b.surfel_count(0)
                assert n == (bounds[r + 1] - bounds[r]) * W
                total += n
                b.trace(0)
                part = b.results(0)
                lo, hi = bounds[r] * W, bounds[r + 1] * W
                # a rank's cells are its strip and nothing else
                assert not part["weight_bits"][:lo].any() and not part["weight_bits"][hi:].any()
                stitched[lo:hi] = part[lo:hi]
            assert total == W * H
            assert bits(stitched) == bits(full)
    w = full["weight_bits"].view(np.float32)
    t = full["total_bits"].view(np.float32)
    assert (w == 256.0).all() and (t == np.round(t)).all() and 0.5 < (t / w).mean() < 0.95
    # the oracle on a pseudo-random 1/1024 of the texels, all 256 rays each, exact
    keep = pseudo_random_subset(surfels, 1024)
    sub = np.ascontiguousarray(surfels[keep])
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    ref, st = orc.trace(nodes, sub, None, opt, W, H, threads=CORES)
    cells = sub["texel_y"].astype(np.int64) * W + sub["texel_x"]
    assert 10000 < len(sub) < 24000 and st.rays == len(sub) * 256
    assert np.array_equal(full["total_bits"][cells], ref["total_bits"][cells])
    assert np.array_equal(full["weight_bits"][cells], ref["weight_bits"][cells])
    print(f"configs[3] on one GPU: trace {ms_full:.1f} ms, {W * H * 256 / ms_full / 1e6:.2f} Grays/s")
    # ... and the resolved image is the one `bench.py --gpus N` (N > 1) checks its own against (config.image_crc32_expected)
    import zlib
    import bench
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0); b.resolve(0)
        img = np.zeros(W * H * 4, dtype=np.uint8)
        b.download(0, img)
    assert zlib.crc32(img.tobytes()) == bench.CONFIG3_IMAGE_CRC32


def run_mgpu(args, timeout=900):
    out = subprocess.run([MGPU] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.returncode, out.stderr[-2000:], out.stdout[-500:])
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("devices,sampler,scene_name", [
    ("0", 1, "terrain"),                 # one device: the RCCL-free single-rank path of the same host code
    ("0,0,0", 1, "terrain"),             # three ranks on the one device: strips of rows, cost re-cuts, gather, resolve on rank 0
    ("0,0,0,0,0,0,0,0", 1, "two_section"),   # eight ranks, overlapping sections of one material, some strips empty
    ("0,0", 0, "cube"),                  # the reference's own sampler: the surfel list is sharded, not the rows
    ("0,0,0", 0, "two_section"),         # ... with overlapping sections: ranges of different ranks share cells, summed on rank 0
])
def test_cpp_multi_gpu_host_matches_single_device(engine, tmp_path, devices, sampler, scene_name):
    """aobaker_amd/host/MultiGpu.cpp through tools/aobake_mgpu: N ranks (threads) bake strips, rank 0 gathers and resolves;
    the downloaded image is byte-identical to the single-device aob_bake of the same scene (--check-single)."""
    sc = {"terrain": lambda: scenes.terrain_scene(grid=64, atlas=(256, 200), seed=3, bpc=4),
          "two_section": lambda: scenes.two_section_scene(40, bpc=4),
          "cube": lambda: scenes.cube_scene(96)}[scene_name]()
    path = tmp_path / "scene.aobs"
    scenefile.write_scene(sc, str(path))
    rec = run_mgpu(["--scene", path, "--devices", devices, "--steps", 2, "--warmup", 2, "--spp", 32, "--sampler", sampler,
                    "--accumulate", 1 if sampler else 0, "--seed", 3, "--normal-fp16", 0, "--check-single"])
    n = len(devices.split(","))
    assert rec["n_gpus"] == n and rec["matches_single_device"] is True and rec["rays"] > 0
    assert len(rec["rank_ms"]) == n and len(rec["rows"]) == n
    assert rec["rows"][0][0] == 0 and all(a[1] == b[0] for a, b in zip(rec["rows"], rec["rows"][1:]))


def test_cpp_multi_gpu_host_names_a_device_that_is_not_there(engine, tmp_path):
    """A device ordinal the box does not have (what the first run on a node with fewer GPUs than asked for would meet): the host
    says which, and exits non-zero, before any thread or communicator exists — no hang, no fault."""
    import torch
    path = tmp_path / "scene.aobs"
    scenefile.write_scene(scenes.cube_scene(32), str(path))
    missing = torch.cuda.device_count()
    out = subprocess.run([MGPU, "--scene", str(path), "--devices", f"0,{missing}", "--steps", "1", "--warmup", "0", "--spp", "4"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert f"device {missing} is not on this box" in out.stderr, out.stderr[-500:]


def test_cpp_multi_gpu_host_config4_shape(engine, tmp_path):
    """configs[3] through the C++ host: 4096^2 x 256 spp in 8 strips (8 ranks rehearsed on one device), image identical to the
    single-device bake; the re-cut strips are uneven (the terrain's border rows are cheaper), the ranks' times level."""
    sc = scenes.terrain_scene(grid=708, atlas=4096, seed=3, bpc=4)
    path = tmp_path / "scene.aobs"
    scenefile.write_scene(sc, str(path))
    rec = run_mgpu(["--scene", path, "--devices", "0,0,0,0,0,0,0,0", "--steps", 2, "--warmup", 3, "--spp", 256, "--sampler", 1,
                    "--accumulate", 1, "--seed", 3, "--normal-fp16", 0, "--check-single"])
    assert rec["matches_single_device"] is True and rec["rays"] == 4096 * 4096 * 256 and rec["triangles"] == 999698
    import bench
    assert rec["image_crc32"] == [bench.CONFIG3_IMAGE_CRC32]
    # the step's anatomy is in the record: every rank's redundant tree build, its trace kernel, and what the gather cost rank 0
    assert len(rec["rank_bvh_build_ms"]) == 8 and all(0.0 < v < 50.0 for v in rec["rank_bvh_build_ms"]) and len(rec["rank_trace_kernel_ms"]) == 8
    assert rec["gather_ms"] > 0.0 and len(rec["rank_gather_ms"]) == 8
    print("aobake_mgpu 8 ranks on one device:", json.dumps({k: rec[k] for k in ("ms_per_step", "rank_ms", "rows")}))


def test_config5_instanced_scene_at_size(engine, orc):
    """BASELINE configs[4] (the fifth): 10 M triangles, 8192^2 atlas, falloff scoring with max ray distance = falloff distance.
    Tree and surfel list bit-exact against the oracle at full size, the trace at 64 spp on a pseudo-random texel subset."""
    sc = scenes.instanced_scene(instances=100, rock_n=224, atlas=8192, seed=5, bpc=4)
    flat = FlatScene(sc)
    assert flat.triangle_count() == 100 * 2 * 224 * 224 + 2
    opt = cosine_options(orc, 64, use_any_hit=0, falloff_distance=1.25, clamp_tmax_to_falloff=1, seed=5)
    tris = orc.scene_triangles(flat.desc)
    ref_nodes, ref_depth = orc.bvh_build(tris)
    del tris
    with Baker(flat, opt) as b:
        b.build_bvh()
        st = b.stats()
        assert (st.triangle_count, st.node_count, st.bvh_depth) == (10035202, 20070403, ref_depth)
        nodes = b.nodes()
        assert bits(nodes) == bits(ref_nodes)
        del nodes
        b.rasterize(0)
        n = b.surfel_count(0)
        surfels = b.surfels(0)
        ref_surfels, sidx = orc.rasterize(flat.desc, 0, opt)
        assert n == len(ref_surfels) and 0.8 * 8192 * 8192 < n < 8192 * 8192
        assert bits(surfels) == bits(ref_surfels)
        del ref_surfels
        b.trace(0)
        got = b.results(0)
        ms = b.last_trace_ms()
    # the oracle on a pseudo-random 1/8192 of the texels: all 64 closest-hit rays each, exact Q32.32 float sums
    keep = pseudo_random_subset(surfels, 8192)
    sub = np.ascontiguousarray(surfels[keep])
    ref, rst = orc.trace(ref_nodes, sub, None, opt, 8192, 8192, threads=CORES)
    cells = sub["texel_y"].astype(np.int64) * 8192 + sub["texel_x"]
    assert 4000 < len(sub) < 12000
    assert np.array_equal(got["total_bits"][cells], ref["total_bits"][cells])
    assert np.array_equal(got["weight_bits"][cells], ref["weight_bits"][cells])
    ao = got["total_bits"].view(np.float32)[cells] / got["weight_bits"].view(np.float32)[cells]
    assert 0.2 < ao.mean() < 0.9 and rst.hits > 0
    print(f"configs[4] on one GPU at 64 spp: depth {st.bvh_depth}, {n} surfels, trace {ms:.1f} ms, {n * 64 / ms / 1e6:.2f} Grays/s, "
          f"oracle visits/ray inner {rst.inner_visits / rst.rays:.2f} leaf {rst.leaf_visits / rst.rays:.2f}")
    del got
    # ... and the configuration's own sample count: all 1024 spp of every texel (~6 x 10^10 closest-hit rays, one GPU), the
    # oracle on a pseudo-random 1/32768 of the texels with all 1024 rays each
    opt_full = cosine_options(orc, 1024, use_any_hit=0, falloff_distance=1.25, clamp_tmax_to_falloff=1, seed=5)
    with Baker(flat, opt_full) as b:
        b.build_bvh(); b.rasterize(0)
        assert b.surfel_count(0) == n
        b.trace(0)
        full = b.results(0)
        ms_full = b.last_trace_ms()
    keep = pseudo_random_subset(surfels, 32768)
    sub = np.ascontiguousarray(surfels[keep])
    assert 1000 < len(sub) < 3000
    ref, _ = orc.trace(ref_nodes, sub, None, opt_full, 8192, 8192, threads=CORES)
    cells = sub["texel_y"].astype(np.int64) * 8192 + sub["texel_x"]
    assert np.array_equal(full["total_bits"][cells], ref["total_bits"][cells])
    assert np.array_equal(full["weight_bits"][cells], ref["weight_bits"][cells])
    all_cells = surfels["texel_y"].astype(np.int64) * 8192 + surfels["texel_x"]
    w = full["weight_bits"].view(np.float32)[all_cells]
    assert ((w == 1024.0) | (w == 0.0)).all() and (w == 1024.0).mean() > 0.999     # unit weights: a texel with a tangent frame cast its 1024 rays
    print(f"configs[4] on one GPU at its 1024 spp: trace {ms_full:.0f} ms, {n * 1024 / ms_full / 1e6:.2f} Grays/s")
