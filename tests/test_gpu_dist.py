"""GPU side of the multi-GPU path that one box can check: the raw result buffer aliased as a torch
tensor (zero-copy), and an RCCL collective running on that foreign allocation."""
import os

import numpy as np
import pytest

from aobaker_amd import Baker, capi, scenes
from aobaker_amd.scene import FlatScene

pytestmark = pytest.mark.gpu


def test_results_alias_and_rccl_on_library_memory(engine):
    import torch
    import torch.distributed as dist
    from aobaker_amd import dist as aod

    flat = FlatScene(scenes.terrain_scene(grid=64, atlas=128, bpc=4))
    opt = capi.default_options()
    opt.sample_count = 8
    opt.sampler, opt.accumulate, opt.seed = capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, 2
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0); b.sync()
        host = b.results(0)
        cells = aod.alias_results(b, 0, as_float=True)
        assert cells.shape == (128 * 128, 2) and cells.is_cuda
        assert np.array_equal(cells.cpu().numpy().view(np.uint32), host.view(np.uint32).reshape(-1, 2))
        # writes through the alias land in the library's buffer
        before = cells[5].clone()
        cells[5] = torch.tensor([3.0, 4.0], device="cuda")
        torch.cuda.synchronize()
        assert b.results(0).view(np.float32).reshape(-1, 2)[5].tolist() == [3.0, 4.0]
        cells[5] = before
        torch.cuda.synchronize()
        # an RCCL collective on it (world of one: the transport is exercised, the data unchanged)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29631")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        try:
            dist.all_reduce(cells)
            aod.gather_strips(cells, (0, 10), 0, 1)
            torch.cuda.synchronize()
        finally:
            dist.destroy_process_group()
        assert np.array_equal(b.results(0).view(np.uint32), host.view(np.uint32))
        b.resolve(0)
        img = np.zeros(128 * 128 * 4, dtype=np.uint8)
        b.download(0, img)
        assert 0.3 < img.view(np.float32).mean() < 1.0


def _bench(*args):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *[str(a) for a in args]], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, (out.returncode, out.stderr[-3000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("ranks", [2, 3, 5])          # (5: with the test process itself, the GPU pool's limit of six processes on one card;
def test_bench_rank_processes_rehearsed_on_one_gpu(ranks):  # eight ranks are rehearsed on the CPU: tests/test_dist_gloo.py, and by the C++ host)
    """`python bench.py --gpus N` end to end with N real rank processes — self-launch, strips of texel rows, cost re-cuts
    during warm-up, gather to rank 0, resolve — rehearsed on the one GPU of the test box: the ranks share the device and the
    strips travel through host memory with gloo (RCCL wants one rank per device); every other line of the N-rank path is
    the one the driver's multi-GPU run executes.  The resolved atlas equals the one-rank atlas."""
    common = ["--atlas", 256, "--grid", 96, "--spp", 32, "--steps", 2, "--warmup", 3, "--no-cpp-host", "--no-cpu-baseline", "--no-wallclock"]
    one = _bench("--gpus", 1, *common)
    many = _bench("--gpus", ranks, "--backend", "gloo", "--oversubscribe", *common)
    assert many["n_gpus"] == ranks and many["scaling"] == "strong" and many["value"] > 0
    assert many["config"]["atlas"] == [256, 256] and many["config"]["surfels"] == one["config"]["surfels"] == 256 * 256
    rows = many["config"]["rows"]
    assert len(rows) == ranks and rows[0][0] == 0 and rows[-1][1] == 256 and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    assert len(many["balance"]["warmup_history"]) == 3 and len(many["stage_ms"]["rank_build_raster_trace"]) == ranks
    assert many["config"]["image_crc32"] == one["config"]["image_crc32"]


def test_the_cpp_host_carries_the_headline_of_a_multi_gpu_line():
    """with N > 1 `value` is the C++ host's (aobake_mgpu: one process, a thread + RCCL rank per device) — here rehearsed with two
    ranks on the one device — the python ranks' figures beside it, both with the one-rank image."""
    common = ["--atlas", 256, "--grid", 96, "--spp", 32, "--steps", 2, "--warmup", 2, "--no-cpu-baseline", "--no-wallclock"]
    one = _bench("--gpus", 1, "--no-cpp-host", *common)
    two = _bench("--gpus", 2, "--backend", "gloo", "--oversubscribe", *common)
    assert two["value_from"] == "cpp_host", two.get("value_from")
    assert two["value"] == two["cpp_host"]["value"] and two["ms_per_step"] == two["cpp_host"]["ms_per_step"]
    assert two["python_ranks"]["value"] > 0 and "C++" in two["config"]["host"]
    assert two["config"]["image_crc32"] == one["config"]["image_crc32"] == two["cpp_host"]["image_crc32"][0]
    assert len(two["stage_ms"]["rank_build_raster_trace"]) == 2


def test_bench_rank_failure_ends_the_run_with_an_error_line():
    """A rank that dies before the first gather (fault injected: AOB_BENCH_FAIL_RANK) ends the whole N-rank run in seconds —
    one JSON line with value null and the failing rank named, exit code non-zero — instead of the peers waiting in the
    gather until a collective timeout."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["AOB_BENCH_FAIL_RANK"] = "1"
    t = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--oversubscribe", "--atlas", "256", "--grid", "96",
                          "--spp", "32", "--steps", "1", "--warmup", "1", "--no-cpp-host", "--no-cpu-baseline", "--no-wallclock"],
                         capture_output=True, text=True, env=env, timeout=600)
    took = time.monotonic() - t
    assert out.returncode != 0
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["value"] is None and "exited with code" in line["error"] and line["rank_exit_codes"][1] != 0
    assert "AOB_BENCH_FAIL_RANK" in out.stderr
    assert took < 120.0          # imports + scene generation of the surviving rank; never a collective timeout
