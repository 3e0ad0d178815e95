#!/usr/bin/env python3
"""bench.py — AO rays/s of the bake hot path on MI355X (BASELINE.json metric).

A step is one full device-side bake of the workload with the scene already resident in HBM:
BVH build -> UV rasterise + compact -> fused trace -> (N>1: gather of texel strips to rank 0)
-> resolve + bleed.  value = rays all ranks cast / max-over-ranks step time.

Workloads (all: COSINE sampler, any-hit scoring, float accumulation, 256 spp, seed 3):
  N = 1   BASELINE configs[2]: synthetic 999 698-triangle terrain, 2048^2 atlas.
  N > 1   BASELINE configs[3]: the same terrain, ONE fixed 4096^2 atlas cut into N strips of texel rows, one per rank
          (strong scaling: total work is fixed).  The strips start equal and are re-cut during the warm-up steps in
          proportion to the ranks' measured times (the border strips of the terrain are cheaper per row).
          `--scaling weak` keeps round 1's N x 2048^2 atlas instead.

Launch: `python bench.py --gpus N` starts its own N rank processes (fresh children, created before this process
touches the GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the launcher's ranks are
used.  One process per GPU, RCCL (`nccl`) for the gather, one JSON line from rank 0.

The JSON line also carries
  roofline      the trace kernel against its roofs, every fraction physically bounded: `frac` = the bytes its lanes REQUEST from the CU's
                vector L1 (node, triangle and surfel-record fetches, counted live by the oracle walking the product's own schedule — tile
                cuts included — on a sample of tiles) / the kernel's time (HIP events on the library's stream) / (64 B per clock and CU): the
                binding roof, as the slope builds say (`roofs`); beside it the L1 -> L2 request rate against L2 peak, the fabric-side bytes
                (FETCH_SIZE: Infinity-Cache hits included) against HBM peak, the compulsory footprint and how many times over it was fetched
                — those from the committed rocprofv3 PMC passes of THIS kernel source and configuration (profiles/k_trace_counters.json),
                null when either differs.  SURVEY.md §8(d)'s contract figure — 64 B x the oracle's node visits FROM THE ROOT + 40 / spp per
                ray against HBM peak — is kept as `algorithmic_frac`: a cache-oblivious demand figure that may exceed 1.
  cpu_baseline  the CPU oracle (a port, oracle/) timed on this box's host cores on a bounded texel subsample (N = 1 only).
  bake_wallclock_ms  the PCIe-inclusive `aob_bake` (allocation + upload + device step + download), first call of the
                process (cold) and second (warm), beside the device step (N = 1 only) — SURVEY.md §8(d)(ii).
  cpp_host      the same workload through the C++ multi-GPU host (one process, one thread + one RCCL rank per device,
                aobaker_amd/host/MultiGpu.cpp), run by rank 0 as a child process after the timed region.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# CRC-32 of the resolved float atlases of BASELINE configs[2] (999 698-triangle terrain, seed 3, 2048^2, 256 spp, cosine / any hit / float) and
# configs[3] (the same terrain at 4096^2: what `--gpus N`, N > 1, bakes in N strips).  ORACLE-DERIVED since round 6: tests/golden/make_fullsize_golden.py
# bakes both atlases completely with the CPU oracle (every texel: 1.07 G / 4.29 G rays) and writes tests/golden/fullsize_crc.json; rounds 4 and 5
# carried constants the GPU path had produced (they turned out to be these very numbers).  What every builder, schedule, accelerator and host
# must print: the line carries them as config.image_crc32_expected, tests/test_gpu_fullsize.py and tests/test_gpu_configs.py assert them.
def _golden_crc(config, fallback):
    try:
        with open(os.path.join(ROOT, "tests", "golden", "fullsize_crc.json")) as f:
            return int(json.load(f)[config]["image_crc32"])
    except Exception:
        return fallback


CONFIG2_IMAGE_CRC32 = _golden_crc("2", 259823415)
CONFIG3_IMAGE_CRC32 = _golden_crc("3", 3632858360)
# what k_trace's time depends on: the kernel, and the builder whose tree it walks
KERNEL_SOURCES = ("aobaker_amd/csrc/trace.hip", "aobaker_amd/csrc/aob_math.h", "aobaker_amd/csrc/aob_internal.h", "aobaker_amd/csrc/bvh_build.hip",
                  "aobaker_amd/csrc/horizon_kernel.h", "aobaker_amd/csrc/horizon.hip")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # (the default line is the driver's: 20 timed steps after 5 — a second of GPU time; with 3 after 1 the clocks had not settled: 43.0 ms a step against 40.9)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--atlas", type=int, default=0, help="0 = the BASELINE config of --gpus (2048 at N=1, 4096 at N>1)")
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--grid", type=int, default=708, help="terrain grid (708 -> 999 698 triangles)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--bvh-width", type=int, default=0, help="0 = library default (4), 2 = BVH2 nodes, 4 = 4-ary nodes")
    ap.add_argument("--bvh-quantised", type=int, default=-1, help="-1 = library default (1); 1 = 15-bit quantised boxes, lane per ray; 0 = fp32 boxes")
    ap.add_argument("--bvh-builder", type=int, default=-1,
                    help="-1 = the library's default: by workload (aob_resolve_bvh_builder: PLOC under a binned-SAH top for this one); "
                         "2 = PLOC clusters under a binned-SAH top, 1 = PLOC, 0 = LBVH")
    ap.add_argument("--no-balance", action="store_true", help="N>1: keep the equal strips (no re-cut during warm-up)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpp-host", action="store_true")
    ap.add_argument("--no-wallclock", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="gloo: REHEARSAL of the N-rank run where RCCL cannot form (ranks sharing one GPU): strips travel through host memory")
    ap.add_argument("--oversubscribe", action="store_true", help="rehearsal: rank r uses device r %% (visible devices)")
    ap.add_argument("--launch-timeout", type=float, default=540.0, help="self-launched ranks: seconds before the parent ends all of them and reports an error")
    ap.add_argument("--collective-timeout", type=float, default=120.0, help="torch.distributed timeout (s): a dead peer is an error, not a wait")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# parent of a self-launched N-rank run: never imports torch, never touches the GPU

def visible_devices():
    """GPUs a fresh process sees (counted in a child: counting must not initialise the GPU in THIS process)."""
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def supervise(commands, deadline_s, poll_s=0.2, grace_s=5.0):
    """Run one child per (argv, env) pair and watch ALL of them.  As soon as any exits non-zero, or `deadline_s` passes, the
    others are terminated (SIGTERM, SIGKILL after `grace_s`): a rank that died must not leave its peers waiting in a
    collective until a store / RCCL timeout.  Returns (exit codes, stdout of child 0, stderr tail per child, reason or None)."""
    import tempfile
    outs, procs = [], []
    for argv, env in commands:
        so, se = tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")
        outs.append((so, se))
        procs.append(subprocess.Popen(argv, env=env, stdout=so, stderr=se, text=True))
    t0 = time.monotonic()
    reason = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            # (a moment for the peers that die of the same cause — a closed connection — so that the report names them all)
            time.sleep(min(0.5, poll_s * 2))
            codes = [p.poll() for p in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            reason = "; ".join(f"rank {r} exited with code {codes[r]}" for r in bad)
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() - t0 > deadline_s:
            reason = f"ranks still running after {deadline_s:.0f} s"
            break
        time.sleep(poll_s)
    if reason:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < grace_s:
            time.sleep(0.05)
        for p in procs:
            if p.poll() is None:
                p.kill()
    codes = [p.wait() for p in procs]
    texts = []
    for so, se in outs:
        so.seek(0); se.seek(0)
        texts.append((so.read(), se.read()))
        so.close(); se.close()
    return codes, texts[0][0], [t[1][-2000:] for t in texts], reason


def launch_ranks(args):
    import socket
    have = visible_devices()
    if have < (1 if args.oversubscribe else args.gpus):
        msg = f"bench.py --gpus {args.gpus} needs {args.gpus} devices, {have} visible on this box: nothing was run"
        print(msg, file=sys.stderr)
        print(json.dumps({"metric": "AO rays/sec", "value": None, "unit": "rays/s", "n_gpus": args.gpus, "skipped": msg}), flush=True)
        return 0
    for attempt in range(3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        commands = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            commands.append(([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env))
        codes, out0, errs, reason = supervise(commands, args.launch_timeout)
        # the port was free when probed; another process may have taken it since: once more on another
        if reason and attempt < 2 and any("address already in use" in e.lower() or "eaddrinuse" in e.lower() for e in errs):
            print(f"bench.py: port {port} was taken, launching the ranks again", file=sys.stderr)
            continue
        break
    line = None
    for l in (out0 or "").splitlines():
        if l.startswith("{"):
            line = l
        else:
            print(l, file=sys.stderr)
    for r, e in enumerate(errs):
        if e.strip() and (reason or r == 0):
            print(f"---- rank {r} stderr (tail) ----\n{e}", file=sys.stderr)
    if reason or any(codes) or not line:
        why = reason or ("rank 0 printed no result line" if not line else f"rank exit codes {codes}")
        print(f"bench.py: {why}; rank exit codes {codes}", file=sys.stderr)
        print(json.dumps({"metric": "AO rays/sec", "value": None, "unit": "rays/s", "n_gpus": args.gpus, "error": why, "rank_exit_codes": codes}), flush=True)
        return 1
    print(line, flush=True)
    return 0


# ---------------------------------------------------------------------------------------------------------------

def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch                       # first: one HIP runtime for torch and the library
    import torch.distributed as dist
    import numpy as np
    from aobaker_amd import BakeEngine, Baker, capi, scenes
    from aobaker_amd import dist as aod
    from aobaker_amd.scene import FlatScene

    if args.oversubscribe:
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    host_group = None
    rehearsal = args.backend == "gloo"
    comm_device = "cpu" if rehearsal else "cuda"            # where the small control tensors of the collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        import datetime
        limit = datetime.timedelta(seconds=args.collective_timeout)
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=limit)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=limit)
        host_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=max(args.collective_timeout, 300.0)))   # host-side waits that keep the GPUs idle (cpp_host leg)
        # fault injection for the launcher's tests (tests/test_gpu_dist.py): this rank dies before the first gather
        if os.environ.get("AOB_BENCH_FAIL_RANK") == str(rank):
            raise SystemExit(f"rank {rank}: AOB_BENCH_FAIL_RANK")

    strong = args.scaling == "strong"
    if world == 1:
        W = H = args.atlas or 2048
    elif strong:
        W = H = args.atlas or 4096
    else:
        W, H = aod.weak_scaling_atlas(args.atlas or 2048, world)
    scene = scenes.terrain_scene(grid=args.grid, atlas=(W, H), seed=3, bpc=4)
    flat = FlatScene(scene)
    BakeEngine.Init(local_rank)
    opts = capi.default_options()
    opts.sample_count = args.spp
    opts.sampler = capi.SAMPLER_COSINE
    opts.seed = 3
    opts.use_any_hit = 1
    opts.accumulate = capi.ACCUMULATE_FLOAT
    opts.normal_fp16 = capi.FP16_OFF
    if args.bvh_width:
        opts.bvh_width = args.bvh_width
    if args.bvh_quantised >= 0:
        opts.bvh_quantised = args.bvh_quantised
    if args.bvh_builder >= 0:
        opts.bvh_builder = args.bvh_builder
    builder_by = "asked for" if opts.bvh_builder >= 0 else "the library's default for this workload"
    # (what the library settles on for this scene — aob_bake_create does the same — written out so that the line, the C++ host leg and the
    # oracle's visit count all name the builder that ran)
    import ctypes
    opts.bvh_builder = capi.load().aob_resolve_bvh_builder(ctypes.byref(opts), W * H, flat.triangle_count())
    args.bvh_builder_resolved = int(opts.bvh_builder)

    wallclock = None
    if world == 1 and not args.no_wallclock:
        # what corresponds to the reference's Bake() — allocation, upload (28 MB), device step, download
        # (W*H*4 B) — through the one-shot entry point; first call of the process (code objects load, first launches) and again
        times = []
        for _ in range(2):
            t = time.perf_counter()
            BakeEngine.Options = opts
            BakeEngine.Bake(scene)
            times.append((time.perf_counter() - t) * 1e3)
        st = BakeEngine.LastStats
        wallclock = {"aob_bake_cold_ms": round(times[0], 2), "aob_bake_warm_ms": round(times[1], 2),
                     "of_which_ms": {"upload": round(st.ms_upload, 3), "bvh_build": round(st.ms_bvh_build, 3), "rasterize": round(st.ms_rasterize, 3),
                                     "trace": round(st.ms_trace, 3), "resolve": round(st.ms_resolve, 3), "download": round(st.ms_download, 3)},
                     "note": "PCIe-inclusive one-shot aob_bake (create + upload + device step + download + destroy); never `value`"}

    baker = Baker(flat, opts)           # uploads the scene: inputs resident in HBM from here on
    cells = aod.alias_results(baker, 0, as_float=True) if world > 1 else None
    trace_ms, step_ms, gather_ms, build_raster_ms, stage_hist = [], [], [], [], []
    counts = {}
    bounds = aod.equal_rows(H, world)   # every rank rasterises and traces its own strip of texel rows; the BVH is built by all

    def step():
        t = time.perf_counter()
        rows = (bounds[rank], bounds[rank + 1])
        baker.build_bvh()
        baker.rasterize(0, rows if world > 1 else None)
        build_raster_ms.append((time.perf_counter() - t) * 1e3)     # (the rasteriser reads its counts back: the host has waited for the build and for it)
        n = baker.surfel_count(0)
        baker.trace(0, 0, n)
        if world > 1:
            baker.sync()
            mine = (time.perf_counter() - t) * 1e3
            tg = time.perf_counter()
            all_cells = [(bounds[r] * W, bounds[r + 1] * W - 1) for r in range(world)]
            if rehearsal:
                # through host memory: a copy of the whole cell array, the same gather on it, rank 0's strips copied back
                host_cells = cells.cpu()
                aod.gather_strips(host_cells, all_cells[rank], rank, world, 0, False, all_cells)
                if rank == 0:
                    for c0, c1 in all_cells[1:]:
                        if c0 <= c1:
                            cells[c0:c1 + 1].copy_(host_cells[c0:c1 + 1])
            else:
                aod.gather_strips(cells, all_cells[rank], rank, world, 0, False, all_cells)
            torch.cuda.synchronize()
            gather_ms.append((time.perf_counter() - tg) * 1e3)      # (rank 0: until the last strip has arrived; a peer: until its send has left)
        if rank == 0:
            baker.resolve(0)
        baker.sync()
        if world == 1:
            mine = (time.perf_counter() - t) * 1e3
        trace_ms.append(baker.last_trace_ms())
        stage_hist.append(baker.last_stage_ms())           # (HIP events on the library's stream: build, rasterisation, and the trace's tile cuts / horizon tables / walk)
        step_ms.append(mine)
        counts["mine"] = n

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        baker.sync()

    history = []
    for _ in range(args.warmup):
        step()
        if world > 1 and not args.no_balance:
            # re-cut the strips in proportion to what every rank's strip just cost (host clock: build + raster + trace)
            t = torch.tensor([step_ms[-1]], dtype=torch.float64, device=comm_device)
            every = [torch.zeros(1, dtype=torch.float64, device=comm_device) for _ in range(world)]
            dist.all_gather(every, t)
            ms = [float(v.item()) for v in every]
            history.append({"rows": list(bounds), "rank_ms": [round(v, 3) for v in ms]})
            bounds = aod.rebalance_rows(bounds, ms, H)
    trace_ms.clear()
    stage_hist.clear()
    step_ms.clear()
    gather_ms.clear()
    build_raster_ms.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stats = baker.stats()

    def stage_mean(name):
        """mean over the timed steps of one stage's HIP-event time (the same clock for every stage: the library's stream)"""
        v = [h[name] for h in stage_hist if h.get(name) is not None]
        return sum(v) / len(v) if v else 0.0
    # every buffer one trace launch reads, once (the compulsory footprint the fabric-side traffic is read against)
    footprint = None
    try:
        import ctypes as C
        n4q, tiles_c, recs_c = C.c_uint64(), C.c_uint32(), C.c_uint32()
        capi.check(baker.lib.aob_bake_node4q_count(baker.handle, C.byref(n4q)))
        baker.lib.aob_bake_copy_tile_cuts(baker.handle, 0, None, 0, C.byref(tiles_c), C.byref(recs_c))
        parts = {"nodes4q": int(n4q.value) * 64, "leaf_tris": int(stats.triangle_count) * 48, "frames": int(counts["mine"]) * 64,
                 "tile_cuts": int(tiles_c.value) * int(recs_c.value) * 512, "result_cells": int(counts["mine"]) * 8,
                 "horizon_tables": int(counts["mine"]) * (16 if args.spp <= 256 else 32) if int(stats.trace_horizon) == 1 else 0}
        footprint = {"bytes": sum(parts.values()), "of_which": parts}
    except Exception as e:
        print(f"bench: footprint failed: {e!r}", file=sys.stderr)
    image_crc = None
    if rank == 0:
        import zlib
        m = flat.material_array[0]
        img = np.zeros(m.width * m.height * m.channels * m.bytes_per_channel, dtype=np.uint8)
        baker.download(0, img)                              # after the timed region: the resolved atlas of the last step
        image_crc = zlib.crc32(img.tobytes())
    my_rays = counts["mine"] * args.spp
    rank_ms = [sum(step_ms) / len(step_ms)]
    if world > 1:
        t = torch.tensor([counts["mine"]], dtype=torch.int64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        counts["surfels"] = int(t.item())
        t = torch.tensor(rank_ms, dtype=torch.float64, device=comm_device)
        every = [torch.zeros(1, dtype=torch.float64, device=comm_device) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [float(v.item()) for v in every]
        # every rank's own stages of the last step (each builds the whole tree — redundantly — and rasterises and traces its strip) and what
        # it spent in the gather: the redundant build's share of a rank's step is visible here, not inferred
        mine4 = torch.tensor([stage_mean("bvh_build"), stage_mean("rasterize"), sum(trace_ms) / len(trace_ms), sum(gather_ms) / max(1, len(gather_ms))],
                             dtype=torch.float64, device=comm_device)
        every4 = [torch.zeros(4, dtype=torch.float64, device=comm_device) for _ in range(world)]
        dist.all_gather(every4, mine4)
        rank_stages = [[float(x) for x in v.tolist()] for v in every4]
    else:
        counts["surfels"] = counts["mine"]
        rank_stages = [[stage_mean("bvh_build"), stage_mean("rasterize"), sum(trace_ms) / len(trace_ms), 0.0]]
    total_rays = counts["surfels"] * args.spp                  # all ranks together
    ms_per_step = elapsed / args.steps * 1e3
    value = total_rays / (elapsed / args.steps)
    trace_total_ms = sum(trace_ms) / len(trace_ms)
    # the dominant kernel alone: k_trace (the walk); the tile cuts and the horizon tables — kernels of their own inside the trace stage — beside it
    kernel_ms = stage_mean("walk") or trace_total_ms
    horizon_ms, cuts_ms = stage_mean("horizon"), stage_mean("tile_cuts")
    rays_cast = int(stats.rays_cast) if stats.trace_horizon == 1 else None
    hz_tables = None
    if rank == 0 and world == 1 and rays_cast is not None:
        try:
            hz_tables = baker.horizon(0)
        except Exception as e:
            print(f"bench: horizon tables not read back: {e!r}", file=sys.stderr)
    baker.close()
    BakeEngine.Dispose()

    # the same workload through the C++ multi-GPU host (rank 0 starts it as a child once every rank has let go of its GPU work)
    cpp = None
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    if rank == 0 and not args.no_cpp_host:
        # (a rehearsal's ranks share the visible devices: the C++ host is told the same, rank r on device r % visible)
        cpp = cpp_host_leg(scene, world, args, [r % max(1, torch.cuda.device_count()) for r in range(world)] if (world > 1 and rehearsal) else None)
    if world > 1:
        dist.barrier(group=host_group)      # gloo: the peers wait on the host, their GPUs stay idle meanwhile

    if rank == 0:
        cpu = None
        visits = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, visits = cpu_baseline(flat, opts, W, H, args.cpu_seconds)
        if visits is None:
            visits = visits_per_ray(flat, opts, W, H)
        requested = None
        try:
            requested = requested_per_ray(flat, opts, W, H, tables=hz_tables)
        except Exception as e:                       # (a report, never a reason to lose the bench line)
            requested = None
            print(f"bench: requested_per_ray failed: {e!r}", file=sys.stderr)
        roofline = roofline_record(opts, args, W, H, world, visits, my_rays, kernel_ms, requested, footprint, rays_cast)
        line = {
            "metric": "AO rays/sec", "value": value, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"terrain {flat.triangle_count()} tris, {W}x{H} atlas, {args.spp} spp, "
                                   f"cosine sampler, any-hit, float accumulate, {('LBVH', 'PLOC', 'PLOC + binned-SAH top (bvh_builder 2)')[opts.bvh_builder]} ({builder_by}) bvh_width {opts.bvh_width or 2}"
                                   f"{' quantised' if opts.bvh_quantised else ''} — "
                                   + ("BASELINE configs[2]" if world == 1 else
                                      ("BASELINE configs[3]: one fixed atlas in N strips of texel rows" if strong else "configs[2] scaled: N x 2048^2 texels")),
                       "mesh": f"synthetic fBm heightfield, {args.grid}x{args.grid} vertices, seed 3 (aobaker_amd/scenes.py terrain_scene): "
                               "the BASELINE terrain itself, not a stand-in",
                       "triangles": flat.triangle_count(), "atlas": [W, H], "spp": args.spp,
                       "surfels": counts["surfels"], "rays": total_rays, "sharding": f"strips of texel rows x{world}",
                       "rows": [[bounds[r], bounds[r + 1]] for r in range(world)],
                       "bvh_nodes": stats.node_count, "bvh_depth": stats.bvh_depth,
                       # horizon tables (AobOptions::trace_horizon, horizon.hip): of this rank's rays, how many the trace kernel cast — the rest sit in cells of the
                       # sampler's grid that no triangle can reach (proven per surfel from the tree, inside the timed step) and score as unoccluded without a walk;
                       # `value` counts every ray the bake resolved, as the metric defines it (covered surfels x spp), the image is the oracle's
                       "horizon_tables": ({"rays_cast": rays_cast, "rays_cast_share": round(rays_cast / max(1, my_rays), 4), "rays_resolved_by_tables": my_rays - rays_cast,
                                           "table_kernel_ms": round(horizon_ms, 3), "cast_rays_per_s": round(rays_cast / (kernel_ms * 1e-3), 1)} if rays_cast is not None else None),
                       "image_crc32": image_crc,
                       "image_crc32_expected": (CONFIG2_IMAGE_CRC32 if (W, H, args.spp, args.grid) == (2048, 2048, 256, 708) else
                                                CONFIG3_IMAGE_CRC32 if (W, H, args.spp, args.grid) == (4096, 4096, 256, 708) else None),
                       "image_crc32_ok": None if image_crc is None or (W, H, args.spp, args.grid) not in ((2048, 2048, 256, 708), (4096, 4096, 256, 708)) else
                                         bool(image_crc == (CONFIG2_IMAGE_CRC32 if W == 2048 else CONFIG3_IMAGE_CRC32)),
                       "host": ("REHEARSAL: ranks share devices, strips gathered with gloo through host memory" if rehearsal else
                                "python ranks, one process per GPU, torch.distributed nccl (= RCCL) point-to-point gather to rank 0")},
            "stage_ms": {"bvh_build": round(stage_mean("bvh_build"), 3), "rasterize": round(stage_mean("rasterize"), 3), "trace": round(trace_total_ms, 3),
                         "of_which_trace": {"tile_cuts": round(cuts_ms, 3), "horizon_tables": round(horizon_ms, 3), "k_trace": round(kernel_ms, 3)},
                         "trace_kernel": round(kernel_ms, 3),
                         "rank_build_raster_trace": [round(v, 3) for v in rank_ms],
                         # per rank, means over the timed steps of the library's own HIP events: the tree's build, the rasterisation of its strip, its trace stage
                         # (tile cuts + horizon tables + k_trace), and the host time it spent in the gather (N > 1)
                         "rank_bvh_build": [round(v[0], 3) for v in rank_stages], "rank_rasterize": [round(v[1], 3) for v in rank_stages],
                         "rank_trace_kernel": [round(v[2], 3) for v in rank_stages],
                         "gather_ms": round(rank_stages[0][3], 3) if world > 1 else None,
                         "rank_gather_ms": [round(v[3], 3) for v in rank_stages] if world > 1 else None},
            "bake_wallclock_ms": dict(wallclock or {}, device_step_ms=round(ms_per_step, 3)),
            "roofline": roofline, "cpu_baseline": cpu, "cpp_host": cpp,
        }
        if world > 1:
            line["balance"] = {"enabled": not args.no_balance, "warmup_history": history}
            if strong:
                base = strong_scaling_base(W, args.spp)
                line["strong_scaling_base"] = base
                # what one GPU needed for each rank's strip of this atlas cut N ways, before any re-cut (tools/run_shapes.py ranks, the
                # same kernel source): the first real N-GPU run's rank_build_raster_trace is to be read against these
                line["expected_rank_ms"] = ((base or {}).get("rank_strip_ms") or {}).get(str(world))
        if world > 1:
            # with more than one GPU the headline is the C++ host's — ONE process, a host thread and an RCCL rank per device
            # (aobaker_amd/host/MultiGpu.cpp: SURVEY 8(e)'s process model, the north star's "the repo's own C++") — timed by itself over the same K
            # steps (barrier on all threads + stream drains on both sides, the slowest rank's time); the python ranks' figures stay beside it.
            # Only a C++ run that printed the expected image counts; otherwise the python ranks' value stands and the line says why.
            py = {"value": value, "ms_per_step": round(ms_per_step, 3), "host": line["config"]["host"]}
            expected = line["config"]["image_crc32_expected"]
            crcs = (cpp or {}).get("image_crc32") or []
            cpp_ok = bool(cpp) and "error" not in cpp and cpp.get("value") and (expected is None or (len(crcs) > 0 and all(c == expected for c in crcs)))
            if cpp_ok:
                line["value"] = cpp["value"]; line["ms_per_step"] = cpp["ms_per_step"]; line["steps"] = cpp.get("steps", args.steps); line["warmup"] = cpp.get("warmup", args.warmup)
                line["config"]["host"] = "C++: aobake_mgpu, one process, one host thread + one RCCL rank per device (ncclCommInitAll), strips gathered to rank 0 with grouped ncclSend / ncclRecv"
                line["config"]["image_crc32"] = crcs[0]; line["config"]["image_crc32_ok"] = None if expected is None else True
                line["stage_ms"]["rank_build_raster_trace"] = cpp.get("rank_ms"); line["stage_ms"]["rank_bvh_build"] = cpp.get("rank_bvh_build_ms")
                line["stage_ms"]["rank_trace_kernel"] = cpp.get("rank_trace_kernel_ms"); line["stage_ms"]["gather_ms"] = cpp.get("gather_ms"); line["stage_ms"]["rank_gather_ms"] = cpp.get("rank_gather_ms")
                line["value_from"] = "cpp_host"
            else:
                line["value_from"] = "python_ranks: " + ("the C++ host leg was skipped" if cpp is None else str((cpp or {}).get("error") or "the C++ host's image CRC is not the expected one"))
            line["python_ranks"] = py
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpp_host_leg(scene, world, args, devices=None):
    """`aobake_mgpu` on the same scene: one process, one host thread + one RCCL rank per device (MultiGpu.cpp)."""
    import tempfile
    from aobaker_amd import scenefile
    exe = os.path.join(ROOT, "aobaker_amd", "host", "aobake_mgpu")
    if not os.path.exists(exe):
        return {"error": f"{exe} has not been built"}
    try:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "scene.aobs")
            scenefile.write_scene(scene, path)
            # (N > 1: this run carries the line's `value` — exactly the K steps after W warm-ups the contract names; N = 1: a side-leg, bounded)
            k_steps = args.steps if world > 1 else max(1, min(args.steps, 5))
            k_warm = max(2, args.warmup) if world > 1 else max(2, min(args.warmup, 3))
            cmd = [exe, "--scene", path, "--gpus", str(world), "--steps", str(k_steps), "--warmup", str(k_warm),
                   "--spp", str(args.spp), "--sampler", "1", "--accumulate", "1", "--seed", "3", "--any-hit", "1", "--normal-fp16", "0", "--builder", str(args.bvh_builder_resolved)]
            if args.no_balance:
                cmd.append("--no-balance")
            if devices:
                cmd += ["--devices", ",".join(str(d) for d in devices)]
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=180 if world == 1 else 420)
        if out.returncode != 0:
            return {"error": f"rc {out.returncode}: {out.stderr.strip()[-300:]}"}
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:                       # the leg is a report, never a reason to lose the bench line
        return {"error": repr(e)[:300]}


def strong_scaling_base(atlas, spp, path=None):
    """The single-GPU time of the SAME fixed atlas the N > 1 value is to be read against (measured, committed) — if it was
    measured on THIS kernel source; a base from another kernel is refused, with the reason."""
    p = path or os.path.join(ROOT, "profiles", "strong_scaling_base.json")
    try:
        with open(p) as f:
            rec = json.load(f)
    except Exception:
        return None
    if rec.get("atlas") != atlas or rec.get("spp") != spp:
        return None
    if rec.get("kernel_source_sha256_16") != kernel_source_hash():
        return {"refused": f"profiles/strong_scaling_base.json was measured on kernel source {rec.get('kernel_source_sha256_16')}, this is {kernel_source_hash()}"}
    return rec


def kernel_source_hash():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def counter_key(opts, args, W, H, world):
    """What a counter record must have been measured on to describe this run: the line carries it (roofline.counter_key) and
    tools/pmc_record.py copies it from the line of the PROFILED run into the record — never from literals."""
    return {"atlas": [W, H], "spp": args.spp, "grid": args.grid, "n_gpus": world, "bvh_width": opts.bvh_width, "bvh_quantised": opts.bvh_quantised,
            "bvh_builder": opts.bvh_builder, "trace_cut": opts.trace_cut, "trace_horizon": opts.trace_horizon}


def counter_record(opts, args, W, H, world, path=None):
    """profiles/k_trace_counters.json if it was measured on this kernel source and this configuration, else (None, why)."""
    p = path or os.path.join(ROOT, "profiles", "k_trace_counters.json")
    try:
        with open(p) as f:
            rec = json.load(f)
    except Exception:
        return None, "no profiles/k_trace_counters.json"
    want = counter_key(opts, args, W, H, world)
    if rec.get("source_sha256_16") != kernel_source_hash():
        return None, f"counters were measured on kernel source {rec.get('source_sha256_16')}, this is {kernel_source_hash()}"
    if rec.get("config") != want:
        return None, "counters were measured on another configuration"
    return rec, None


L2_PEAK_GBS = 34500.0                    # MI355X_MICROARCH.md "L2 (per XCD)": 34.5 TB/s aggregate


def requested_per_ray(flat, opts, W, H, tiles=96, tables=None):
    """What ONE ray asks the vector-memory path for, counted by the oracle walking the product's schedule on `tiles` random tiles of the trace
    order: every ray of a tile tests its tile's cut (orc_tile_cut_reach: k_tile_cuts' rule) and walks below the entries it entered
    (orc_cut_walk).  Returns node fetches, triangle fetches, entries tested / entered per ray — the counts the kernel's own loop counters
    give (tools/time_trace.py on the -DAOB_TRACE_STATS build; profiles/): the oracle's are live, on this very run's scene and options.
    `tables`: the horizon tables the run's trace used (Baker.horizon: free columns per surfel and row, surfel order) — then only the rays the
    kernel CAST are walked, and the figures are per cast ray."""
    import numpy as np
    native_oracle()
    from oracle import oracle as orc
    tris = orc.scene_triangles(flat.desc)
    nodes2, _ = orc.bvh_build(tris, opts.bvh_builder, opts.morton_adaptive)
    n4, lt = orc.bvh4_build(nodes2, pairs=True)
    _, frame = orc.bvh2q_build(nodes2)
    n4q = orc.bvh4q_build(n4, frame)
    surfels, _ = orc.rasterize(flat.desc, 0, opts)
    x, y = surfels["texel_x"].astype(np.int64), surfels["texel_y"].astype(np.int64)
    key = ((y // 8) * ((W + 7) // 8) + x // 8) * 64 + (y % 8) * 8 + x % 8
    order = np.argsort(key, kind="stable")
    ordered = surfels[order]
    sx = int(np.floor(np.sqrt(opts.sample_count))); sy = opts.sample_count // max(1, sx)
    free_all = None
    if tables is not None and len(tables) == len(surfels) and sx * sy == opts.sample_count:
        free_all = np.minimum(tables[order][:, :sy].astype(np.int64), sx)
    nt = len(ordered) // 64
    picks = np.random.default_rng(7).choice(nt, size=min(tiles, nt), replace=False)
    reach = float((np.float32(opts.ray_tmax) + np.float32(abs(opts.ray_bias))) * np.float32(1.01))
    rays_n = inner = leaf = tested = entered = 0
    for t in picks:
        sub = np.ascontiguousarray(ordered[64 * t:64 * t + 64])
        rays = np.ascontiguousarray(orc.generate_rays(sub, None, opts, W))
        if free_all is not None:
            # sample i = (cx, cy) = (i % sx, i // sx) of a surfel is cast iff cx >= free[cy]
            f = free_all[64 * t:64 * t + 64]
            cast = (np.arange(sx)[None, None, :] >= f[:, :, None]).reshape(len(sub), sx * sy)
            rays = np.ascontiguousarray(rays.reshape(len(sub), -1)[:, :sx * sy][cast])
            if len(rays) == 0:
                continue
        pos = sub["position"].astype(np.float32)
        pmin, pmax = pos.min(axis=0), pos.max(axis=0)
        pull = np.float32(0.25) * (pmax - pmin)
        cut = orc.tile_cut_reach(n4q, frame, pmin + pull, pmax - pull, pmin, pmax, reach, 32)
        st, _ = orc.cut_walk(n4q, frame, lt, rays, cut)
        rays_n += st.rays; inner += st.inner_visits; leaf += st.leaf_visits; tested += st.entries_tested; entered += st.mask_bits
    r = float(max(1, rays_n))
    return {"node_fetches": inner / r, "triangle_fetches": leaf / r, "cut_entries_tested": tested / r, "cut_entries_entered": entered / r,
            "per": "cast ray (the rays the horizon tables left to the walk)" if free_all is not None else "ray",
            "counted_on": f"{len(picks)} random tiles of 64 surfels ({rays_n} rays), the oracle walking the tiles' cuts (32 entries; the kernel tests fewer: its "
                          "records are culled by sample part, which changes no walk)"}


def roofline_record(opts, args, W, H, world, visits, my_rays, kernel_ms, requested=None, footprint=None, rays_cast=None):
    """The trace kernel against its roofs, every fraction physically bounded (<= 1):
      frac / achieved / peak   the BINDING roof: the CU's vector-memory (L1) request path — bytes the kernel's lanes ask it for (node, triangle,
                               surfel-record and frame fetches, counted live by the oracle walking the product's schedule) / kernel time /
                               (64 B per clock and CU).  Why this one binds: the slope builds (roofs.vmem / roofs.valu below).
      l2 / fabric              the same for the L1 -> L2 requests and for what leaves the L2s (FETCH_SIZE: Infinity-Cache hits INCLUDED,
                               MI355X_MICROARCH.md "HBM": this is fabric-side traffic, an upper bound on HBM's) — from the committed PMC passes
                               of THIS kernel source and configuration, null when either differs.
      algorithmic_frac         SURVEY §8(d)'s contract figure: 64 B x the node visits of the oracle's walk FROM THE ROOT + 40 / spp per ray,
                               against HBM peak.  A cache-oblivious DEMAND figure — it can exceed 1 (the kernel starts below tile cuts and
                               fetches 6 nodes where that walk visits 16, and L1 / L2 / MALL serve most fetches); it is not a distance from a roof.
      compulsory_bytes / refetch_ratio   every buffer the launch reads, once, and how many times over the fabric side moved it."""
    n_inner, n_leaf = visits
    seconds = kernel_ms * 1e-3
    # SURVEY §8(d) as written: 64 B per node packet fetched (a leaf visit is counted as one packet too) + (32 B surfel + 8 B result cell) / spp
    b_ray = 64.0 * (n_inner + n_leaf) + 40.0 / args.spp
    algorithmic = my_rays * b_ray / seconds / 1e9
    rec, why = counter_record(opts, args, W, H, world)
    # ---- the L1 request path: what the lanes ask for ----
    walked = rays_cast if rays_cast is not None else my_rays          # the rays k_trace walked (horizon tables: the cast ones)
    req = None
    if requested:
        # per walked ray: 64 B per node fetched below the cut, 48 B per triangle fetched, 16 B of the surfel's record when a lane takes the ray off the
        # queue, 48 B of its frame (+ 4 B of its horizon-table word) when the sample is generated, the 8-B cell per surfel and sample part
        b_req = 64.0 * requested["node_fetches"] + 48.0 * requested["triangle_fetches"] + 16.0 + 48.0 + (4.0 if rays_cast is not None else 0.0) + 8.0 / args.spp
        req = {"bytes_per_ray": round(b_req, 1), "per_ray": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in requested.items()},
               "rays_walked": walked, "GBps": round(walked * b_req / seconds / 1e9, 1),
               "counted_by": "the oracle walking the product's schedule on sampled tiles (the rays the run's own horizon tables left); the kernel's own loop counters "
                             "(-DAOB_TRACE_STATS build) are in profiles/: roofs/loop_counters.jsonl of the round's final profile"}
    cus = 256
    l1_peak = 64.0 * cus * 2.4
    l1_frac = round(req["GBps"] / l1_peak, 4) if req else None
    l2 = fabric = None
    roofs = None
    if rec:
        c = rec.get("counters", {})
        if c.get("TCP_TCC_READ_REQ_sum"):
            l2_bytes = c["TCP_TCC_READ_REQ_sum"] * 64.0
            l2 = {"bytes_per_launch": l2_bytes, "GBps": round(l2_bytes / seconds / 1e9, 1), "peak": L2_PEAK_GBS, "frac": round(l2_bytes / seconds / 1e9 / L2_PEAK_GBS, 4),
                  "note": "TCP_TCC_READ_REQ_sum x 64 B (the request size is not calibrated on gfx950: 128-B requests would double it; below the roof either way)",
                  "L2_hit_rate": rec.get("L2_hit_rate")}
        fb = rec.get("fabric_side_bytes_per_launch", rec.get("hbm_side_bytes_per_launch"))
        if fb:
            fabric = {"bytes_per_launch": fb, "GBps": round(fb / seconds / 1e9, 1), "peak": HBM_PEAK_GBS, "frac": round(fb / seconds / 1e9 / HBM_PEAK_GBS, 4),
                      "note": "FETCH_SIZE + WRITE_SIZE, corrected by the known-byte kernel k_gather64 (tools/fetch_calib.hip): requests leaving the L2s — Infinity-Cache "
                              "hits are counted in (MI355X_MICROARCH.md, HBM section), so this bounds HBM traffic from above"}
        roofs = rec.get("roof")
    refetch = None
    if footprint and fabric:
        refetch = round(fabric["bytes_per_launch"] / footprint["bytes"], 1)
    fracs = {"l1_request_path": l1_frac, "l2": (l2 or {}).get("frac"), "fabric": (fabric or {}).get("frac")}
    known = {k: v for k, v in fracs.items() if v is not None}
    largest = max(known, key=known.get) if known else None
    # WHICH unit binds is a measurement of this kernel source (the slope builds and class counters of profiles/k_trace_counters.json): named only when
    # the committed record is of this source and configuration
    bound = (roofs or {}).get("bound")
    inst = ({"bound": roofs.get("bound"), "frac": roofs.get("frac"), "valu_frac": (roofs.get("valu") or {}).get("frac"),
             "vmem_frac": (roofs.get("vmem") or {}).get("frac")} if roofs else None)
    return {"bound": bound,
            "bound_is": ("the unit the slope builds and class counters of THIS kernel source name (profiles/k_trace_counters.json): "
                         + (f"{bound}, {roofs.get('frac')} of its issue slots" if roofs else "no record of this kernel source and configuration is committed: not named")
                         + " — this walk is bound by instruction issue, not by bytes; the byte-rate fractions (L1 request path, L2, fabric / HBM) are beside it"),
            "instruction_roof": inst,
            # the contract's byte figures: `achieved` / `peak` / `frac` are the LARGEST byte-rate fraction's — the CU's vector-memory (L1) request path —
            # and `hbm_frac` is the one the north star asks for: what leaves the L2s against HBM's 8 TB/s
            "achieved": (req or {}).get("GBps"), "peak": round(l1_peak, 1), "unit": "GB/s", "frac": l1_frac,
            "frac_is": f"bytes the kernel's lanes request from the vector L1 / k_trace's time / (64 B per clock and CU x {cus} CUs x 2.4 GHz, the rate tools/l1_probe.hip measures)",
            "hbm_frac": (fabric or {}).get("frac"),
            "hbm_frac_is": "fabric-side bytes per launch (FETCH_SIZE + WRITE_SIZE, calibrated) / k_trace's time / 8 TB/s: requests leaving the L2s, Infinity-Cache (MALL) hits "
                           "INCLUDED — an upper bound on what HBM served; the guide's gfx950 caveat (FETCH_SIZE may count 64-byte requests as 32) would at most double it; "
                           "null when the committed counters are of another kernel source or configuration",
            "byte_rate_fractions": fracs, "largest_byte_rate_fraction": largest,
            "requested": req, "l2": l2, "fabric": fabric,
            "traffic": (fabric or {}).get("bytes_per_launch"),
            "traffic_is": "fabric-side bytes per launch (see fabric.note); null when the committed counters are of another kernel source or configuration",
            "compulsory": footprint, "refetch_ratio": refetch,
            "refetch_ratio_is": "fabric-side bytes / the bytes of every buffer the launch reads once: how many times the L2s fetched the working set",
            "algorithmic_GBps": round(algorithmic, 1), "algorithmic_frac": round(algorithmic / HBM_PEAK_GBS, 4),
            "algorithmic_frac_is": "SURVEY §8(d): 64 B x node visits of the oracle's walk from the root + 40 / spp per ray, x EVERY ray of the bake, / k_trace's time / HBM peak — "
                                   "a cache-oblivious demand figure that may exceed 1 (the kernel starts below tile cuts, and with horizon tables does not walk most rays at all); "
                                   "not a roofline fraction",
            "bytes_per_ray_algorithmic": round(b_ray, 1),
            "node_visits_per_ray_from_the_root": round(n_inner + n_leaf, 3), "inner_visits_per_ray_from_the_root": round(n_inner, 3), "leaf_visits_per_ray_from_the_root": round(n_leaf, 3),
            # the two instruction-rate roofs from the committed measurements of THIS kernel source (profiles/k_trace_counters.json): which unit,
            # how full, and what one more instruction of either kind per inner visit costs
            "roofs": roofs,
            "valu_lane_occupancy": (rec or {}).get("valu_lane_occupancy"),
            "valu_wave_instr_per_ray": (rec or {}).get("valu_wave_instr_per_ray"), "vmem_wave_instr_per_ray": (rec or {}).get("vmem_wave_instr_per_ray"),
            "counters": (rec or {}).get("provenance") or why,
            "kernel": "k_trace", "kernel_ms": round(kernel_ms, 3), "kernel_source_sha256_16": kernel_source_hash(),
            "rays_per_launch": my_rays, "rays_walked_per_launch": walked, "counter_key": counter_key(opts, args, W, H, world)}


def host_cores():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (a GPU box's share of
    its host is 16 CPUs although 256 are visible)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


_ORACLE_BUILD = None


def native_oracle():
    """SURVEY §8(d) asks for the CPU baseline compiled `-O3 -march=native`: the shipped liboracle.so is x86-64-v3 (it is built in one container and
    runs on another host), so the bench compiles oracle/aob_oracle.c once more ON THE HOST THAT TIMES IT — same source, same floating-point contract
    (-ffp-contract=off, no fast-math) — into a temporary directory and makes the oracle load that (AOB_ORACLE_LIBRARY).  No compiler, or a failed
    build: the shipped library, and the record says so.  Must run before the oracle is first loaded.  Returns the flags used."""
    global _ORACLE_BUILD
    if _ORACLE_BUILD is not None:
        return _ORACLE_BUILD
    shipped = "gcc -O2 -mavx2 -mfma -ffp-contract=off (oracle/Makefile: x86-64-v3, the shipped library; SURVEY §8(d) names -O3 -march=native)"
    _ORACLE_BUILD = shipped
    if os.environ.get("AOB_ORACLE_LIBRARY") or "oracle.oracle" in sys.modules:
        return _ORACLE_BUILD
    import shutil
    import tempfile
    gcc = shutil.which("gcc")
    if not gcc:
        return _ORACLE_BUILD
    try:
        out = os.path.join(tempfile.mkdtemp(prefix="aob_oracle_native_"), "liboracle_native.so")
        flags = ["-O3", "-march=native", "-std=gnu11", "-fPIC", "-ffp-contract=off", "-fno-math-errno", "-fno-fast-math"]
        r = subprocess.run([gcc] + flags + [os.path.join(ROOT, "oracle", "aob_oracle.c"), "-o", out, "-shared", "-pthread", "-lm"],
                           capture_output=True, text=True, timeout=180)
        if r.returncode == 0 and os.path.exists(out):
            os.environ["AOB_ORACLE_LIBRARY"] = out
            _ORACLE_BUILD = "gcc " + " ".join(flags[:2]) + " -ffp-contract=off -fno-fast-math, built on this host for this run (SURVEY §8(d))"
    except Exception:
        pass
    return _ORACLE_BUILD


def _oracle_inputs(flat, opts):
    native_oracle()
    from oracle import oracle as orc
    tris = orc.scene_triangles(flat.desc)
    nodes, _ = orc.bvh_build(tris, opts.bvh_builder, opts.morton_adaptive)
    surfels, _ = orc.rasterize(flat.desc, 0, opts)
    return orc, nodes, surfels


def _subsample(surfels, one_in):
    """A fixed pseudo-random 1/one_in of the texels.  (Not a regular stride: texels at multiples of 2^k sit on the
    lattice points of the terrain's value noise, where it is locally flat — every 32nd texel in x and y sees 10 % fewer
    node visits per ray than the whole atlas.)"""
    import numpy as np
    x, y = surfels["texel_x"].astype(np.uint32), surfels["texel_y"].astype(np.uint32)
    h = x * np.uint32(0x9E3779B1) ^ y * np.uint32(0x85EBCA77)
    h ^= h >> np.uint32(15)
    h *= np.uint32(0x2C1B3C6D)
    h ^= h >> np.uint32(13)
    return np.ascontiguousarray(surfels[h % np.uint32(one_in) == 0])


def visits_per_ray(flat, opts, W, H):
    """Mean BVH node fetches per ray, counted by the oracle on 1/256 of the texels (1/1024 on atlases above 2048^2)."""
    orc, nodes, surfels = _oracle_inputs(flat, opts)
    sub = _subsample(surfels, 256 if W * H <= 2048 * 2048 else 1024)
    _, st = orc.trace(nodes, sub, None, opts, W, H, threads=host_cores())
    return st.inner_visits / max(1, st.rays), st.leaf_visits / max(1, st.rays)


def cpu_baseline(flat, opts, W, H, seconds):
    """The CPU oracle on the same BVH, sampler and rays, on a bounded pseudo-random subset of the texels."""
    cores = host_cores()
    orc, nodes, surfels = _oracle_inputs(flat, opts)
    probe = _subsample(surfels, 1024)
    t = time.perf_counter()
    _, st = orc.trace(nodes, probe, None, opts, W, H, threads=cores)
    rate = st.rays / max(1e-9, time.perf_counter() - t)
    # the largest sample (1 texel in 4^k) that stays within the time budget (at least ~10 s of work)
    one_in = 1024
    while one_in > 1 and len(surfels) / (one_in // 4) * opts.sample_count / rate <= 2.0 * seconds:
        one_in //= 4
    while True:
        sub = _subsample(surfels, one_in)
        t = time.perf_counter()
        _, st = orc.trace(nodes, sub, None, opts, W, H, threads=cores)
        dt = time.perf_counter() - t
        # the probe underestimates the rate of a many-core host (thread start-up): if the sample came out short of the
        # 10-30 s the contract asks for and four times the work still fits, take the next denser sample
        if one_in > 1 and dt < 0.5 * seconds and 4.0 * dt <= 2.0 * seconds:
            one_in //= 4
            continue
        break
    visits = (st.inner_visits / max(1, st.rays), st.leaf_visits / max(1, st.rays))
    return ({"value": st.rays / dt, "unit": "rays/s", "cores": cores, "kind": "port",
             "compiler": native_oracle(),
             "sample": f"a fixed pseudo-random 1/{one_in} of the texels of the same atlas ({len(sub)} surfels, {st.rays} rays, "
                       f"{dt:.1f} s); trace stage only, same BVH/sampler/rays as the GPU",
             "extrapolated_full_trace_s": len(surfels) * opts.sample_count / (st.rays / dt)}, visits)


if __name__ == "__main__":
    main()
