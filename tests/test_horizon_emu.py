"""The horizon kernel's LOGIC without a GPU: aobaker_amd/csrc/horizon_kernel.h compiled for the CPU (tests/horizon_emu/horizon_emu.cpp: one host thread per lane,
LDS as static storage) builds the tables of whole tiles, and every cell they skip must hold only rays the oracle finds unoccluded — on scenes of
every kind the suite has, several sample counts, both tree builders.  (The same kernel source on the GPU: tests/test_gpu_horizon.py.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import horizon_emu as he          # noqa: E402
from aobaker_amd import scenes    # noqa: E402

CASES = [
    ("cube", lambda: scenes.cube_scene(64), 16, 8, 1),
    ("cube", lambda: scenes.cube_scene(64), 64, 8, 2),
    ("inward cube", lambda: scenes.cube_scene(64, inward=True), 64, 8, 1),
    ("displaced sphere", lambda: scenes.sphere_scene(n=40, atlas=64), 64, 6, 1),
    ("triangle soup", lambda: scenes.soup_scene(atlas=64), 64, 6, 1),
    ("floor and wall", lambda: scenes.floor_wall_scene(atlas=32), 256, 6, 1),
    ("two sections", lambda: scenes.two_section_scene(atlas=32), 64, 6, 1),
    ("small terrain", lambda: scenes.terrain_scene(grid=64, atlas=256, seed=3), 256, 4, 1),
    ("small terrain, SAH top", lambda: scenes.terrain_scene(grid=96, atlas=256, seed=5), 64, 5, 2),
    ("rocks", lambda: scenes.instanced_scene(instances=9, rock_n=24, atlas=256), 64, 5, 1),
]


@pytest.mark.parametrize("name,make,spp,tiles,builder", CASES, ids=[f"{c[0]}-{c[2]}spp" for c in CASES])
def test_no_skipped_cell_holds_an_occluded_ray(name, make, spp, tiles, builder):
    rec = he.check(make(), spp, tiles, seed=3, builder=builder)
    assert rec["skipped_cells_with_an_occluded_ray"] == 0, rec
    assert 0.0 <= rec["skipped_share"] <= 1.0 - rec["occluded_share"] + 1e-9, rec


def test_open_scenes_are_mostly_skipped_and_closed_ones_not_at_all():
    """the tables are worth something where there is sky, and claim nothing inside a closed box"""
    open_cube = he.check(scenes.cube_scene(64), 64, 10, seed=5)
    closed = he.check(scenes.cube_scene(64, inward=True), 64, 10, seed=5)
    assert open_cube["skipped_share"] > 0.5, open_cube
    assert closed["skipped_share"] == 0.0 and closed["occluded_share"] == 1.0, closed
