"""The C++ host side (include/BakeEngine.h + aobaker_amd/host: BakeEngine::Bake, the OBJ importer, the PNG / EXR writers, the CLI) produces
the ORACLE's bytes — orc.bake on the same scene, imported by the Python twin of the importer — and, besides, the same bytes as the Python
host side (both are thin callers of the same C ABI; that alone would compare the HIP path with itself: VERDICT r4, rows f1 / f2)."""
import os
import subprocess

import numpy as np
import pytest

from aobaker_amd import capi, scenes
from aobaker_amd.scene import FlatScene
from aobaker_amd.scenefile import write_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "aobaker_amd", "host", "aobake_cli")


@pytest.mark.parametrize("mode", ["compat", "cosine"])
def test_cpp_bake_engine_matches_the_oracle_and_the_python_host(engine, orc, tmp_path, mode):
    assert os.path.exists(CLI), "build the C++ host side: make -C aobaker_amd/host"
    bpc = 1 if mode == "compat" else 4
    mk = lambda: scenes.terrain_scene(grid=64, atlas=128, bpc=bpc)
    opts = capi.default_options()
    opts.sample_count = 16
    opts.seed = 1
    args = ["--spp", "16", "--seed", "1"]
    if mode == "cosine":
        opts.sampler, opts.accumulate, opts.normal_fp16 = capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF
        args += ["--sampler", "1", "--accumulate", "1", "--normal-fp16", "0"]
    sc = mk()
    engine.Options = opts
    engine.Bake(sc)
    path = tmp_path / "scene.aobs"
    write_scene(mk(), str(path))
    out = tmp_path / "ao"
    r = subprocess.run([CLI, "--scene", str(path), "--out", str(out)] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "surfels 16384" in r.stdout
    raw = np.fromfile(str(out) + "_0.raw", dtype=np.uint8)
    cpu = mk()
    orc.bake(FlatScene(cpu).desc, opts, threads=8)
    assert np.array_equal(raw, cpu.GetMaterial(0).GetAOTexture().GetData())            # the C++ host's image is the oracle's
    assert np.array_equal(raw, sc.GetMaterial(0).GetAOTexture().GetData())             # ... and the Python host's
    if bpc == 1:
        pgm = open(str(out) + "_0.pgm", "rb").read()
        assert pgm.startswith(b"P5\n128 128\n255\n") and pgm[15:] == raw.tobytes()


def _grid_obj(n=24):
    """A bumpy n x n quad grid with planar UVs and no normals, as OBJ text (right-handed, counter-clockwise)."""
    lines = ["o terrain", "usemtl ground"]
    for j in range(n + 1):
        for i in range(n + 1):
            x, z = i / n, j / n
            y = 0.08 * np.sin(7.0 * x) * np.cos(5.0 * z) + (0.25 if 8 <= i <= 12 and 8 <= j <= 12 else 0.0)
            lines.append(f"v {x:.6f} {y:.6f} {-z:.6f}")
    for j in range(n + 1):
        for i in range(n + 1):
            lines.append(f"vt {i / n:.6f} {j / n:.6f}")
    for j in range(n):
        for i in range(n):
            a, b, c, d = j * (n + 1) + i + 1, j * (n + 1) + i + 2, (j + 1) * (n + 1) + i + 2, (j + 1) * (n + 1) + i + 1
            lines.append(f"f {a}/{a} {b}/{b} {c}/{c} {d}/{d}")
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("bpc", [1, 4])
def test_cli_obj_to_png_matches_the_oracle_and_the_python_host(engine, orc, tmp_path, bpc):
    """OBJ -> BakeObject::LoadFromFile -> Bake -> BakeTexture::SavePNG / SaveEXR in C++: the baked bytes are the ORACLE's bake of the
    imported scene (the importer's conventions — triangulation, generated normals, handedness, V flip — are checked on the CPU in
    tests/test_assets.py; here they carry a bake), and the files are the Python chain's, byte for byte."""
    from aobaker_amd import assets
    from aobaker_amd.scene import Scene
    assert os.path.exists(CLI), "build the C++ host side: make -C aobaker_amd/host"
    obj = tmp_path / "grid.obj"
    obj.write_text(_grid_obj())
    sc = Scene()
    assets.load_obj(str(obj), sc, atlas=(96, 96), channelCount=1, bytesPerChannel=bpc)
    opts = capi.default_options()
    opts.sample_count, opts.seed = 32, 5
    opts.sampler, opts.accumulate, opts.normal_fp16 = capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF
    engine.Options = opts
    engine.Bake(sc)
    tex = sc.GetMaterial(0).GetAOTexture()
    ao = tex.ao()
    assert 0.2 < ao.mean() < 0.999 and ao.min() < 0.9          # the raised block occludes its surroundings
    out = tmp_path / "ao"
    r = subprocess.run([CLI, "--obj", str(obj), "--atlas", "96", "--bpc", str(bpc), "--out", str(out), "--png", "--exr",
                        "--spp", "32", "--seed", "5", "--sampler", "1", "--accumulate", "1", "--normal-fp16", "0"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cli_raw = np.fromfile(str(out) + "_0.raw", dtype=np.uint8)
    cpu = Scene()
    assets.load_obj(str(obj), cpu, atlas=(96, 96), channelCount=1, bytesPerChannel=bpc)
    st = orc.bake(FlatScene(cpu).desc, opts, threads=8)
    assert st.rays > 0 and 0.0 < st.hits / st.rays < 1.0
    assert np.array_equal(cli_raw, cpu.GetMaterial(0).GetAOTexture().GetData())        # OBJ -> C++ importer -> HIP bake == OBJ -> oracle bake
    assert np.array_equal(cli_raw, tex.GetData())
    assert open(str(out) + "_0.png", "rb").read() == assets.png_bytes(tex)
    img = assets.read_png(str(out) + "_0.png")
    assert img.shape == (96, 96, 1) and img.dtype == (np.uint8 if bpc == 1 else np.uint16)
    assert open(str(out) + "_0.exr", "rb").read() == assets.exr_bytes(tex)
    y = assets.read_exr(str(out) + "_0.exr")["Y"]                  # the bake at the texture's full precision
    assert np.array_equal(y, ao.astype(np.float32))
