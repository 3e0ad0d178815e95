"""Writer of the tiny binary scene format tools/aobake_cli.cpp reads (the C++ host side's input)."""
from __future__ import annotations

import struct

import numpy as np

from .scene import RENDER_VERTEX, Scene


def write_scene(scene: Scene, path: str) -> None:
    with open(path, "wb") as f:
        f.write(b"AOBS")
        f.write(struct.pack("<III", 1, scene.GetMaterialCount(), scene.GetObjectCount()))
        for i in range(scene.GetMaterialCount()):
            t = scene.GetMaterial(i).GetAOTexture()
            f.write(struct.pack("<iiii", t.GetWidth(), t.GetHeight(), t.GetChannelCount(), t.GetBytesPerChannel()))
        for i in range(scene.GetObjectCount()):
            o = scene.GetObject(i)
            f.write(np.concatenate([o.Position, o.Rotation, o.Scale]).astype("<f4").tobytes())
            f.write(struct.pack("<I", o.GetSectionCount()))
            for s in range(o.GetSectionCount()):
                sec = o.GetSection(s)
                mesh = sec.GetPreviewMesh()
                mat = scene.FindMaterial(sec.GetMaterial()) if sec.GetMaterial() is not None else -1
                f.write(struct.pack("<iII", mat, mesh.GetVertexCount(), mesh.GetIndexCount()))
                v = np.ascontiguousarray(mesh.GetVertices())
                assert v.dtype == RENDER_VERTEX
                f.write(v.tobytes())
                f.write(np.ascontiguousarray(mesh.GetIndices(), dtype="<u4").tobytes())


def read_scene(path: str) -> Scene:
    """Reader of the same format (what tools/aobake_assets.cpp obj2aobs writes)."""
    from .scene import BakeObject, MaterialObject, RenderMesh, Section
    data = open(path, "rb").read()
    assert data[:4] == b"AOBS"
    version, nmat, nobj = struct.unpack_from("<III", data, 4)
    assert version == 1
    at = 16
    scene = Scene()
    for _ in range(nmat):
        w, h, ch, bpc = struct.unpack_from("<iiii", data, at); at += 16
        scene.AddMaterial(MaterialObject("Material", w, h, ch, bpc))
    for _ in range(nobj):
        trs = np.frombuffer(data, dtype="<f4", count=9, offset=at); at += 36
        (nsec,) = struct.unpack_from("<I", data, at); at += 4
        obj = scene.AddObject(BakeObject("Object", trs[0:3], trs[3:6], trs[6:9]))
        for _ in range(nsec):
            mat, nv, ni = struct.unpack_from("<iII", data, at); at += 12
            v = np.frombuffer(data, dtype=RENDER_VERTEX, count=nv, offset=at).copy(); at += 48 * nv
            idx = np.frombuffer(data, dtype="<u4", count=ni, offset=at).copy(); at += 4 * ni
            obj.AddSection(Section("Geometry", RenderMesh(v, idx), scene.GetMaterial(mat) if mat >= 0 else None))
    assert at == len(data)
    return scene
