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
