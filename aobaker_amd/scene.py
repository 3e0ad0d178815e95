"""Host-side mirror of the reference scene model, reduced to what Bake reads.

Names and accessor meaning follow the reference:
`Scene.GetObjectCount/GetObject/GetMaterialCount/GetMaterial` (Scene/Scene.h:21-32),
`BakeObject.GetSectionCount/GetSection/BuildTransform` (Scene/Objects/BakeObject.h:60-61,
TransformObject.cpp:16-24), `Section.GetPreviewMesh/GetMaterial` (BakeObject.h:32-35),
`RenderMesh.GetVertexCount/GetVertices/GetIndexCount/GetIndices` (Render/RenderMesh.h:32-35),
`MaterialObject.GetAOTexture` (MaterialObject.h:13) and
`BakeTexture.GetWidth/GetHeight/GetChannelCount/GetBytesPerChannel/GetData`
(Scene/BakeTexture.h:9-18).  The compiled twin is include/aobaker_scene.h.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import capi

# Render/RenderMesh.h:13-19 — 48 bytes: pos@0, normal@12, uv@24, color@32
RENDER_VERTEX = np.dtype([("Position", "<f4", 3), ("Normal", "<f4", 3),
                          ("UV", "<f4", 2), ("Color", "<f4", 4)])
assert RENDER_VERTEX.itemsize == 48


class BakeTexture:
    """Scene/BakeTexture.cpp:7-67: CPU pixel store, initialised to 0xFF."""

    _VALID = {1: (1, 2, 4), 2: (1, 2, 4), 4: (1, 2, 3, 4)}

    def __init__(self, width, height, channelCount=1, bytesPerChannel=1):
        if bytesPerChannel not in self._VALID or channelCount not in self._VALID[bytesPerChannel]:
            raise ValueError("Invalid combination of channelCount and bytesPerChannel.")
        self.Width, self.Height = int(width), int(height)
        self.ChannelCount, self.BytesPerChannel = int(channelCount), int(bytesPerChannel)
        self.Data = np.full(self.Width * self.Height * self.ChannelCount * self.BytesPerChannel,
                            0xFF, dtype=np.uint8)

    def GetWidth(self): return self.Width
    def GetHeight(self): return self.Height
    def GetChannelCount(self): return self.ChannelCount
    def GetBytesPerChannel(self): return self.BytesPerChannel
    def GetData(self): return self.Data
    def GetDataLength(self): return self.Data.size
    def RefreshPreview(self): pass      # GPU preview upload: a no-op headless

    def as_array(self):
        """(H, W, C) view in the texture's own element type."""
        dt = {1: np.uint8, 2: np.float16, 4: np.float32}[self.BytesPerChannel]
        return self.Data.view(dt).reshape(self.Height, self.Width, self.ChannelCount)

    def ao(self):
        """Channel 0 as float32 in [0,1]."""
        a = self.as_array()[..., 0]
        return a.astype(np.float32) / 255.0 if self.BytesPerChannel == 1 else a.astype(np.float32)


class MaterialObject:
    def __init__(self, name="Material", width=512, height=512, channelCount=1, bytesPerChannel=1):
        # MaterialObject.cpp:7 hard-codes BakeTexture(512, 512, 1, 1); here it is a parameter
        self.Name = name
        self.AOTexture = BakeTexture(width, height, channelCount, bytesPerChannel)
        self.AOMaxRange = 1.0           # MaterialObject.cpp:4; read by Bake only when AobOptions.material_ranges is set

    def GetAOTexture(self): return self.AOTexture
    def GetAOMaxRange(self): return self.AOMaxRange
    def SetAOMaxRange(self, v): self.AOMaxRange = float(v)


class RenderMesh:
    """Render/RenderMesh.cpp:29-45: keeps CPU copies of vertices and indices."""

    def __init__(self, vertices, indices):
        v = np.ascontiguousarray(vertices)
        if v.dtype != RENDER_VERTEX:
            raise TypeError("vertices must be a RENDER_VERTEX array")
        self.Vertices = v
        self.Indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)

    def GetVertexCount(self): return int(self.Vertices.shape[0])
    def GetVertices(self): return self.Vertices
    def GetIndexCount(self): return int(self.Indices.shape[0])
    def GetIndices(self): return self.Indices


def make_vertices(positions, normals, uvs):
    n = len(positions)
    v = np.zeros(n, dtype=RENDER_VERTEX)
    v["Position"] = positions
    v["Normal"] = normals
    v["UV"] = uvs
    v["Color"] = 1.0
    return v


class Section:
    def __init__(self, name, mesh: RenderMesh, material: MaterialObject | None):
        self.Name, self.PreviewMesh, self.Material = name, mesh, material
        self.VisibleToSelf = True       # BakeObject.cpp:20: both default to true
        self.VisibleToOthers = True

    def GetPreviewMesh(self): return self.PreviewMesh
    def GetMaterial(self): return self.Material
    def IsVisibleToSelf(self): return self.VisibleToSelf
    def SetVisibleToSelf(self, v): self.VisibleToSelf = bool(v)
    def IsVisibleToOthers(self): return self.VisibleToOthers
    def SetVisibleToOthers(self, v): self.VisibleToOthers = bool(v)


def _quat_yaw_pitch_roll(yaw, pitch, roll):
    # DirectXMath XMQuaternionRotationRollPitchYaw(pitch, yaw, roll) as called by
    # SimpleMath Quaternion::CreateFromYawPitchRoll (SimpleMath.inl:3072-3078), in float32
    f = np.float32
    hp, hy, hr = f(pitch) * f(0.5), f(yaw) * f(0.5), f(roll) * f(0.5)
    sp, cp = f(math.sin(hp)), f(math.cos(hp))
    sy, cy = f(math.sin(hy)), f(math.cos(hy))
    sr, cr = f(math.sin(hr)), f(math.cos(hr))
    x = cr * sp * cy + sr * cp * sy
    y = cr * cp * sy - sr * sp * cy
    z = sr * cp * cy - cr * sp * sy
    w = cr * cp * cy + sr * sp * sy
    return np.array([x, y, z, w], dtype=np.float32)


def _matrix_from_quat(q):
    # XMMatrixRotationQuaternion, row-vector convention
    x, y, z, w = [np.float32(v) for v in q]
    one, two = np.float32(1), np.float32(2)
    m = np.zeros((4, 4), dtype=np.float32)
    m[0, 0] = one - two * y * y - two * z * z
    m[0, 1] = two * x * y + two * z * w
    m[0, 2] = two * x * z - two * y * w
    m[1, 0] = two * x * y - two * z * w
    m[1, 1] = one - two * x * x - two * z * z
    m[1, 2] = two * y * z + two * x * w
    m[2, 0] = two * x * z + two * y * w
    m[2, 1] = two * y * z - two * x * w
    m[2, 2] = one - two * x * x - two * y * y
    m[3, 3] = one
    return m


class BakeObject:
    """TransformObject + sections (Scene/Objects/BakeObject.h, TransformObject.cpp)."""

    def __init__(self, name="Object", position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1)):
        self.Name = name
        self.Position = np.asarray(position, dtype=np.float32)
        self.Rotation = np.asarray(rotation, dtype=np.float32)
        self.Scale = np.asarray(scale, dtype=np.float32)
        self.Sections: list[Section] = []

    def AddSection(self, section: Section):
        self.Sections.append(section)
        return section

    def GetSectionCount(self): return len(self.Sections)
    def GetSection(self, i): return self.Sections[i]

    def BuildTransform(self):
        """TransformObject.cpp:16-24: Scale * Rotation(yaw=rot.y, pitch=rot.x, roll=rot.z) * Translation."""
        s = np.diag(np.append(self.Scale, np.float32(1))).astype(np.float32)
        r = _matrix_from_quat(_quat_yaw_pitch_roll(self.Rotation[1], self.Rotation[0], self.Rotation[2]))
        t = np.eye(4, dtype=np.float32)
        t[3, :3] = self.Position
        return ((s @ r).astype(np.float32) @ t).astype(np.float32)


class Scene:
    def __init__(self):
        self.BakeObjects: list[BakeObject] = []
        self.Materials: list[MaterialObject] = []

    def AddObject(self, o): self.BakeObjects.append(o); return o
    def AddMaterial(self, m): self.Materials.append(m); return m
    def GetObjectCount(self): return len(self.BakeObjects)
    def GetObject(self, i): return self.BakeObjects[i]
    def GetMaterialCount(self): return len(self.Materials)
    def GetMaterial(self, i): return self.Materials[i]
    def FindMaterial(self, m):
        for i, x in enumerate(self.Materials):
            if x is m:
                return i
        return -1


class FlatScene:
    """An AobSceneDesc plus the arrays it points into (kept alive with it)."""

    def __init__(self, scene: Scene):
        self.scene = scene
        meshes = []
        self._keep = []
        for m in range(scene.GetObjectCount()):
            obj = scene.GetObject(m)
            model = np.ascontiguousarray(obj.BuildTransform(), dtype=np.float32)
            for s in range(obj.GetSectionCount()):
                sec = obj.GetSection(s)
                mesh = sec.GetPreviewMesh()
                v, idx = mesh.GetVertices(), mesh.GetIndices()
                self._keep += [v, idx]
                d = capi.AobMeshDesc()
                base = v.ctypes.data
                d.positions = base + RENDER_VERTEX.fields["Position"][1]
                d.normals = base + RENDER_VERTEX.fields["Normal"][1]
                d.uvs = base + RENDER_VERTEX.fields["UV"][1]
                d.position_stride = d.normal_stride = d.uv_stride = RENDER_VERTEX.itemsize
                d.vertex_count = mesh.GetVertexCount()
                d.indices = idx.ctypes.data
                d.index_count = mesh.GetIndexCount()
                # material <-> section association is by identity (BakeEngine.cpp:523)
                d.material = scene.FindMaterial(sec.GetMaterial()) if sec.GetMaterial() is not None else -1
                d.model = (C.c_float * 16)(*model.reshape(-1).tolist())
                d.flags = (0 if sec.IsVisibleToSelf() else capi.MESH_HIDDEN_FROM_SELF) | \
                          (0 if sec.IsVisibleToOthers() else capi.MESH_HIDDEN_FROM_OTHERS)
                meshes.append(d)
        mats = []
        for i in range(scene.GetMaterialCount()):
            tex = scene.GetMaterial(i).GetAOTexture()
            d = capi.AobMaterialDesc()
            d.width, d.height = tex.GetWidth(), tex.GetHeight()
            d.channels, d.bytes_per_channel = tex.GetChannelCount(), tex.GetBytesPerChannel()
            d.data = tex.GetData().ctypes.data
            d.ao_max_range = scene.GetMaterial(i).GetAOMaxRange()
            mats.append(d)
        self.mesh_array = (capi.AobMeshDesc * max(1, len(meshes)))(*meshes)
        self.material_array = (capi.AobMaterialDesc * max(1, len(mats)))(*mats)
        self.desc = capi.AobSceneDesc()
        self.desc.meshes = C.cast(self.mesh_array, C.POINTER(capi.AobMeshDesc))
        self.desc.mesh_count = len(meshes)
        self.desc.materials = C.cast(self.material_array, C.POINTER(capi.AobMaterialDesc))
        self.desc.material_count = len(mats)
        # `FlatScene(scene).desc` is a common way to pass a scene on: the descriptor itself keeps what it points into alive
        # (the vertex / index arrays and, through the scene, the textures the material descriptors write to)
        self.desc._keepalive = (self._keep, self.mesh_array, self.material_array, scene)

    def triangle_count(self):
        return sum(int(self.mesh_array[i].index_count) // 3 for i in range(self.desc.mesh_count))
