"""Seeded synthetic scenes for the BASELINE.json configs and the closed-form tests.

There are no assets in the container (no bunny OBJ, assimp is an empty
submodule in the reference), so every input is generated here.
Config mapping: cfg1 `cube_scene`, cfg2 `sphere_scene`
(stand-in for the ~70 k-tri bunny), cfg3/cfg4 `terrain_scene`.
"""
from __future__ import annotations

import numpy as np

from .scene import BakeObject, MaterialObject, RenderMesh, Scene, Section, make_vertices


def _scene_with(meshes, atlas, channels=1, bpc=1, transforms=None):
    """meshes: list of (positions, normals, uvs, indices); one object + section each."""
    if isinstance(atlas, int):
        atlas = (atlas, atlas)
    sc = Scene()
    mat = sc.AddMaterial(MaterialObject("AO", atlas[0], atlas[1], channels, bpc))
    for i, (p, n, uv, idx) in enumerate(meshes):
        kw = transforms[i] if transforms else {}
        obj = sc.AddObject(BakeObject(f"Object{i}", **kw))
        obj.AddSection(Section("Geometry", RenderMesh(make_vertices(p, n, uv), idx), mat))
    return sc


def quad(p00, p10, p11, p01, normal, uv00=(0, 0), uv11=(1, 1)):
    """One quad as 4 vertices / 2 triangles; uv00 at p00, uv11 at p11."""
    p = np.array([p00, p10, p11, p01], dtype=np.float32)
    n = np.tile(np.asarray(normal, dtype=np.float32), (4, 1))
    (u0, v0), (u1, v1) = uv00, uv11
    uv = np.array([[u0, v0], [u1, v0], [u1, v1], [u0, v1]], dtype=np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    return p, n, uv, idx


def merge(parts):
    ps, ns, uvs, idxs, base = [], [], [], [], 0
    for p, n, uv, idx in parts:
        ps.append(p); ns.append(n); uvs.append(uv); idxs.append(idx + base)
        base += len(p)
    return (np.concatenate(ps), np.concatenate(ns), np.concatenate(uvs),
            np.concatenate(idxs).astype(np.uint32))


def plane_scene(atlas=64, size=1.0, **kw):
    """Open unit plane at y=0 facing +y, UV = xz: nothing above it, AO = 1 everywhere."""
    s = size
    return _scene_with([quad((0, 0, 0), (s, 0, 0), (s, 0, s), (0, 0, s), (0, 1, 0))], atlas, **kw)


def _cube_faces(inward):
    h = 0.5
    # (origin corner, u axis, v axis, outward normal)
    faces = [((-h, -h, h), (1, 0, 0), (0, 1, 0), (0, 0, 1)),
             ((h, -h, -h), (-1, 0, 0), (0, 1, 0), (0, 0, -1)),
             ((h, -h, h), (0, 0, -1), (0, 1, 0), (1, 0, 0)),
             ((-h, -h, -h), (0, 0, 1), (0, 1, 0), (-1, 0, 0)),
             ((-h, h, h), (1, 0, 0), (0, 0, -1), (0, 1, 0)),
             ((-h, -h, -h), (1, 0, 0), (0, 0, 1), (0, -1, 0))]
    out = []
    for o, ua, va, n in faces:
        o, ua, va, n = [np.asarray(x, dtype=np.float32) for x in (o, ua, va, n)]
        out.append((o, ua, va, -n if inward else n))
    return out


def cube_scene(atlas=256, inward=False, gutter=2, **kw):
    """BASELINE cfg1: unit cube, 12 triangles (the geometry of CubeVertices/CubeIndices,
    Scene/Objects/BakeObject.cpp:197-225, but with per-face normals and a 6-chart 3x2 atlas
    with `gutter`-texel gutters, since the in-tree fixture has zero normals and UVs).
    inward=True flips the normals: baked on its inner faces a closed cube is fully occluded."""
    W = H = atlas
    cw, ch = W // 3, H // 2
    parts = []
    for f, (o, ua, va, n) in enumerate(_cube_faces(inward)):
        cx, cy = f % 3, f // 3
        u0, v0 = (cx * cw + gutter) / W, (cy * ch + gutter) / H
        u1, v1 = ((cx + 1) * cw - gutter) / W, ((cy + 1) * ch - gutter) / H
        parts.append(quad(o, o + ua, o + ua + va, o + va, n, (u0, v0), (u1, v1)))
    return _scene_with([merge(parts)], atlas, **kw)


def floor_ceiling_scene(atlas=32, height=0.5, extent=200.0, **kw):
    """Unit floor patch under a very large parallel ceiling at y=height (occluder only)."""
    floor = quad((0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0))
    e = extent
    ceil_ = quad((-e, height, -e), (e, height, -e), (e, height, e), (-e, height, e), (0, -1, 0))
    sc = _scene_with([floor], atlas, **kw)
    obj = sc.AddObject(BakeObject("Ceiling"))
    obj.AddSection(Section("Geometry", RenderMesh(make_vertices(*ceil_[:3]), ceil_[3]), None))
    return sc


def floor_wall_scene(atlas=32, wall_height=1.0, length=2000.0, floor_depth=2.0, **kw):
    """Floor strip next to an (effectively) infinitely long perpendicular wall at x=0.
    Cosine-weighted AO at distance d from the wall is 1/2 (1 + d / sqrt(d^2 + H^2))."""
    L = length
    # floor patch: x in [0, floor_depth], z in [0,1]; u along x
    floor = quad((0, 0, 0), (floor_depth, 0, 0), (floor_depth, 0, 1), (0, 0, 1), (0, 1, 0))
    wall = quad((0, 0, -L), (0, 0, L), (0, wall_height, L), (0, wall_height, -L), (1, 0, 0))
    sc = _scene_with([floor], atlas, **kw)
    obj = sc.AddObject(BakeObject("Wall"))
    obj.AddSection(Section("Geometry", RenderMesh(make_vertices(*wall[:3]), wall[3]), None))
    return sc


# ---- value-noise fBm ------------------------------------------------------------

def _value_noise(x, y, lattice):
    n = lattice.shape[0]
    xi, yi = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    fx, fy = x - xi, y - yi
    sx, sy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
    x0, y0, x1, y1 = xi % n, yi % n, (xi + 1) % n, (yi + 1) % n
    a, b = lattice[y0, x0], lattice[y0, x1]
    c, d = lattice[y1, x0], lattice[y1, x1]
    return (a + (b - a) * sx) * (1 - sy) + (c + (d - c) * sx) * sy


def fbm(x, y, seed, octaves=6, base_freq=4.0):
    rng = np.random.default_rng(seed)
    out = np.zeros_like(x, dtype=np.float64)
    amp, freq, norm = 1.0, base_freq, 0.0
    for _ in range(octaves):
        lattice = rng.random((256, 256))
        out += amp * (_value_noise(x * freq, y * freq, lattice) - 0.5)
        norm += amp
        amp *= 0.5
        freq *= 2.0
    return out / norm


def _grid_indices(nx, ny):
    """Triangle list of an nx*ny vertex grid (row-major vertices): 2*(nx-1)*(ny-1) triangles."""
    j, i = np.meshgrid(np.arange(ny - 1), np.arange(nx - 1), indexing="ij")
    v00 = (j * nx + i).reshape(-1)
    v10, v01, v11 = v00 + 1, v00 + nx, v00 + nx + 1
    tri = np.stack([v00, v01, v10, v10, v01, v11], axis=1)
    return tri.reshape(-1).astype(np.uint32)


def terrain_mesh(grid=708, seed=3, amplitude=0.15):
    """BASELINE cfg3/4: grid x grid heightfield over [0,1]^2 (708 -> 999 698 triangles),
    6-octave value-noise fBm of amplitude 0.15 x extent, planar UV = xz (100 % coverage)."""
    t = np.linspace(0.0, 1.0, grid)
    x, z = np.meshgrid(t, t, indexing="xy")           # row j = z, column i = x
    hgt = 2.0 * amplitude * fbm(x, z, seed)
    p = np.stack([x, hgt, z], axis=-1).reshape(-1, 3).astype(np.float32)
    step = 1.0 / (grid - 1)
    dhdx = np.gradient(hgt, step, axis=1)
    dhdz = np.gradient(hgt, step, axis=0)
    n = np.stack([-dhdx, np.ones_like(hgt), -dhdz], axis=-1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    uv = np.stack([x, z], axis=-1).reshape(-1, 2).astype(np.float32)
    return p, n.reshape(-1, 3).astype(np.float32), uv, _grid_indices(grid, grid)


def terrain_scene(grid=708, atlas=2048, seed=3, **kw):
    return _scene_with([terrain_mesh(grid, seed)], atlas, **kw)


def sphere_mesh(n=187, seed=2, displacement=0.08):
    """BASELINE cfg2 stand-in for the ~70 k-triangle bunny: lat-long sphere with
    2*n^2 = 69 938 triangles (n=187), seeded radial fBm displacement, equirectangular UV."""
    nu, nv = n + 1, n + 1
    u = np.linspace(0.0, 1.0, nu)
    v = np.linspace(0.0, 1.0, nv)
    uu, vv = np.meshgrid(u, v, indexing="xy")
    theta, phi = vv * np.pi, uu * 2.0 * np.pi
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], axis=-1)
    # periodic in u so the seam closes
    r = 0.5 * (1.0 + 2.0 * displacement * fbm(np.cos(phi) * 0.5 + 0.5 + d[..., 1], np.sin(phi) * 0.5 + 0.5 + d[..., 1] * 0.7, seed, base_freq=3.0))
    p = d * r[..., None]
    # normals from the displaced surface (finite differences in parameter space)
    dpu = np.gradient(p, axis=1)
    dpv = np.gradient(p, axis=0)
    nrm = np.cross(dpu, dpv)                     # outward: d(theta) x d(phi) points along -d
    ln = np.linalg.norm(nrm, axis=-1, keepdims=True)
    nrm = np.where(ln > 1e-12, nrm / np.maximum(ln, 1e-12), d)
    uv = np.stack([uu, vv], axis=-1)
    return (p.reshape(-1, 3).astype(np.float32), nrm.reshape(-1, 3).astype(np.float32),
            uv.reshape(-1, 2).astype(np.float32), _grid_indices(nu, nv))


def sphere_scene(n=187, atlas=1024, seed=2, **kw):
    return _scene_with([sphere_mesh(n, seed)], atlas, **kw)


def torus_knot_mesh(n_u=700, n_v=50, p=2, q=3, seed=2, tube=0.11, displacement=0.25):
    """A bunny-CLASS mesh for BASELINE configs[1] that is NOT star-shaped (the displaced sphere has no concavities a ray can
    see into, no handles): a (p, q) torus knot — a closed tube that winds p times round the axis and q times through the hole, passing
    over and under itself — with seeded fBm bumps on the tube radius.  2 * n_u * n_v triangles (700 x 50 -> 70 000), one chart: u along
    the knot, v round the tube.  Returns (positions, normals, uvs, indices) like sphere_mesh."""
    u = np.linspace(0.0, 1.0, n_u + 1)
    v = np.linspace(0.0, 1.0, n_v + 1)
    t = 2.0 * np.pi * u
    # the knot's centre line (on a torus of radii 0.5 / 0.2) and its moving frame
    r = 0.5 + 0.2 * np.cos(q * t)
    c = np.stack([r * np.cos(p * t), 0.2 * np.sin(q * t), r * np.sin(p * t)], axis=-1)
    dr = -0.2 * q * np.sin(q * t)
    tan = np.stack([dr * np.cos(p * t) - r * p * np.sin(p * t), 0.2 * q * np.cos(q * t), dr * np.sin(p * t) + r * p * np.cos(p * t)], axis=-1)
    tan /= np.linalg.norm(tan, axis=-1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    b1 = np.cross(tan, up); b1 /= np.linalg.norm(b1, axis=-1, keepdims=True)
    b2 = np.cross(tan, b1)
    uu, vv = np.meshgrid(u, v, indexing="xy")                      # [n_v + 1, n_u + 1]
    phi = 2.0 * np.pi * vv
    # periodic in both parameters so that the tube closes on itself
    bump = fbm(0.5 + 0.5 * np.cos(2 * np.pi * uu) + 0.3 * np.cos(phi), 0.5 + 0.5 * np.sin(2 * np.pi * uu) + 0.3 * np.sin(phi), seed, base_freq=6.0)
    rad = tube * (1.0 + displacement * bump)
    pos = c[None, :, :] + rad[..., None] * (np.cos(phi)[..., None] * b1[None, :, :] + np.sin(phi)[..., None] * b2[None, :, :])
    dpu = np.gradient(pos, axis=1)
    dpv = np.gradient(pos, axis=0)
    nrm = np.cross(dpv, dpu)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-12)
    # outward: away from the centre line
    flip = np.sign((nrm * (pos - c[None, :, :])).sum(-1, keepdims=True))
    nrm = nrm * np.where(flip == 0, 1.0, flip)
    pad = 0.01
    uv = np.stack([pad + uu * (1 - 2 * pad), pad + vv * (1 - 2 * pad)], axis=-1)
    return (pos.reshape(-1, 3).astype(np.float32), nrm.reshape(-1, 3).astype(np.float32),
            uv.reshape(-1, 2).astype(np.float32), _grid_indices(n_u + 1, n_v + 1))


def write_obj(path, positions, uvs, indices, normals=None, name="mesh", material="AO"):
    """A triangle mesh as Wavefront OBJ text (what BakeObject::LoadFromFile / assets.load_obj import): v / vt (/ vn) and 1-based f."""
    idx = np.asarray(indices, dtype=np.int64).reshape(-1, 3) + 1
    with open(path, "w") as f:
        f.write(f"o {name}\nusemtl {material}\n")
        for p in np.asarray(positions, dtype=np.float32):
            f.write(f"v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n")
        for t in np.asarray(uvs, dtype=np.float32):
            f.write(f"vt {t[0]:.9g} {t[1]:.9g}\n")
        if normals is not None:
            for n in np.asarray(normals, dtype=np.float32):
                f.write(f"vn {n[0]:.9g} {n[1]:.9g} {n[2]:.9g}\n")
            for a, b, c in idx:
                f.write(f"f {a}/{a}/{a} {b}/{b}/{b} {c}/{c}/{c}\n")
        else:
            for a, b, c in idx:
                f.write(f"f {a}/{a} {b}/{b} {c}/{c}\n")
    return path


def torus_knot_obj(path, n_u=700, n_v=50, seed=2):
    """The 70 000-triangle displaced torus knot as an OBJ file (no normals: the importer generates them, as assimp does for the
    reference, BakeObject.cpp:183) — `AOB_CFG2_OBJ=<this file>` makes it BASELINE configs[1]'s mesh."""
    p, _, uv, idx = torus_knot_mesh(n_u, n_v, seed=seed)
    return write_obj(path, p, uv, idx, name="torus_knot")


def config2_scene(atlas=1024, seed=2, bpc=4):
    """BASELINE configs[1]'s mesh: returns (scene, mesh name).  The Stanford bunny is not in the image (no network), so the
    stand-in is the 69 938-triangle displaced sphere; AOB_CFG2_OBJ=/path/to/mesh.obj bakes that file instead (imported as
    BakeObject::LoadFromFile imports it, assets.load_obj, every material on an atlas x atlas texture) — for whoever has the
    bunny, or any UV-mapped mesh of that size.  The name goes into the records made from the scene."""
    import os
    path = os.environ.get("AOB_CFG2_OBJ")
    if not path:
        return sphere_scene(n=187, atlas=atlas, seed=seed, bpc=bpc), "displaced sphere n=187 (bunny stand-in)"
    from .assets import load_obj
    sc = Scene()
    load_obj(path, sc, atlas=(atlas, atlas), channelCount=1, bytesPerChannel=bpc)
    if sc.GetMaterialCount() == 0:
        raise ValueError(f"AOB_CFG2_OBJ={path}: the file has no faces")
    return sc, os.path.basename(path)


def two_section_scene(atlas=32, **kw):
    """Two sections of ONE material whose charts overlap in UV space: the reference bakes each
    section separately and both accumulate into the same result cells (BakeEngine.cpp:473,659)."""
    a = quad((0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0), (0.0, 0.0), (0.75, 0.75))
    b = quad((0, 0.5, 0), (1, 0.5, 0), (1, 0.5, 1), (0, 0.5, 1), (0, 1, 0), (0.25, 0.25), (1.0, 1.0))
    sc = Scene()
    mat = sc.AddMaterial(MaterialObject("AO", atlas, atlas, kw.get("channels", 1), kw.get("bpc", 1)))
    obj = sc.AddObject(BakeObject("Object"))
    obj.AddSection(Section("A", RenderMesh(make_vertices(*a[:3]), a[3]), mat))
    obj.AddSection(Section("B", RenderMesh(make_vertices(*b[:3]), b[3]), mat))
    return sc


def instanced_scene(instances=100, rock_n=224, atlas=8192, seed=5, **kw):
    """BASELINE cfg5: `instances` copies of one displaced-sphere "rock" (2*rock_n^2 triangles; 224 -> 100 352)
    scattered over a ground plane with seeded yaw/pitch/roll, non-uniform scale and translation —
    every copy its own BakeObject with its own transform (TransformObject.cpp:16-24), all sections
    sharing one material whose atlas is a grid of per-instance charts.  The reference has no
    instancing either (it never calls CreateInstance): each copy is a full shape."""
    rng = np.random.default_rng(seed)
    p, n, uv, idx = sphere_mesh(rock_n, seed)
    cols = int(np.ceil(np.sqrt(instances + 1)))
    rows = int(np.ceil((instances + 1) / cols))
    sc = Scene()
    if isinstance(atlas, int):
        atlas = (atlas, atlas)
    mat = sc.AddMaterial(MaterialObject("AO", atlas[0], atlas[1], kw.get("channels", 1), kw.get("bpc", 1)))

    def chart(k, uv_in):
        cx, cy = k % cols, k // cols
        pad = 0.02
        out = np.empty_like(uv_in)
        out[:, 0] = (cx + pad + uv_in[:, 0] * (1 - 2 * pad)) / cols
        out[:, 1] = (cy + pad + uv_in[:, 1] * (1 - 2 * pad)) / rows
        return out.astype(np.float32)

    ground = quad((-6, 0, -6), (6, 0, -6), (6, 0, 6), (-6, 0, 6), (0, 1, 0))
    g = sc.AddObject(BakeObject("Ground"))
    g.AddSection(Section("Geometry", RenderMesh(make_vertices(ground[0], ground[1], chart(0, ground[2])), ground[3]), mat))
    for i in range(instances):
        o = sc.AddObject(BakeObject(f"Rock{i}",
                                    position=(rng.uniform(-5, 5), rng.uniform(0.1, 0.6), rng.uniform(-5, 5)),
                                    rotation=tuple(rng.uniform(-np.pi, np.pi, 3)),
                                    scale=tuple(rng.uniform(0.5, 1.5, 3))))
        o.AddSection(Section("Geometry", RenderMesh(make_vertices(p, n, chart(i + 1, uv)), idx), mat))
    return sc


def soup_scene(seed=11, n=300, atlas=96, spread=0.02, slivers=True, **kw):
    """A triangle soup in ONE section: `n` random small triangles with random UV charts around random centres —
    overlapping charts (AOB_LAYERS_ALL), and with `slivers` charts thinner than a texel (AOB_RASTER_CONSERVATIVE).
    Seeded; the reference has nothing like it — it exercises what real atlases have and clean synthetic ones lack."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(0, 1, (n, 1, 3))
    v = (c + rng.normal(scale=0.05, size=(n, 3, 3))).astype(np.float32).reshape(-1, 3)
    uvc = rng.uniform(0, 1, (n, 1, 2))
    uv = (uvc + rng.normal(scale=spread, size=(n, 3, 2)) * (rng.uniform(0.02, 1.0, (n, 1, 2)) if slivers else 1.0)).astype(np.float32).reshape(-1, 2)
    nrm = rng.normal(size=(3 * n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    if isinstance(atlas, int):
        atlas = (atlas, atlas)
    sc = Scene()
    m = sc.AddMaterial(MaterialObject("A", atlas[0], atlas[1], kw.get("channels", 1), kw.get("bpc", 1)))
    o = sc.AddObject(BakeObject("Soup"))
    o.AddSection(Section("s", RenderMesh(make_vertices(v, nrm, uv), np.arange(3 * n, dtype=np.uint32)), m))
    return sc
