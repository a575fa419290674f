"""Mesh-file inputs of the hot path: `Dataset('mesh', paths)` of the reference (main.py:29-39 -> mesh_to_pc.py:42-57).

The reference loads the file with trimesh, draws 4096 surface points with `mesh.sample(n, return_index=True)` and attaches the normal
of the face each point fell on; the (4096, 6) float16 cloud then takes the same road as a `pc_normal` .npy.  trimesh is not installed
here, so this module restates the two things it is used for, in numpy:

* reading triangle soups from .obj / .ply (ascii, binary little / big endian) / .off / .stl (ascii, binary): vertices + faces, polygons
  fanned into triangles, vertex indices that are negative (OBJ relative) or out of range handled / rejected;
* area-weighted uniform surface sampling: a face is drawn with probability proportional to its area (inverse CDF on the cumulative
  areas), a point inside it by two uniforms folded back into the triangle (u + v > 1 -> (1 - u, 1 - v)), the published method
  trimesh's `sample_surface` follows.  Draws come from the GLOBAL numpy RNG, like every other random choice of the reference's
  input side (main.py:129-133 seeds it), in the order faces -> barycentric pairs.

Parity: *unpinned* -- the draws are NOT trimesh's.  The order follows what trimesh 4.2.3 (the reference's pin) publishes for
`sample_surface` (`random(count)` for the faces, then `random((count, 2, 1))` for the barycentric pairs, both from the global numpy RNG),
but there is no trimesh here to generate fixtures, and the face areas / their cumulative sums may round differently: treat the sampled
cloud as this package's own, statistically equivalent, not draw-for-draw equal.  What is tested instead
(tests/test_mesh_input.py) are the properties the rest of the path relies on: points on the surface, unit normals of the face under
each point, density proportional to area, identical geometry through every file format.

`--mc` (mesh2sdf + marching cubes to make the input watertight first, mesh_to_pc.py:13-40) needs mesh2sdf and scikit-image, neither
of which is installed: it raises NotImplementedError with that explanation.
"""
from __future__ import annotations

import os
import struct
from typing import List, Tuple

import numpy as np

Mesh = Tuple[np.ndarray, np.ndarray]          # vertices (V, 3) float64, faces (F, 3) int64


def _fan(polys: List[List[int]]) -> np.ndarray:
    tris = []
    for p in polys:
        for i in range(1, len(p) - 1):
            tris.append((p[0], p[i], p[i + 1]))
    return np.asarray(tris, dtype=np.int64).reshape(-1, 3)


def _load_obj(path: str) -> Mesh:
    verts: List[List[float]] = []
    polys: List[List[int]] = []
    with open(path, "r", errors="replace") as f:
        for line in f:
            if line.startswith("v "):
                verts.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)           # negative = relative to the vertices read so far
                if len(idx) >= 3:
                    polys.append(idx)
    return np.asarray(verts, dtype=np.float64).reshape(-1, 3), _fan(polys)


def _load_off(path: str) -> Mesh:
    with open(path, "r", errors="replace") as f:
        toks = [t for line in f for t in line.split("#")[0].split()]
    if not toks or not toks[0].upper().startswith("OFF"):
        raise ValueError(f"{path}: not an OFF file")
    head = toks[0][3:]
    toks = ([head] if head else []) + toks[1:]
    nv, nf = int(toks[0]), int(toks[1])
    pos = 3
    verts = np.asarray([float(t) for t in toks[pos:pos + 3 * nv]], dtype=np.float64).reshape(nv, 3)
    pos += 3 * nv
    polys = []
    for _ in range(nf):
        n = int(toks[pos])
        polys.append([int(t) for t in toks[pos + 1:pos + 1 + n]])
        pos += 1 + n
    return verts, _fan(polys)


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def _load_ply(path: str) -> Mesh:
    with open(path, "rb") as f:
        raw = f.read()
    end = raw.find(b"end_header")
    if not raw.startswith(b"ply") or end < 0:
        raise ValueError(f"{path}: not a PLY file")
    body = raw.find(b"\n", end) + 1
    fmt = None
    elements = []                                                # (name, count, [(kind, ...)])
    for line in raw[:end].decode("ascii", "replace").splitlines():
        t = line.split()
        if not t:
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            elements.append((t[1], int(t[2]), []))
        elif t[0] == "property" and elements:
            if t[1] == "list":
                elements[-1][2].append(("list", _PLY_TYPES[t[2]], _PLY_TYPES[t[3]], t[4]))
            else:
                elements[-1][2].append(("scalar", _PLY_TYPES[t[1]], t[2]))
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise ValueError(f"{path}: unsupported PLY format {fmt}")
    verts = np.zeros((0, 3))
    polys: List[List[int]] = []
    if fmt == "ascii":
        toks = raw[body:].split()
        pos = 0
        for name, count, props in elements:
            rows = []
            for _ in range(count):
                row = {}
                for p in props:
                    if p[0] == "scalar":
                        row[p[2]] = float(toks[pos]); pos += 1
                    else:
                        n = int(toks[pos]); pos += 1
                        row[p[3]] = [int(float(t)) for t in toks[pos:pos + n]]; pos += n
                rows.append(row)
            if name == "vertex":
                verts = np.asarray([[r["x"], r["y"], r["z"]] for r in rows], dtype=np.float64).reshape(-1, 3)
            elif name == "face":
                key = next((p[3] for p in props if p[0] == "list"), None)
                polys = [r[key] for r in rows if key and len(r[key]) >= 3]
        return verts, _fan(polys)
    bo = "<" if fmt == "binary_little_endian" else ">"
    pos = body
    for name, count, props in elements:
        if all(p[0] == "scalar" for p in props):
            dt = np.dtype([(p[2], bo + p[1]) for p in props])
            arr = np.frombuffer(raw, dtype=dt, count=count, offset=pos)
            pos += dt.itemsize * count
            if name == "vertex":
                verts = np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float64)
            continue
        rows = []
        for _ in range(count):
            row = {}
            for p in props:
                if p[0] == "scalar":
                    dt = np.dtype(bo + p[1])
                    row[p[2]] = np.frombuffer(raw, dtype=dt, count=1, offset=pos)[0]; pos += dt.itemsize
                else:
                    dn, di = np.dtype(bo + p[1]), np.dtype(bo + p[2])
                    n = int(np.frombuffer(raw, dtype=dn, count=1, offset=pos)[0]); pos += dn.itemsize
                    row[p[3]] = np.frombuffer(raw, dtype=di, count=n, offset=pos).astype(np.int64).tolist(); pos += di.itemsize * n
            rows.append(row)
        if name == "face":
            key = next((p[3] for p in props if p[0] == "list"), None)
            polys = [r[key] for r in rows if key and len(r[key]) >= 3]
    return verts, _fan(polys)


def _load_stl(path: str) -> Mesh:
    with open(path, "rb") as f:
        raw = f.read()
    tri = None
    if len(raw) >= 84:
        n = struct.unpack_from("<I", raw, 80)[0]
        if 84 + 50 * n == len(raw):                              # binary: 80-byte header, count, 50-byte records
            rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
            tri = rec["v"].astype(np.float64)
    if tri is None:
        pts = [[float(x) for x in line.split()[1:4]] for line in raw.decode("ascii", "replace").splitlines() if line.strip().startswith("vertex")]
        tri = np.asarray(pts, dtype=np.float64).reshape(-1, 3, 3)
    flat = tri.reshape(-1, 3)
    verts, inv = np.unique(flat, axis=0, return_inverse=True)    # STL repeats every corner: merge identical positions
    return verts, np.asarray(inv, dtype=np.int64).reshape(-1, 3)


_LOADERS = {".obj": _load_obj, ".ply": _load_ply, ".off": _load_off, ".stl": _load_stl}


def load_mesh(path: str) -> Mesh:
    """`trimesh.load(path)` as far as main.py:31-33 uses it: vertices and triangular faces of a mesh file."""
    ext = os.path.splitext(path)[1].lower()
    if ext not in _LOADERS:
        raise ValueError(f"{path}: unsupported mesh format {ext or '(none)'}; supported: {', '.join(sorted(_LOADERS))}")
    verts, faces = _LOADERS[ext](path)
    if verts.shape[0] == 0 or faces.shape[0] == 0:
        raise ValueError(f"{path}: no triangles found")
    if faces.min() < 0 or faces.max() >= verts.shape[0]:
        raise ValueError(f"{path}: a face refers to vertex {int(faces.max())} of {verts.shape[0]}")
    if not np.isfinite(verts).all():
        raise ValueError(f"{path}: non-finite vertex coordinates")
    return verts, faces


def face_normals_and_areas(vertices: np.ndarray, faces: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    tri = vertices[faces]
    cross = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    norm = np.linalg.norm(cross, axis=1)
    normals = np.zeros_like(cross)
    ok = norm > 0
    normals[ok] = cross[ok] / norm[ok, None]
    return normals, 0.5 * norm


def sample_surface(vertices: np.ndarray, faces: np.ndarray, count: int) -> Tuple[np.ndarray, np.ndarray]:
    """`mesh.sample(count, return_index=True)` (mesh_to_pc.py:52): `count` points uniformly distributed over the surface and the index
    of the face under each.  Degenerate (zero-area) faces are never drawn."""
    _, areas = face_normals_and_areas(vertices, faces)
    cum = np.cumsum(areas)
    if not cum[-1] > 0:
        raise ValueError("the mesh has no surface area")
    pick = np.random.random(count) * cum[-1]
    face_idx = np.minimum(np.searchsorted(cum, pick, side="right"), len(cum) - 1)
    uv = np.random.random((count, 2))
    fold = uv.sum(axis=1) > 1.0
    uv[fold] = 1.0 - uv[fold]
    tri = vertices[faces[face_idx]]
    points = tri[:, 0] + uv[:, :1] * (tri[:, 1] - tri[:, 0]) + uv[:, 1:] * (tri[:, 2] - tri[:, 0])
    return points, face_idx


def mesh_to_pc_normal(vertices: np.ndarray, faces: np.ndarray, sample_num: int = 4096) -> np.ndarray:
    """One mesh of `process_mesh_to_pc(mesh_list, marching_cubes=False)` (mesh_to_pc.py:42-57): (sample_num, 6) float16 =
    surface points + the normal of the face under each."""
    points, face_idx = sample_surface(vertices, faces, sample_num)
    normals, _ = face_normals_and_areas(vertices, faces)
    return np.concatenate([points, normals[face_idx]], axis=-1, dtype=np.float16)
