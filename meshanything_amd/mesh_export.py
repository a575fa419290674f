"""Output side of the hot path: (n_max_triangles, 3, 3) coordinates -> a triangle mesh file (main.py:156-175).

The reference hands the vertices to trimesh (`Trimesh(..., merge_primitives=True)`, `merge_vertices()`,
`update_faces(unique_faces())`, `fix_normals()`, `export(.obj)`).  trimesh is not part of this image; the steps that
change the *content* of the file are restated here on exact coordinates (the detokenizer emits multiples of 1/128, so
"same vertex" is exact equality -- no tolerance needed):
  drop NaN faces -> merge identical vertices -> drop faces that repeat an earlier face's vertex SET (trimesh's
  unique_faces sorts each face's indices) -> write OBJ.
`fix_normals()` = trimesh.repair.fix_winding + fix_inversion: faces that share an edge are given consistent winding by a
breadth-first traversal of the face-adjacency graph (edges shared by exactly two faces); then -- `Trimesh.fix_normals`
resolves `multibody=None` to `body_count > 1`, and MeshAnything outputs usually have several bodies -- every connected
body whose own signed volume is negative is flipped (one body: the whole mesh by its total volume, which is the same
thing).  Faces that share no manifold edge with any other face are not part of any body in trimesh's
`connected_components(face_adjacency)` and keep their winding.  `fix_normals` below restates that; where a body is not
orientable (or an edge has more than two faces) the result depends on the traversal order, here lowest face index first --
trimesh's order comes from networkx and is not reproduced bit for bit.  (trimesh is not installed in this image: the
restatement follows its published algorithm and cannot be pinned against it here.)
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def faces_from_coords(coords: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """coords (F, 3, 3) with NaN rows for invalid faces -> (vertices (V, 3) float32, faces (M, 3) int64)."""
    coords = np.asarray(coords, dtype=np.float32)
    valid = ~np.isnan(coords[:, 0, 0])                         # main.py:158
    tri = coords[valid].reshape(-1, 3)                          # 3 * n_valid vertices, face i = rows 3i..3i+2
    if tri.shape[0] == 0:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64)
    verts, inverse = np.unique(tri, axis=0, return_inverse=True)            # merge_vertices on exact coordinates
    faces = inverse.reshape(-1, 3).astype(np.int64)
    key = np.sort(faces, axis=1)
    _, first = np.unique(key, axis=0, return_index=True)                     # unique_faces: first occurrence of each vertex set
    faces = faces[np.sort(first)]
    return verts.astype(np.float32), faces


def fix_normals(verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Consistent winding across shared edges, then outward orientation (positive signed volume) per connected body.
    Returns new faces."""
    faces = np.array(faces, dtype=np.int64, copy=True)
    n = len(faces)
    if n == 0:
        return faces
    # undirected edge -> faces using it
    edge_faces = {}
    for f in range(n):
        a, b, c = faces[f]
        for u, v in ((a, b), (b, c), (c, a)):
            edge_faces.setdefault((min(u, v), max(u, v)), []).append(f)
    adj = [[] for _ in range(n)]
    for (u, v), fs in edge_faces.items():
        if len(fs) == 2:                                     # manifold edges only, as trimesh.face_adjacency
            adj[fs[0]].append((fs[1], u, v))
            adj[fs[1]].append((fs[0], u, v))

    def directed(f, u, v):                                   # does face f traverse the edge as u -> v ?
        a, b, c = faces[f]
        return (a == u and b == v) or (b == u and c == v) or (c == u and a == v)

    seen = np.zeros(n, dtype=bool)
    bodies = []
    for root in range(n):
        if seen[root]:
            continue
        seen[root] = True
        queue = [root]
        body = [root]
        bodies.append(body)
        while queue:
            f = queue.pop(0)
            for g, u, v in sorted(adj[f]):
                if seen[g]:
                    continue
                # consistently wound neighbours traverse their shared edge in opposite directions
                if directed(f, u, v) == directed(g, u, v):
                    faces[g] = faces[g][::-1]
                seen[g] = True
                queue.append(g)
                body.append(g)
    tri = verts[faces].astype(np.float64)
    signed6 = np.einsum("ij,ij->i", tri[:, 0], np.cross(tri[:, 1], tri[:, 2]))      # 6 x signed tetrahedron volume per face
    for body in bodies:
        if len(body) > 1 and signed6[body].sum() < 0:
            faces[body] = faces[body][:, ::-1]
    return faces


FACE_COLOR = (255, 165, 0)                                   # main.py:170 (`brown_color`, alpha 255)


def write_obj(path: str, verts: np.ndarray, faces: np.ndarray, color=FACE_COLOR) -> None:
    """main.py:170-174 sets one colour on every face and calls `export(.obj)`.  The OBJ format has no per-face colour; trimesh's
    OBJ writer turns face colours into vertex colours and appends them to the vertex lines (`v x y z r g b`, components in
    0..1) -- with one colour for all faces every vertex gets that colour.  `color=None` writes plain `v x y z` lines."""
    tail = "" if color is None else " " + " ".join(f"{c / 255.0:.8f}" for c in color[:3])
    with open(path, "w") as f:
        f.write("# MeshAnything (meshanything_amd)\n")
        for v in verts:
            f.write(f"v {v[0]:.8f} {v[1]:.8f} {v[2]:.8f}{tail}\n")
        for t in faces:
            f.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
