"""Output side of the hot path: (n_max_triangles, 3, 3) coordinates -> a triangle mesh file (main.py:156-175).

The reference hands the vertices to trimesh (requirements.txt pins trimesh==4.2.3): `Trimesh(vertices, faces)` (process=True:
`merge_vertices()` once at construction), `merge_vertices()`, `update_faces(unique_faces())`, `fix_normals()`,
`visual.face_colors = (255, 165, 0, 255)`, `export(.obj)`.  trimesh is not part of this image, so nothing here can be run
against it; the steps below restate what trimesh 4.2.3 publishes for each call, and tests/test_mesh_export.py holds
hand-derived expectations (file text included) for them:

  merge_vertices   grouping.merge_vertices: rows of round(vertices * 1e8) (tol.merge = 1e-8) that are equal become one vertex;
                   `unique_rows(..., keep_order=True)` keeps the merged vertices in the order of their FIRST OCCURRENCE.  The
                   detokenizer emits multiples of 1/128 in [-0.5, 0.5], which differ by >= 7.8e-3 or not at all, so "equal after
                   rounding to 8 decimals" is exact equality of the float32 values here.
  unique_faces     rows of np.sort(faces, axis=1): the first face of every vertex SET stays, in the original face order.
                   Degenerate faces (a repeated vertex) stay: the constructor runs process(validate=False).
  fix_normals      repair.fix_winding + repair.fix_inversion, `multibody=None` resolved to `body_count > 1` where body_count
                   counts the connected components of the VERTEX graph (every edge of every face):
                   * fix_winding: breadth-first over the face-adjacency graph (pairs of faces sharing an edge that exactly two
                     faces use; a face is never adjacent to itself); the second face of a traversed pair is reversed ([::-1]) when
                     both run along the shared edge in the same direction.
                   * fix_inversion, multibody: groups = connected components of the face-adjacency graph (faces with no manifold
                     edge belong to no group); exactly one group -> the WHOLE mesh is reversed when its total signed volume is
                     negative; several -> every group whose own signed volume is negative is reversed (np.fliplr).
                   * fix_inversion, one body: the whole mesh is reversed when its total signed volume is negative.
                   For an orientable body the result does not depend on where the traversal starts (consistent winding is unique
                   up to a flip of the body, and the volume sign settles the flip).  Where a body is NOT orientable the result
                   depends on the traversal order -- lowest face index first here, networkx's node order in trimesh: not
                   reproduced, and said so.
  export(.obj)     exchange/obj.export_obj: a `# <header>` line, `v x y z r g b` lines ('{:.8f}'; face colours become vertex colours,
                   one colour on every face gives every vertex that colour, written as uint8 / 255), `f a b c` lines (1-based), no
                   normals (none cached), no texture.  The header comment is this package's, not trimesh's URL.
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
    # merge_vertices on exact coordinates, merged vertices in first-occurrence order (unique_rows(..., keep_order=True))
    verts, first_v, inverse = np.unique(tri, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first_v, kind="stable")
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    verts = verts[order]
    faces = rank[np.asarray(inverse).reshape(-1)].reshape(-1, 3).astype(np.int64)
    key = np.sort(faces, axis=1)
    _, first = np.unique(key, axis=0, return_index=True)                     # unique_faces: first occurrence of each vertex set
    faces = faces[np.sort(first)]
    return verts.astype(np.float32), faces


def _vertex_body_count(n_verts: int, faces: np.ndarray) -> int:
    """Trimesh.body_count: connected components of the vertex graph whose edges are the edges of every face (union-find)."""
    parent = np.arange(n_verts)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for a, b, c in faces:
        ra, rb, rc = find(a), find(b), find(c)
        parent[rb] = ra
        parent[find(rc)] = ra
    return len({find(v) for v in range(n_verts)})


def fix_normals(verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Trimesh.fix_normals() of trimesh 4.2.3 (module docstring): consistent winding across manifold edges, then outward
    orientation by signed volume -- per face-adjacency group when the vertex graph has several components, else as a whole.
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
        if len(fs) == 2 and fs[0] != fs[1]:                  # manifold edges only, never a face with itself (graph.face_adjacency)
            adj[fs[0]].append((fs[1], u, v))
            adj[fs[1]].append((fs[0], u, v))

    def directed(f, u, v):                                   # does face f traverse the edge as u -> v ?
        a, b, c = faces[f]
        return (a == u and b == v) or (b == u and c == v) or (c == u and a == v)

    seen = np.zeros(n, dtype=bool)
    groups = []                                              # face-adjacency components: faces with at least one manifold edge
    for root in range(n):
        if seen[root] or not adj[root]:
            continue
        seen[root] = True
        queue = [root]
        body = [root]
        groups.append(body)
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
    # 6 x the face's term of the volume integral trimesh evaluates (triangles.mass_properties, integral[0], used by Trimesh.volume and by
    # fix_inversion's per-group volume): cross_x * (x0 + x1 + x2) with cross = (v1 - v0) x (v2 - v0).  On a CLOSED surface the sum equals
    # the sum of signed tetrahedron volumes v0 . (v1 x v2); on an open one (MeshAnything's outputs usually are) the two differ -- e.g. a
    # flat patch parallel to the xy plane has cross_x = 0 and is never reversed, wherever it lies.
    tri = np.asarray(verts)[faces].astype(np.float64)
    signed6 = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])[:, 0] * tri[:, :, 0].sum(axis=1)
    multibody = _vertex_body_count(len(verts), faces) > 1
    if multibody and len(groups) != 1:
        for body in groups:                                  # (no group at all: nothing to orient)
            if signed6[body].sum() < 0:
                faces[body] = faces[body][:, ::-1]
    elif signed6.sum() < 0:                                  # one body, or one group: Trimesh.invert() of the whole mesh
        faces = faces[:, ::-1].copy()
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
