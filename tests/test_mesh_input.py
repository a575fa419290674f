"""Mesh-file inputs (`meshanything_amd/mesh_input.py`; reference: main.py:29-39 -> mesh_to_pc.py:42-57).  trimesh is absent, so there is
no reference-generated fixture for this step ("parity unpinned"); the tests hold the restatement to the properties the path needs:
every sample on the surface, the unit normal of the face under it, density proportional to area, the same geometry through every
supported file format, and the Dataset('mesh', ...) plumbing up to the normalised float16 cloud."""
import struct

import numpy as np
import pytest

from meshanything_amd.data import Dataset
from meshanything_amd.mesh_input import face_normals_and_areas, load_mesh, mesh_to_pc_normal, sample_surface


def box(sx=1.0, sy=2.0, sz=3.0):
    """Axis-aligned box [0,sx] x [0,sy] x [0,sz] as 6 quads with outward orientation."""
    v = np.array([[x, y, z] for x in (0, sx) for y in (0, sy) for z in (0, sz)], dtype=np.float64)
    quads = [[0, 1, 3, 2], [4, 6, 7, 5], [0, 4, 5, 1], [2, 3, 7, 6], [0, 2, 6, 4], [1, 5, 7, 3]]
    return v, quads


def write_obj(path, v, polys, relative=False):
    with open(path, "w") as f:
        f.write("# box\n")
        for p in v:
            f.write(f"v {p[0]} {p[1]} {p[2]}\n")
        f.write("vn 0 0 1\nvt 0 0\n")
        for q in polys:
            idx = [(i - len(v)) if relative else (i + 1) for i in q]
            f.write("f " + " ".join(f"{i}/1/1" for i in idx) + "\n")


def write_ply(path, v, polys, fmt):
    head = f"ply\nformat {fmt} 1.0\ncomment box\nelement vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n" \
           f"element face {len(polys)}\nproperty list uchar int vertex_indices\nend_header\n"
    with open(path, "wb") as f:
        f.write(head.encode())
        if fmt == "ascii":
            for p in v:
                f.write(f"{p[0]} {p[1]} {p[2]} 255\n".encode())
            for q in polys:
                f.write((f"{len(q)} " + " ".join(str(i) for i in q) + "\n").encode())
        else:
            bo = "<" if fmt == "binary_little_endian" else ">"
            for p in v:
                f.write(struct.pack(bo + "fffB", *p, 255))
            for q in polys:
                f.write(struct.pack(bo + "B" + "i" * len(q), len(q), *q))


def write_off(path, v, polys):
    with open(path, "w") as f:
        f.write(f"OFF\n{len(v)} {len(polys)} 0\n")
        for p in v:
            f.write(f"{p[0]} {p[1]} {p[2]}\n")
        for q in polys:
            f.write(f"{len(q)} " + " ".join(str(i) for i in q) + "\n")


def write_stl(path, v, tris, binary):
    n, _ = face_normals_and_areas(v, tris)
    if binary:
        with open(path, "wb") as f:
            f.write(b"\0" * 80 + struct.pack("<I", len(tris)))
            for t, nn in zip(tris, n):
                f.write(struct.pack("<12fH", *nn, *v[t].reshape(-1), 0))
    else:
        with open(path, "w") as f:
            f.write("solid box\n")
            for t, nn in zip(tris, n):
                f.write(f"facet normal {nn[0]} {nn[1]} {nn[2]}\n outer loop\n")
                for p in v[t]:
                    f.write(f"  vertex {p[0]} {p[1]} {p[2]}\n")
                f.write(" endloop\nendfacet\n")
            f.write("endsolid box\n")


def canonical_triangles(v, f):
    """Triangle soup as a sorted array of corner coordinates, rotation of the corners within a triangle removed."""
    t = v[f]
    out = []
    for tri in t:
        rots = [np.roll(tri, -k, axis=0).reshape(-1) for k in range(3)]
        out.append(min(rots, key=lambda r: tuple(r)))
    out = np.array(out)
    return out[np.lexsort(out.T[::-1])]


def test_every_format_gives_the_same_triangles(tmp_path):
    v, quads = box()
    paths = {}
    write_obj(tmp_path / "a.obj", v, quads); paths["obj"] = tmp_path / "a.obj"
    write_obj(tmp_path / "r.obj", v, quads, relative=True); paths["obj-relative"] = tmp_path / "r.obj"
    for fmt in ("ascii", "binary_little_endian", "binary_big_endian"):
        write_ply(tmp_path / f"{fmt}.ply", v, quads, fmt); paths[fmt] = tmp_path / f"{fmt}.ply"
    write_off(tmp_path / "a.off", v, quads); paths["off"] = tmp_path / "a.off"
    ref_v, ref_f = load_mesh(str(paths["obj"]))
    assert ref_v.shape == (8, 3) and ref_f.shape == (12, 3)                       # 6 quads fanned into 12 triangles
    ref = canonical_triangles(ref_v, ref_f)
    for name, p in paths.items():
        vv, ff = load_mesh(str(p))
        assert np.allclose(canonical_triangles(vv, ff), ref, atol=1e-6), name
    for binary in (True, False):
        write_stl(tmp_path / "a.stl", ref_v, ref_f, binary)
        vv, ff = load_mesh(str(tmp_path / "a.stl"))
        assert vv.shape == (8, 3), "STL corners were not merged"
        assert np.allclose(canonical_triangles(vv, ff), ref, atol=1e-6)
    n, a = face_normals_and_areas(ref_v, ref_f)
    assert np.isclose(a.sum(), 2 * (1 * 2 + 2 * 3 + 1 * 3))
    centre = ref_v.mean(0)
    assert ((ref_v[ref_f].mean(1) - centre) * n).sum(1).min() > 0, "the fan kept the outward orientation"


def test_samples_lie_on_the_surface_with_the_normal_of_their_face():
    v, quads = box()
    from meshanything_amd.mesh_input import _fan
    f = _fan(quads)
    np.random.seed(0)
    n_s = 60000
    pts, fi = sample_surface(v, f, n_s)
    normals, areas = face_normals_and_areas(v, f)
    # on the plane of the face it was drawn from, and inside the box
    d = ((pts - v[f[fi, 0]]) * normals[fi]).sum(1)
    assert np.abs(d).max() < 1e-12
    assert (pts >= -1e-12).all() and (pts <= np.array([1, 2, 3]) + 1e-12).all()
    # density proportional to area: the six sides hold area-proportional shares of the samples (5 sigma)
    for axis, size in enumerate((1.0, 2.0, 3.0)):
        for side in (0.0, size):
            on = np.isclose(pts[:, axis], side, atol=1e-9) & (np.abs(normals[fi][:, axis]) > 0.5)
            share = 6.0 / size / 22.0                                   # side area = volume / size = 6 / size; total 22
            sigma = np.sqrt(n_s * share * (1 - share))
            assert abs(on.sum() - n_s * share) < 5 * sigma, (axis, side, on.sum(), n_s * share)
    # uniform inside a face: on the z = 3 side (area 1 x 2) the mean is the centre and the covariance that of a uniform rectangle
    top = np.isclose(pts[:, 2], 3.0, atol=1e-9) & (normals[fi][:, 2] > 0.5)
    m = pts[top][:, :2]
    assert np.allclose(m.mean(0), [0.5, 1.0], atol=0.03)
    assert np.allclose(m.var(0), [1 / 12, 4 / 12], rtol=0.08)
    # the two triangles of a quad get equal shares (equal areas)
    cnt = np.bincount(fi, minlength=len(f))
    assert np.all(np.abs(cnt - n_s * areas / areas.sum()) < 5 * np.sqrt(n_s * areas / areas.sum()))
    # seeded: the same draws again
    np.random.seed(0)
    again, fi2 = sample_surface(v, f, n_s)
    assert np.array_equal(again, pts) and np.array_equal(fi, fi2)


def test_degenerate_faces_are_never_drawn_and_bad_files_are_refused(tmp_path):
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 0, 0]], dtype=np.float64)
    f = np.array([[0, 1, 3], [0, 1, 2]])                                 # the first triangle is a line (area 0)
    np.random.seed(1)
    _, fi = sample_surface(v, f, 1000)
    assert (fi == 1).all()
    pc = mesh_to_pc_normal(v, f, 64)
    assert pc.dtype == np.float16 and pc.shape == (64, 6) and np.allclose(pc[:, 3:], [0, 0, 1])
    with pytest.raises(ValueError):
        sample_surface(v, f[:1], 10)                                     # no area at all
    (tmp_path / "empty.obj").write_text("v 0 0 0\n")
    with pytest.raises(ValueError):
        load_mesh(str(tmp_path / "empty.obj"))
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n")
    with pytest.raises(ValueError):
        load_mesh(str(tmp_path / "bad.obj"))
    with pytest.raises(ValueError):
        load_mesh(str(tmp_path / "mesh.glb"))


def test_dataset_mesh_reaches_the_normalised_cloud(tmp_path):
    """Dataset('mesh', [...]) (main.py:29-39): 4096 surface samples + face normals as float16, then the same normalisation as a
    pc_normal input (main.py:45-58): centred on the bounding box, max |coordinate| = 0.9995, unit normals."""
    v, quads = box(2.0, 1.0, 0.5)
    write_obj(tmp_path / "crate.obj", v + 10.0, quads)
    np.random.seed(0)
    ds = Dataset("mesh", [str(tmp_path / "crate.obj")])
    assert len(ds) == 1 and ds[0]["uid"] == "crate"
    pc = ds[0]["pc_normal"]
    assert pc.dtype == np.float16 and pc.shape == (4096, 6)
    xyz, nrm = pc[:, :3].astype(np.float64), pc[:, 3:].astype(np.float64)
    assert np.isclose(np.abs(xyz).max(), 0.9995, atol=2e-3)
    assert np.allclose(xyz.max(0) + xyz.min(0), 0, atol=1e-2)                   # centred (float16 cloud around 10: coarse grid)
    assert np.allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-3)
    assert np.isin(np.abs(nrm), [0.0, 1.0]).all()                               # axis-aligned faces
    # each normal belongs to the side its point lies on (the extreme coordinate along the normal's axis)
    ax = np.abs(nrm).argmax(1)
    ext = xyz.max(0)
    along = xyz[np.arange(len(xyz)), ax] * nrm[np.arange(len(xyz)), ax]
    assert np.allclose(along, ext[ax], atol=2e-2)
