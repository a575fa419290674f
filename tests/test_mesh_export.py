import numpy as np

from meshanything_amd.mesh_export import faces_from_coords, fix_normals, write_obj


def test_faces_from_coords_merges_and_dedups(tmp_path):
    c = np.full((6, 3, 3), np.nan, dtype=np.float32)
    a, b, d, e = [0, 0, 0], [0.5, 0, 0], [0, 0.5, 0], [0, 0, 0.5]
    c[0] = [a, b, d]
    c[2] = [b, d, e]          # shares an edge with face 0
    c[3] = [d, a, b]          # same vertex set as face 0, rotated: dropped (unique_faces)
    c[5] = [a, e, b]
    v, f = faces_from_coords(c)
    assert v.shape == (4, 3) and f.shape == (3, 3)
    tri = v[f]
    assert np.array_equal(tri[0], np.array([a, b, d], np.float32)) and np.array_equal(tri[2], np.array([a, e, b], np.float32))
    p = tmp_path / "m.obj"
    write_obj(str(p), v, f)
    lines = p.read_text().splitlines()
    assert sum(l.startswith("v ") for l in lines) == 4 and sum(l.startswith("f ") for l in lines) == 3
    # main.py:170-173: one colour (255, 165, 0) on every face -> carried as vertex colours on the `v` lines
    vl = [l.split() for l in lines if l.startswith("v ")]
    assert all(len(t) == 7 and [float(x) for x in t[4:]] == [1.0, round(165 / 255, 8), 0.0] for t in vl)
    write_obj(str(p), v, f, color=None)
    assert all(len(l.split()) == 4 for l in p.read_text().splitlines() if l.startswith("v "))
    v0, f0 = faces_from_coords(np.full((4, 3, 3), np.nan, np.float32))
    assert v0.shape == (0, 3) and f0.shape == (0, 3)


def _signed_volume(v, f):
    t = v[f].astype(np.float64)
    return np.einsum("ij,ij->i", t[:, 0], np.cross(t[:, 1], t[:, 2])).sum() / 6.0


def test_fix_normals_makes_a_scrambled_tetrahedron_consistent_and_outward():
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    good = np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]])          # outward: volume +1/6
    assert abs(_signed_volume(v, good) - 1 / 6) < 1e-9
    scrambled = good.copy()
    scrambled[1] = scrambled[1][::-1]
    scrambled[3] = scrambled[3][::-1]
    fixed = fix_normals(v, scrambled)
    assert abs(_signed_volume(v, fixed) - 1 / 6) < 1e-9
    # every shared edge is traversed in opposite directions by its two faces
    dirs = {}
    for a, b, c in fixed:
        for e in ((a, b), (b, c), (c, a)):
            dirs[e] = dirs.get(e, 0) + 1
    assert all((b, a) in dirs for (a, b) in dirs) and all(n == 1 for n in dirs.values())
    inward = good[:, ::-1]
    assert abs(_signed_volume(v, fix_normals(v, inward)) - 1 / 6) < 1e-9    # all faces inverted: flipped as a whole
    assert fix_normals(v, np.zeros((0, 3), np.int64)).shape == (0, 3)
    two = np.array([[0, 1, 2], [0, 1, 3]])                                   # open surface: made consistent, orientation by volume sign
    f2 = fix_normals(v, two)
    e01 = [(f[i], f[(i + 1) % 3]) for f in f2 for i in range(3) if {f[i], f[(i + 1) % 3]} == {0, 1}]
    assert len(e01) == 2 and e01[0] == e01[1][::-1]


def test_fix_normals_orients_every_body_on_its_own():
    """Two tetrahedra far apart, one wound outward and one inward: trimesh's fix_normals resolves multibody to
    body_count > 1 and fixes each connected body by its own signed volume; a single global flip would leave one inverted."""
    v1 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    good = np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]])
    v = np.concatenate([v1, v1 * 3 + 10])                        # the second body is 27x larger: it dominates the total volume
    faces = np.concatenate([good[:, ::-1], good + 4])            # small body inverted, big body fine: total volume > 0
    fixed = fix_normals(v, faces)
    assert abs(_signed_volume(v, fixed[:4]) - 1 / 6) < 1e-9
    assert _signed_volume(v, fixed[4:]) > 0
    faces = np.concatenate([good, (good + 4)[:, ::-1]])          # big body inverted: total volume < 0, the small one must stay
    fixed = fix_normals(v, faces)
    assert abs(_signed_volume(v, fixed[:4]) - 1 / 6) < 1e-9 and np.array_equal(fixed[:4], good)
    assert _signed_volume(v, fixed[4:]) > 0
