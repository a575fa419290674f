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


# ---- hand-derived expectations from trimesh 4.2.3's published behaviour (mesh_export.py docstring; trimesh itself is not installed here) ----
def _square(z):
    c = np.full((4, 3, 3), np.nan, dtype=np.float32)
    c[0] = [[0, 0, z], [0.5, 0, z], [0, 0.5, z]]             # counter-clockwise seen from +z
    c[2] = [[0.5, 0, z], [0, 0.5, z], [0.5, 0.5, z]]         # clockwise: runs along the shared edge the same way as face 0
    c[3] = [[0, 0.5, z], [0, 0, z], [0.5, 0, z]]             # the vertex set of face 0 again: unique_faces drops it
    return c


COL = " 1.00000000 0.64705882 0.00000000"


def test_obj_text_of_a_hand_derived_case(tmp_path):
    """Vertices in first-occurrence order; face 2 reversed by fix_winding ([::-1] of (2, 3, 4) = (4, 3, 2), 1-based).  trimesh's volume is
    the integral sum(cross_x * (x0 + x1 + x2)) / 6 (triangles.mass_properties): a flat patch parallel to the xy plane has cross_x = 0,
    volume 0, and is NOT reversed whichever side of the origin it lies on (a sum of signed tetrahedron volumes would reverse it at z < 0)."""
    for z, faces_txt in ((0.25, ["f 1 2 3", "f 4 3 2"]), (-0.25, ["f 1 2 3", "f 4 3 2"])):
        v, f = faces_from_coords(_square(z))
        f = fix_normals(v, f)
        p = tmp_path / "sq.obj"
        write_obj(str(p), v, f)
        zt = f"{z:.8f}"
        want = ["# MeshAnything (meshanything_amd)",
                f"v 0.00000000 0.00000000 {zt}{COL}", f"v 0.50000000 0.00000000 {zt}{COL}",
                f"v 0.00000000 0.50000000 {zt}{COL}", f"v 0.50000000 0.50000000 {zt}{COL}"] + faces_txt
        assert p.read_text().splitlines() == want


def test_open_patch_is_oriented_by_the_volume_integral_not_by_tetrahedra():
    """An open patch in the plane x = c (normal along +x or -x): integral term = cross_x * (x0 + x1 + x2) = (+-2 A) * 3 c.  With the
    normal along +x and c < 0 the integral is negative -> the patch is reversed; at c > 0 it stays.  (Hand-derived: two triangles of
    area 1/8 each, cross_x = +1/4, sum x = 3 c -> 6 V = 2 * 3 c / 4.)"""
    for c, flipped in ((0.25, False), (-0.25, True)):
        coords = np.array([[[c, 0, 0], [c, 0.5, 0], [c, 0, 0.5]], [[c, 0.5, 0], [c, 0.5, 0.5], [c, 0, 0.5]]], np.float32)
        v, f = faces_from_coords(coords)
        assert f.tolist() == [[0, 1, 2], [1, 3, 2]]
        got = fix_normals(v, f)
        assert got.tolist() == ([[2, 1, 0], [2, 3, 1]] if flipped else [[0, 1, 2], [1, 3, 2]])


def test_vertices_keep_first_occurrence_order_not_sorted_order():
    c = np.array([[[0.5, 0.5, 0.5], [-0.5, 0, 0], [0, 0.25, 0]], [[0, 0.25, 0], [-0.5, 0, 0], [0.25, -0.5, 0.125]]], np.float32)
    v, f = faces_from_coords(c)
    assert np.array_equal(v, np.array([[0.5, 0.5, 0.5], [-0.5, 0, 0], [0, 0.25, 0], [0.25, -0.5, 0.125]], np.float32))
    assert np.array_equal(f, [[0, 1, 2], [2, 1, 3]])


def test_degenerate_faces_stay_and_never_pair_with_themselves():
    c = np.array([[[0, 0, 0], [0, 0, 0], [0.5, 0, 0]], [[0, 0, 0], [0.5, 0, 0], [0, 0.5, 0]]], np.float32)   # face 0 repeats a vertex
    v, f = faces_from_coords(c)
    assert f.tolist() == [[0, 0, 1], [0, 1, 2]]              # process(validate=False): not removed
    assert fix_normals(v, f).tolist() == [[0, 0, 1], [0, 1, 2]]   # edge (0, 1) is used three times: not manifold, nothing to traverse


def test_one_group_plus_a_loose_face_is_oriented_as_a_whole():
    """Vertex graph with two components (body_count 2 -> multibody) but ONE face-adjacency group: fix_inversion's single-group
    escape reverses the whole mesh by its total volume -- the loose triangle included.  The loose triangle lies in a plane z = const, so
    its term of trimesh's volume integral is cross_x * sum(x) = 0: the tetrahedron's sign alone decides."""
    v1 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    good = np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]])
    v = np.concatenate([v1, np.array([[5, 5, 5], [6, 5, 5], [5, 6, 5]], np.float32)])
    loose = np.array([[4, 5, 6]])                            # cross = (0, 0, 1): contributes 0 whichever way it is wound
    for lf in (loose, loose[:, ::-1]):
        fixed = fix_normals(v, np.concatenate([good[:, ::-1], lf]))             # inverted tetrahedron: 6 V = -1 < 0 -> every face reversed
        assert np.array_equal(fixed, np.concatenate([good, lf[:, ::-1]]))
        fixed = fix_normals(v, np.concatenate([good, lf]))                      # 6 V = +1: nothing reversed
        assert np.array_equal(fixed, np.concatenate([good, lf]))


def test_bodies_that_touch_in_one_vertex_count_as_one_body():
    """Two tetrahedra sharing a single vertex: body_count == 1 (the vertex graph is connected), so multibody is False and the mesh is
    reversed as a whole by its TOTAL volume, although face adjacency sees two groups."""
    v1 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    good = np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]])
    v = np.concatenate([v1, -3 * v1[1:]])                    # second tetrahedron (0, 4, 5, 6): mirrored through the shared origin, 27 x the volume
    big = np.array([[0, 5, 4], [0, 4, 6], [4, 5, 6], [0, 6, 5]])            # the mirror image of `good`'s pattern: inward (V = -27/6)
    assert _signed_volume(v, big) < 0 < _signed_volume(v, good)
    fixed = fix_normals(v, np.concatenate([good, big]))
    assert np.array_equal(fixed, np.concatenate([good, big])[:, ::-1])      # total < 0: both reversed, the small one now inward
