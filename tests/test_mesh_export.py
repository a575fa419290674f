import numpy as np

from meshanything_amd.mesh_export import faces_from_coords, write_obj


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
    v0, f0 = faces_from_coords(np.full((4, 3, 3), np.nan, np.float32))
    assert v0.shape == (0, 3) and f0.shape == (0, 3)
