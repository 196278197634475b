"""Host side of nerf_sh/gen_mesh.py: the marching-cubes replacement for PyMCubes (reference gen_mesh.py:124), the grid
construction (:105-111) and the OBJ writer (:133-158).  No GPU: `sigma_grid` is driven by an analytic torch field."""
import numpy as np
import torch

from plenoctree_amd.nerf_sh import gen_mesh, isosurface


def _directed_edges(f, nv):
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    return e[:, 0] * (nv + 1) + e[:, 1], e[:, 1] * (nv + 1) + e[:, 0]


def _grid(n):
    return np.stack(np.meshgrid(*[np.linspace(-1, 1, n)] * 3, indexing="ij"), -1)


def test_case_table_is_complete_and_complementary():
    t, c = isosurface.TRI_TABLE, isosurface.TRI_COUNT
    assert c[0] == 0 and c[255] == 0 and (c[1:255] > 0).all()
    assert t.shape[1] == c.max() == 5
    for case in range(256):
        used = t[case, :c[case]]
        assert (used >= 0).all() and (t[case, c[case]:] == -1).all()
        # every edge whose end corners differ in sign is used, and no other edge
        active = {e for e, (a, o) in enumerate(isosurface._EDGES) if ((case >> o) & 1) != ((case >> (o | (1 << a))) & 1)}
        assert set(used.reshape(-1).tolist()) == active


def test_sphere_is_closed_outward_and_on_the_level_set():
    n, R = 48, 0.6
    g = _grid(n)
    v, f = isosurface.marching_cubes(R - np.linalg.norm(g, axis=-1), 0.0)
    key, rkey = _directed_edges(f, len(v))
    assert np.unique(key).size == key.size                       # manifold: each directed edge once ...
    assert np.array_equal(np.sort(key), np.sort(rkey))           # ... and its opposite exactly once: closed
    assert len(v) - key.size // 2 + len(f) == 2                  # Euler characteristic of a sphere
    p = v / (n - 1) * 2 - 1
    assert np.abs(np.linalg.norm(p, axis=1) - R).max() < 1e-3    # linear interpolation error, O(h^2)
    tri = p[f]
    normal = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    assert ((normal * tri.mean(1)).sum(1) > 0).all()             # normals point to lower values (out of the solid)
    vol = np.einsum("ij,ij->i", tri[:, 0], np.cross(tri[:, 1], tri[:, 2])).sum() / 6
    assert abs(vol - 4 / 3 * np.pi * R ** 3) < 5e-3


def test_torus_genus_and_noise_field_has_no_boundary():
    g = _grid(56)
    tor = 0.25 - np.sqrt((np.sqrt(g[..., 0] ** 2 + g[..., 1] ** 2) - 0.6) ** 2 + g[..., 2] ** 2)
    v, f = isosurface.marching_cubes(tor, 0.0)
    key, rkey = _directed_edges(f, len(v))
    assert np.array_equal(np.sort(key), np.sort(rkey)) and len(v) - np.unique(key).size // 2 + len(f) == 0
    # white noise exercises every case including the ambiguous faces; padded with "outside" so nothing reaches the border
    rng = np.random.default_rng(0)
    vol = rng.standard_normal((30, 23, 17)).astype(np.float32)
    vol[[0, -1]] = -5; vol[:, [0, -1]] = -5; vol[:, :, [0, -1]] = -5
    v, f = isosurface.marching_cubes(vol, 0.1)
    assert set(np.unique(isosurface.TRI_COUNT)) >= {0, 1, 2, 3, 4}
    key, rkey = _directed_edges(f, len(v))
    assert np.array_equal(np.sort(key), np.sort(rkey))           # the triangle set is a cycle: no holes
    assert (f[:, 0] != f[:, 1]).all() and (f[:, 1] != f[:, 2]).all() and (f[:, 0] != f[:, 2]).all()
    # one vertex per sign-changing grid edge
    inside = vol >= 0.1
    n_cross = sum(int((np.diff(inside.astype(np.int8), axis=a) != 0).sum()) for a in range(3))
    assert len(v) == n_cross


def test_edge_cases():
    v, f = isosurface.marching_cubes(np.zeros((4, 4, 4)), 1.0)   # nothing inside
    assert v.shape == (0, 3) and f.shape == (0, 3)
    v, f = isosurface.marching_cubes(np.ones((4, 4, 4)), 1.0)    # value == iso counts as inside: everything inside
    assert len(v) == 0 and len(f) == 0
    v, f = isosurface.marching_cubes(np.zeros((1, 4, 4)), 0.5)
    assert len(v) == 0 and len(f) == 0
    vol = np.zeros((3, 3, 3)); vol[1, 1, 1] = 2.0                # one inside sample: an octahedron at t = 0.5
    v, f = isosurface.marching_cubes(vol, 1.0)
    assert len(v) == 6 and len(f) == 8
    assert np.allclose(np.sort(np.abs(v - 1.0).sum(1)), 0.5)


def test_sigma_grid_matches_reference_point_list_and_mesh_scaling(tmp_path):
    c1, c2, reso = [-2.0, -1.0, -1.5], [2.0, 1.0, 1.5], [21, 17, 13]
    field = lambda p: (1.0 - p.norm(dim=-1, keepdim=True)) * 3.0
    sig = gen_mesh.sigma_grid(field, c1, c2, reso, 1000, torch.device("cpu"))
    grid = np.vstack(np.meshgrid(*(np.linspace(lo, hi, sz, dtype=np.float32) for lo, hi, sz in zip(c1, c2, reso)),
                                 indexing="ij")).reshape(3, -1).T                     # gen_mesh.py:105-111
    assert torch.equal(sig, field(torch.from_numpy(np.ascontiguousarray(grid))).reshape(*reso))
    verts, faces = gen_mesh.marching_cubes(field, c1, c2, reso, 0.0, 1000, torch.device("cpu"))
    # reference scaling (:127): index units * (c2 - c1) / reso + c1
    vi, fi = isosurface.marching_cubes(sig.numpy(), 0.0)
    assert np.array_equal(faces, fi)
    assert np.allclose(verts, vi * ((np.array(c2) - np.array(c1)) / np.array(reso)) + np.array(c1))
    path = str(tmp_path / "m.obj")
    gen_mesh.save_obj(verts, faces, path)
    lines = open(path).read().splitlines()
    assert lines[0] == "v %.4f %.4f %.4f" % tuple(verts[0])
    assert lines[len(verts)] == "f %d %d %d" % tuple(faces[0] + 1) and len(lines) == len(verts) + len(faces)
    gen_mesh.save_obj(verts[:2], faces[:0], path, vert_rgb=np.array([[1, 0, 0.5], [0, 1, 0.25]]))
    assert open(path).read().splitlines()[1].endswith("0.0000 1.0000 0.2500")
    args = gen_mesh.define_flags().parse_args([])
    assert (args.reso, args.c1, args.c2, args.iso, args.coarse, args.point_chunk) == \
        ("300 300 300", "-2 -2 -2", "2 2 2", 6.0, False, 720720)                       # gen_mesh.py:49-76
    assert gen_mesh._triple("5", int) == [5, 5, 5] and gen_mesh._triple("1 2 3", float) == [1.0, 2.0, 3.0]


def _reference_function(path, name):
    """Source of one top-level function of a reference file, compiled on its own (the file's imports - jax, flax,
    mcubes - are not needed by the functions taken this way)."""
    import ast
    import os
    import pytest
    if not os.path.exists(path):
        pytest.skip("needs /root/reference")
    src = open(path).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"np": np}
    exec(compile(ast.Module([fn], []), path, "exec"), ns)
    return ns[name]


def test_save_obj_and_grid_match_the_references_own_code(tmp_path):
    """nerf_sh/gen_mesh.py's own `save_obj` (:133-158) run next to ours on the same mesh: byte-identical files, with and
    without vertex colours; and the reference's grid construction + vertex rescaling (:105-111, :126-129, executed with the
    sigma evaluation and PyMCubes replaced by stand-ins) gives the same vertices as `gen_mesh.marching_cubes`."""
    ref_save = _reference_function("/root/reference/nerf_sh/gen_mesh.py", "save_obj")
    rng = np.random.default_rng(3)
    verts = rng.normal(size=(57, 3)) * 3.0
    faces = rng.integers(0, 57, size=(101, 3))
    cols = rng.random(size=(57, 3))
    for c in (None, cols):
        a, b = str(tmp_path / "ref.obj"), str(tmp_path / "ours.obj")
        ref_save(verts, faces, a, vert_rgb=c)
        gen_mesh.save_obj(verts, faces, b, vert_rgb=c)
        assert open(a, "rb").read() == open(b, "rb").read()
    # marching_cubes(): the reference's body with its three externals stubbed
    import types
    ref_mc = _reference_function("/root/reference/nerf_sh/gen_mesh.py", "marching_cubes")
    field = lambda p: (1.0 - p.norm(dim=-1, keepdim=True)) * 3.0
    c1, c2, reso = [-2.0, -1.0, -1.5], [2.0, 1.0, 1.5], [21, 17, 13]
    g = ref_mc.__globals__
    g["h0print"] = lambda *a, **k: None
    g["utils"] = types.SimpleNamespace(eval_points=lambda fn, grid, chunk: (None, fn(torch.from_numpy(np.ascontiguousarray(grid))).numpy()))
    g["jax"] = types.SimpleNamespace(host_id=lambda: 0)
    g["mcubes"] = types.SimpleNamespace(marching_cubes=lambda vol, iso: isosurface.marching_cubes(vol, iso))
    v_ref, f_ref = ref_mc(field, c1, c2, reso, 0.0, 1000)
    v_our, f_our = gen_mesh.marching_cubes(field, c1, c2, reso, 0.0, 1000, torch.device("cpu"))
    assert np.array_equal(f_ref, f_our) and np.allclose(v_ref, v_our, rtol=0, atol=1e-12)
