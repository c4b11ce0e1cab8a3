"""Mesh readers (mitsuba-im_amd/meshio.py) against the reference's own loaders: fixtures tests/golden/mesh_*.npz were dumped from the
reference's obj / cube plugins and TriMesh::loadCompressed by tests/golden/make_mesh_golden.py (src/shapes/obj.cpp, cube.cpp,
src/librender/trimesh.cpp); bunny.ply is the asset the reference's own tests hold (data/tests/bunny.ply)."""
import importlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
MESHES = os.path.join(GOLD, "meshes")
meshio = importlib.import_module("mitsuba-im_amd.meshio")

NORMAL_TOL = 2e-6      # generated / transformed normals: float32 asin / sqrt / divide differ by an ulp between libm and numpy


def check(meshes, g, exact_normals=False):
    assert len(meshes) == int(g["n_meshes"])
    for i, m in enumerate(meshes):
        if f"name{i}" in g:
            assert m.name == str(g[f"name{i}"]) or str(g[f"name{i}"]) == "", (m.name, str(g[f"name{i}"]))
        np.testing.assert_array_equal(m.triangles, g[f"triangles{i}"])
        np.testing.assert_allclose(m.positions, g[f"positions{i}"], rtol=0, atol=1e-6)
        assert (m.normals is not None) == (f"normals{i}" in g.files)
        assert (m.uv is not None) == (f"uv{i}" in g.files)
        if m.normals is not None:
            np.testing.assert_allclose(m.normals, g[f"normals{i}"], rtol=0, atol=0 if exact_normals else NORMAL_TOL)
        if m.uv is not None:
            np.testing.assert_array_equal(m.uv, g[f"uv{i}"])


@pytest.mark.parametrize("variant,kw", [
    ("plain", {}), ("xf", {"xf": True}), ("noflipuv_facen", {"flip_tex_coords": False, "face_normals": True}),
    ("flipn", {"flip_normals": True, "xf": True}), ("shape2", {"shape_index": 2})])
def test_obj_loader_matches_reference_plugin(variant, kw):
    g = np.load(os.path.join(GOLD, f"mesh_obj_{variant}.npz"))
    kw = dict(kw)
    tw = g["to_world"] if kw.pop("xf", False) else None
    meshes = meshio.load_obj(os.path.join(MESHES, "statements.obj"), to_world=tw, **kw)
    check(meshes, g)
    if variant == "plain":      # untransformed positions are the parsed decimals: bit-exact
        for i, m in enumerate(meshes):
            np.testing.assert_array_equal(m.positions, g[f"positions{i}"])
        assert [m.material for m in meshes] == ["red", "red", "green", "green"][:len(meshes)]


def test_obj_errors_like_the_reference(tmp_path):
    p = tmp_path / "bad.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 0\n")
    with pytest.raises(meshio.MeshError, match="Out of bounds: tried to access vertex 0"):      # src/shapes/obj.cpp:649
        meshio.load_obj(str(p))
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n")
    with pytest.raises(meshio.MeshError, match="vertex 7"):
        meshio.load_obj(str(p))
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1/1/1 2 3\n")
    with pytest.raises(meshio.MeshError, match="Invalid OBJ face format"):                     # src/shapes/obj.cpp:391
        meshio.load_obj(str(p))
    with pytest.raises(meshio.MeshError, match="not found"):
        meshio.load_obj(str(tmp_path / "missing.obj"))
    p.write_text("# nothing\n")
    assert meshio.load_obj(str(p)) == []


def test_cube_matches_reference_plugin():
    g = np.load(os.path.join(GOLD, "mesh_cube.npz"))
    tw = g["to_world"]
    meshes = meshio.make_cube(tw) + meshio.make_cube(tw, flip_normals=True) + meshio.make_cube(tw, face_normals=True, flip_normals=True)
    check(meshes, g)


def test_serialized_reader_and_writer():
    g = np.load(os.path.join(GOLD, "mesh_serialized.npz"))
    two = os.path.join(MESHES, "two_meshes.serialized")          # written by save_serialized, read back by the reference's reader
    ref_written = os.path.join(MESHES, "statements_mesh0_reference.serialized")     # written by TriMesh::serialize
    meshes = meshio.load_serialized(two, 1) + meshio.load_serialized(two, 0) + meshio.load_serialized(ref_written, 0)
    check(meshes, g, exact_normals=True)
    # the reference-written file holds mesh 0 of the transformed OBJ
    ours = meshio.load_obj(os.path.join(MESHES, "statements.obj"), to_world=np.load(os.path.join(GOLD, "mesh_obj_xf.npz"))["to_world"])[0]
    np.testing.assert_array_equal(meshes[2].triangles, ours.triangles)
    np.testing.assert_allclose(meshes[2].positions, ours.positions, atol=1e-6)
    with pytest.raises(meshio.MeshError, match="out of range"):
        meshio.load_serialized(two, 2)
    with pytest.raises(meshio.MeshError, match="nonnegative"):
        meshio.load_serialized(two, -1)


def test_serialized_roundtrip_and_plugin_transform(tmp_path):
    ours = meshio.load_obj(os.path.join(MESHES, "statements.obj"))
    p = str(tmp_path / "all.serialized")
    meshio.save_serialized(p, ours)
    for i, m in enumerate(ours):
        back = meshio.load_serialized(p, i)[0]
        np.testing.assert_array_equal(back.positions, m.positions); np.testing.assert_array_equal(back.triangles, m.triangles)
        np.testing.assert_array_equal(back.normals, m.normals)
        assert (back.uv is None) == (m.uv is None) and back.name == m.name
    # src/shapes/serialized.cpp:188-203: a mirroring toWorld swaps the winding so geometric normals keep pointing outwards
    mirror = np.diag([-1.0, 1.0, 1.0, 1.0]).astype(np.float32)
    mm = meshio.load_serialized(p, 0, to_world=mirror)[0]
    np.testing.assert_array_equal(mm.triangles, ours[0].triangles[:, [1, 0, 2]])
    np.testing.assert_array_equal(mm.positions[:, 0], -ours[0].positions[:, 0])
    (tmp_path / "junk.serialized").write_bytes(b"\x00\x01\x02\x03\x04\x05\x06\x07")
    with pytest.raises(meshio.MeshError, match="invalid file format"):
        meshio.load_serialized(str(tmp_path / "junk.serialized"))


def test_bunny_ply_and_generated_normals(tmp_path):
    """The reference's own test asset (69451 triangles; its PLY plugin is not buildable here): read by our PLY reader, written as OBJ, and
    loaded by both OBJ loaders -- pins the vertex merge at scale (35947 -> 34834 vertices) and TriMesh::computeNormals."""
    g = np.load(os.path.join(GOLD, "mesh_bunny.npz"))
    raw = meshio.load_ply(os.path.join(MESHES, "bunny.ply"))[0]
    assert len(raw.positions) == 35947 and len(raw.triangles) == 69451          # the counts the file's header states
    raw.normals = None
    meshio.save_obj(str(tmp_path / "bunny.obj"), raw)
    m = meshio.load_obj(str(tmp_path / "bunny.obj"))[0]
    assert len(m.positions) == int(g["n_verts"]) and len(m.triangles) == int(g["n_tris"])
    sel = g["sel"]
    np.testing.assert_array_equal(m.positions[sel], g["positions_sel"])
    np.testing.assert_array_equal(m.triangles[::16], g["triangles_sel"])
    np.testing.assert_array_equal(m.triangles.astype(np.int64).sum(0), g["triangles_sum"])
    np.testing.assert_allclose(m.positions.astype(np.float64).sum(0), g["positions_sum"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(m.normals[sel], g["normals_sel"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(m.normals.astype(np.float64).sum(0), g["normals_sum"], rtol=0, atol=2e-2)
    # the PLY mesh itself: same surface, unmerged vertices
    np.testing.assert_array_equal(raw.positions[raw.triangles.reshape(-1)], m.positions[m.triangles.reshape(-1)])


def test_ply_ascii_quads_and_properties(tmp_path):
    p = tmp_path / "q.ply"
    p.write_text("ply\nformat ascii 1.0\ncomment test\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
                 "property float s\nproperty float t\nproperty uchar red\nelement face 2\nproperty list uchar int vertex_indices\nend_header\n"
                 "0 0 0 0 0 2 0 0 255\n1 0 0 0 0 2 1 0 255\n1 1 0 0 0 2 1 1 255\n0 1 0 0 0 2 0 1 255\n4 0 1 2 3\n3 0 2 3\n")
    m = meshio.load_ply(str(p), to_world=np.diag([2.0, 2.0, 2.0, 1.0]).astype(np.float32))[0]
    np.testing.assert_array_equal(m.triangles, [[0, 1, 2], [3, 0, 2], [0, 2, 3]])                # src/shapes/ply.cpp:288-297
    np.testing.assert_array_equal(m.positions[2], [2, 2, 0])
    np.testing.assert_array_equal(m.normals, np.tile([0, 0, 1], (4, 1)))                        # normalised after the transform
    np.testing.assert_array_equal(m.uv, [[0, 0], [1, 0], [1, 1], [0, 1]])
    # binary big endian, ragged face list
    import struct
    hdr = "ply\nformat binary_big_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nelement face 2\nproperty list uchar uint vertex_index\nend_header\n"
    body = b"".join(struct.pack(">3f", *v) for v in [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)]) + struct.pack(">B3I", 3, 0, 1, 2) + struct.pack(">B4I", 4, 0, 1, 2, 3)
    (tmp_path / "b.ply").write_bytes(hdr.encode() + body)
    mb = meshio.load_ply(str(tmp_path / "b.ply"), face_normals=True)[0]
    np.testing.assert_array_equal(mb.triangles, [[0, 1, 2], [0, 1, 2], [3, 0, 2]])
    assert mb.normals is None and mb.face_normals
    (tmp_path / "penta.ply").write_text("ply\nformat ascii 1.0\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n"
                                        "0 0 0\n1 0 0\n1 1 0\n0 1 0\n0 0 1\n5 0 1 2 3 4\n")
    with pytest.raises(meshio.MeshError, match="Only triangle and quad"):
        meshio.load_ply(str(tmp_path / "penta.ply"))


def test_max_smooth_angle_matches_reference(tmp_path):
    """`maxSmoothAngle` = TriMesh::rebuildTopology (trimesh.cpp:468-606): vertex splitting at creases, then regenerated normals -- vertex numbering,
    triangles and positions identical to the reference's, on the OBJ statements file (with texture coordinates) and on the bunny (35947 -> 37862 vertices)."""
    g = np.load(os.path.join(GOLD, "mesh_obj_smooth40.npz"))
    check(meshio.load_obj(os.path.join(MESHES, "statements.obj"), max_smooth_angle=40.0), g)
    raw = meshio.load_ply(os.path.join(MESHES, "bunny.ply"))[0]; raw.normals = None
    meshio.save_obj(str(tmp_path / "bunny.obj"), raw)
    m = meshio.load_obj(str(tmp_path / "bunny.obj"), max_smooth_angle=25.0)[0]
    g = np.load(os.path.join(GOLD, "mesh_bunny_smooth25.npz"))
    assert len(m.positions) == int(g["n_verts"]) == 37862
    np.testing.assert_array_equal(m.triangles[::16], g["triangles_sel"]); np.testing.assert_array_equal(m.triangles.astype(np.int64).sum(0), g["triangles_sum"])
    np.testing.assert_array_equal(m.positions[::16], g["positions_sel"]); np.testing.assert_allclose(m.normals[::16], g["normals_sel"], rtol=0, atol=2e-6)
    # the PLY loader takes the same parameter (same surface: the PLY holds duplicate positions the OBJ path merges first, so only the corner normals are compared)
    mp = meshio.load_ply(os.path.join(MESHES, "bunny.ply"), max_smooth_angle=25.0)[0]
    np.testing.assert_array_equal(mp.positions[mp.triangles.reshape(-1)], m.positions[m.triangles.reshape(-1)])
    with pytest.raises(meshio.MeshError, match="can't be specified at the same time"):
        meshio.load_obj(os.path.join(MESHES, "statements.obj"), max_smooth_angle=40.0, face_normals=True)
