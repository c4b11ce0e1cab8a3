"""CPU: host-side logic -- the C-ABI library loads and exports every symbol include/mi355pt.h declares, argument validation and
error strings (no compute calls without a GPU), scene generators, tile sharding."""
import ctypes as C
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(mi):
    mi.build()
    hdr = open(os.path.join(ROOT, "include", "mi355pt.h")).read() + open(os.path.join(ROOT, "include", "mi355pt_host.h")).read()
    declared = set(re.findall(r"\b(mi_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = C.CDLL(mi.api.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert declared == set(mi.api.EXPORTS) | set(mi.api.HOST_EXPORTS)


def test_argument_validation_without_gpu(mi):
    L = mi.lib(); h = C.c_void_p()
    L.check(L.L.mi_scene_create(C.byref(h)))
    sc = mi.scenes.cornell_box(16, 9, 1)
    shapes = (mi.api.MiShape * 1)(mi.api.MiShape(0, 999, 0, 4, 0, -1, 1, 0))
    rc = L.L.mi_scene_set_triangles(h, sc.pos.ctypes.data, None, None, sc.idx.ctypes.data, len(sc.pos), len(sc.idx), C.cast(shapes, C.c_void_p), 1)
    assert rc == 1 and b"shape range" in L.L.mi_last_error()
    bad_idx = sc.idx.copy(); bad_idx[0, 0] = 10 ** 6
    shapes[0].tri_count = 2
    rc = L.L.mi_scene_set_triangles(h, sc.pos.ctypes.data, None, None, bad_idx.ctypes.data, len(sc.pos), len(sc.idx), C.cast(shapes, C.c_void_p), 1)
    assert rc == 1 and b"index out of range" in L.L.mi_last_error()
    assert L.L.mi_scene_set_film(h, 0, 10, 0, 0.5, 0.5) == 1
    assert L.L.mi_scene_set_materials(h, None, 0) == 1
    mats = (mi.api.MiMaterial * 1)(mi.api.MiMaterial(99, 0, 0, 0.1))
    assert L.L.mi_scene_set_materials(h, C.cast(mats, C.c_void_p), 1) == 3          # MI_ERR_UNSUPPORTED
    assert L.L.mi_scene_commit(h, 0) == 1 and b"must be set first" in L.L.mi_last_error()
    assert L.L.mi_scene_set_envmap(h, None, 0, 0, None, 1.0) == 1 and b"mi_scene_set_envmap" in L.L.mi_last_error()
    ems = (mi.api.MiEmitter * 2)(mi.api.MiEmitter(1, -1), mi.api.MiEmitter(1, -1))
    assert L.L.mi_scene_set_emitters(h, C.cast(ems, C.c_void_p), 2) == 1 and b"only contain one environment emitter" in L.L.mi_last_error()
    L.L.mi_scene_destroy(h)
    with pytest.raises(mi.MiError):
        mi.api.Lib("/nonexistent/libmi355pt.so")


def test_round3_material_validation_without_gpu(mi):
    """mi_scene_set_materials on the adapter records added in round 3: a coating nests ONE plain reflective record and needs two different indices of refraction
    (coating.cpp:119-121); a blendbsdf blends two plain, untextured records; neither may be the child of a mixturebsdf; anisotropic `ward` passes the material
    check (the texture-coordinate requirement is a commit-time check)."""
    L = mi.lib(); h = C.c_void_p(); L.check(L.L.mi_scene_create(C.byref(h))); M = mi.api.MiMaterial
    def mat(t, flags=0, distr=0, alpha=0.1, refl=(0.5, 0.5, 0.5), eta=(0, 0, 0), k=(0, 0, 0), spec=(1, 1, 1)):
        return M(t, flags, distr, alpha, (C.c_float * 3)(*refl), (C.c_float * 3)(*eta), (C.c_float * 3)(*k), (C.c_float * 3)(*spec))
    def rc(*ms):
        arr = (M * len(ms))(*ms); return L.L.mi_scene_set_materials(h, C.cast(arr, C.c_void_p), len(ms)), L.L.mi_last_error()
    COAT, BLEND, MIX, DIEL, WARD = 17, 18, 10, 3, 16
    assert rc(mat(0), mat(COAT, distr=0, eta=(1.5, 0, 0)))[0] == 0
    r, msg = rc(mat(0), mat(COAT, distr=0, eta=(1.0, 0, 0))); assert r == 1 and b"must be positive and differ" in msg
    r, msg = rc(mat(0), mat(COAT, distr=5, eta=(1.5, 0, 0))); assert r == 3 and b"coating nests a plain BSDF" in msg
    r, msg = rc(mat(DIEL, eta=(1.5, 0, 0)), mat(COAT, distr=0, eta=(1.5, 0, 0))); assert r == 3 and b"reflective" in msg
    r, msg = rc(mat(0, flags=1), mat(COAT, distr=0, eta=(1.5, 0, 0))); assert r == 3 and b"twosided" in msg
    r, msg = rc(mat(0), mat(COAT, distr=0, eta=(1.5, 0, 0)), mat(COAT, distr=1, eta=(1.3, 0, 0))); assert r == 3          # a coating over a coating
    assert rc(mat(0), mat(0), mat(BLEND, eta=(0, 1, 0)))[0] == 0
    r, msg = rc(mat(0), mat(0, flags=1 << 8), mat(BLEND, eta=(0, 1, 0))); assert r == 3 and b"textures on the BSDFs inside a blendbsdf" in msg
    r, msg = rc(mat(0), mat(0), mat(BLEND, eta=(0, 1, 0)), mat(BLEND, eta=(0, 2, 0))); assert r == 3 and b"plain BSDF records" in msg
    r, msg = rc(mat(0), mat(COAT, distr=0, eta=(1.5, 0, 0)), mat(MIX, distr=2, refl=(0, 1, 0), k=(0.5, 0.5, 0))); assert r == 3 and b"children of a mixturebsdf" in msg
    assert rc(mat(WARD, flags=8, distr=2, alpha=0.1, k=(0.3, 0.4, 0), spec=(0.2, 0.2, 0.2)))[0] == 0
    RCOAT, COND = 19, 2
    r, msg = rc(mat(COND), mat(RCOAT, distr=0, eta=(1.5, 1.0, 0.0), k=(0, 0, 100))); assert r == 3 and b"without a Dirac delta lobe" in msg
    assert rc(mat(0), mat(RCOAT, distr=0, eta=(1.5, 1.0, 1.0), k=(0, 0, 100)))[0] == 0          # (its transmittance slice is checked at commit)
    r, msg = rc(mat(0), mat(RCOAT, distr=0, eta=(1.5, 1.0, 3.0), k=(0, 0, 100))); assert r == 1 and b"invalid distribution" in msg
    assert rc(mat(20))[0] == 3
    L.L.mi_scene_destroy(h)


def test_scene_generators(mi):
    S = mi.scenes
    sc = S.cornell_box()
    assert len(sc.idx) == 32 and sc.width == 1920 and sc.height == 1080 and len(sc.emitters) == 1
    # consistent winding: wall normals face the room centre, block normals face away from the block centres
    c = np.array([278, 273, 280], np.float32)
    for t in range(12):
        p0, p1, p2 = sc.pos[sc.idx[t]]; n = np.cross(p1 - p0, p2 - p0)
        assert np.dot(n, c - (p0 + p1 + p2) / 3) > 0
    for si in (6, 7):
        s = sc.shapes[si]; verts = sc.pos[s["first_vert"]:s["first_vert"] + s["vert_count"]]; centre = verts.mean(0)
        for t in range(s["first_tri"], s["first_tri"] + s["tri_count"]):
            p0, p1, p2 = sc.pos[sc.idx[t]]; n = np.cross(p1 - p0, p2 - p0)
            assert np.dot(n, (p0 + p1 + p2) / 3 - centre) > 0
    # camera matrix: sample (0.5, 0.5) maps to the optical axis
    m = sc.sample_to_camera.astype(np.float64); p = m @ np.array([0.5, 0.5, 0, 1.0]); p = p[:3] / p[3]
    assert abs(p[0]) < 1e-6 and abs(p[1]) < 1e-6 and abs(p[2] - sc.near) < 1e-4
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".miscene") as f:
        S.save_scene(sc, f.name); assert os.path.getsize(f.name) > 1000


def test_tile_sharding(mi):
    d = importlib_dist(mi)
    for h, world in [(1080, 1), (1080, 2), (1080, 8), (7, 8), (2160, 5)]:
        bands = [d.shard_rows(h, r, world) for r in range(world)]
        assert bands[0][0] == 0 and bands[-1][1] == h
        assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in bands]; assert max(sizes) - min(sizes) <= 1


def importlib_dist(mi):
    import importlib
    return importlib.import_module("mitsuba-im_amd.dist")
