"""Scene front end (mitsuba-im_amd/xml_scene.py): Mitsuba XML -> flattened scene.  The reference's own loader (src/librender/scenehandler.cpp)
needs xerces-c and cannot be built here, so this layer is checked against (a) the conventions that file states, (b) the synthetic generators
through export_scene -> load_scene round trips, with the oracle rendering both sides, (c) the reference's conversion of spectra
(tests/golden/spectrum_rgb.npz, from oracle/_ref/harness `spectrum`)."""
import importlib
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
MESHES = os.path.join(HERE, "golden", "meshes")
X = importlib.import_module("mitsuba-im_amd.xml_scene")
S = importlib.import_module("mitsuba-im_amd.scenes")

GENERATORS = ["cornell_box", "cbox_shapes", "cbox_materials", "cbox_lights", "cbox_collimated", "cbox_roughdiffuse", "cbox_phong", "cbox_ward", "ward_room", "cbox_coating", "blend_room", "cbox_roughcoating", "open_constant", "cbox_translucent", "cbox_roughplastic", "textured_room",
              "shape_lights", "veach_mis", "veach_microfacets", "textured_plastics", "bitmap_room", "glass_pane", "masked_room", "textured_shapes"]


def assert_same_scene(a, b, exact_analytic=False):
    np.testing.assert_array_equal(a.pos[a.idx.astype(np.int64)], b.pos[b.idx.astype(np.int64)])
    if a.nrm is not None:
        for sa, sb in zip(a.shapes, b.shapes):
            if not sa["face_normals"]:
                ia = a.idx[sa["first_tri"]:sa["first_tri"] + sa["tri_count"]].astype(np.int64); ib = b.idx[sb["first_tri"]:sb["first_tri"] + sb["tri_count"]].astype(np.int64)
                np.testing.assert_allclose(a.nrm[ia], b.nrm[ib], rtol=0, atol=1.5e-7)       # the OBJ reader re-normalises (obj.cpp:658-660)
    for k in ("width", "height", "spp", "sampler", "max_depth", "rr_depth", "strict_normals", "hide_emitters", "filter", "seed"):
        assert a[k] == b[k], k
    assert a.filter_radius == pytest.approx(b.filter_radius) and a.filter_stddev == pytest.approx(b.filter_stddev)
    np.testing.assert_array_equal(a.cam_to_world, b.cam_to_world)
    np.testing.assert_allclose(a.sample_to_camera, b.sample_to_camera, rtol=3e-7, atol=1e-9)      # the file holds the field of view as a float32 (like the reference's m_xfov)
    assert len(a.shapes) == len(b.shapes) and len(a.bsdfs) == len(b.bsdfs) and len(a.emitters) == len(b.emitters)
    for sa, sb in zip(a.shapes, b.shapes):
        for k in ("first_tri", "tri_count", "bsdf", "emitter", "face_normals"):
            assert sa[k] == sb[k], k
    for ba, bb in zip(a.bsdfs, b.bsdfs):
        assert ba.get("aniso", 0) == bb.get("aniso", 0) and ba.get("nonlinear", 0) == bb.get("nonlinear", 0)
        for k in ("type", "twosided", "distr", "texture"):
            assert ba[k] == bb[k], (k, ba, bb)
        assert (ba["sample_visible"] & 1) == (bb["sample_visible"] & 1) or ba["type"] not in (S.BSDF_ROUGHCONDUCTOR, S.BSDF_ROUGHDIELECTRIC)
        for k in ("reflectance", "specular", "eta", "k"):
            if k == "reflectance" and ba.get("texture", -1) >= 0: continue          # the texture replaces it
            np.testing.assert_allclose(ba[k], bb[k], rtol=2e-7, atol=0, err_msg=k)
        assert ba["alpha"] == pytest.approx(bb["alpha"], rel=1e-7)
    for ea, eb in zip(a.emitters, b.emitters):
        assert ea["type"] == eb["type"] and ea["shape"] == eb["shape"]
        np.testing.assert_allclose(ea["radiance"], eb["radiance"], rtol=1e-7)
        if "to_world" in ea:
            np.testing.assert_allclose(ea["to_world"], eb["to_world"], rtol=0, atol=1e-6)
    assert len(a.analytic) == len(b.analytic)
    for xa, xb in zip(a.analytic, b.analytic):
        assert xa["type"] == xb["type"] and xa["bsdf"] == xb["bsdf"] and xa["emitter"] == xb["emitter"] and xa["flags"] == xb["flags"]
        np.testing.assert_allclose(xa["to_world"], xb["to_world"], rtol=0, atol=2e-5 * max(1.0, float(np.abs(xa["to_world"]).max())))
        assert xa["radius"] == pytest.approx(xb["radius"], rel=1e-6) and xa["length"] == pytest.approx(xb["length"], rel=1e-6)
    assert len(a.textures) == len(b.textures)
    for ta, tb in zip(a.textures, b.textures):
        for k in ("type", "color0", "color1", "uoffset", "voffset", "uscale", "vscale", "wrap_u", "wrap_v", "filter", "n_levels"):
            assert ta[k] == pytest.approx(tb[k]), k
    if a.texture_texels is not None:
        np.testing.assert_array_equal(a.texture_levels, b.texture_levels); np.testing.assert_array_equal(a.texture_texels, b.texture_texels)


@pytest.mark.parametrize("gen", GENERATORS)
@pytest.mark.parametrize("fmt", ["serialized", "obj"])
def test_export_load_round_trip(gen, fmt, tmp_path):
    sc = S.veach_mis(width=48, height=32, spp=4, microfacets=S.VEACH_MICROFACETS) if gen == "veach_microfacets" else getattr(S, gen)(width=48, height=32, spp=4)
    path = X.export_scene(sc, str(tmp_path), mesh_format=fmt)
    assert_same_scene(sc, X.load_scene(path))


def test_crop_window_round_trip(tmp_path):
    """Film crop windows (src/librender/film.cpp:35-47, perspective.cpp:129-152): the rendered film is the crop, the camera keeps the full frame's aspect."""
    sc = S.set_crop_window(S.cornell_box(width=64, height=36, spp=4), 192, 108, 70, 40)
    sc2 = X.load_scene(X.export_scene(sc, str(tmp_path)))
    assert sc2.crop == (192, 108, 70, 40) and (sc2.width, sc2.height) == (64, 36)
    np.testing.assert_allclose(sc.sample_to_camera, sc2.sample_to_camera, rtol=3e-7, atol=1e-9)
    full = S.cornell_box(width=192, height=108, spp=4)
    # the crop's pixel (0, 0) looks where the full frame's pixel (70, 40) looks
    import oracle; oracle.build()
    a = np.zeros(8, np.float32); b = np.zeros(8, np.float32); L = oracle.lib()
    L.orc_camera_ray(oracle.Oracle(sc).h, 0.5, 0.5, a.ctypes.data); L.orc_camera_ray(oracle.Oracle(full).h, 70.5, 40.5, b.ctypes.data)
    np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-6)
    with pytest.raises(ValueError, match="Invalid crop window"):
        S.set_crop_window(S.cornell_box(width=64, height=36, spp=4), 100, 108, 70, 40)


def test_round_trip_renders_identically_in_the_oracle(tmp_path):
    """The loaded scene is not just field-equal: the oracle traces the same image from it (Cornell box: bit-identical film)."""
    import oracle
    oracle.build()
    def f32_fov(sc):       # the generators keep the field of view in double; a scene file (and the reference's camera) holds a float32
        sc.xfov = float(np.float32(sc.xfov)); sc.sample_to_camera = S.sample_to_camera(sc.xfov, sc.near, sc.far, sc.width / sc.height); return sc
    sc = f32_fov(S.cornell_box(width=40, height=30, spp=4))
    sc2 = X.load_scene(X.export_scene(sc, str(tmp_path)))
    np.testing.assert_array_equal(sc.sample_to_camera, sc2.sample_to_camera)
    a = oracle.Oracle(sc).render_image(threads=4)[0]; b = oracle.Oracle(sc2).render_image(threads=4)[0]
    np.testing.assert_array_equal(a, b)
    sc = f32_fov(S.cbox_materials(width=40, height=30, spp=4))
    sc2 = X.load_scene(X.export_scene(sc, str(tmp_path), name="m"))
    a = oracle.Oracle(sc).render_image(threads=4)[0]; b = oracle.Oracle(sc2).render_image(threads=4)[0]
    assert np.linalg.norm(a - b) / np.linalg.norm(a) < 2e-2          # analytic transforms are re-derived (sphere scale split): paths may diverge at edges


def test_spectrum_conversion_matches_reference():
    g = np.load(os.path.join(HERE, "golden", "spectrum_rgb.npz"))
    for i in range(int(g["n"])):
        pairs = g[f"pairs{i}"]
        mine = X.spectrum_to_rgb([(w, v) for w, v in pairs])
        np.testing.assert_allclose(mine, g[f"rgb{i}"], rtol=float(g[f"tol{i}"]), atol=1e-5)
    assert X.srgb_to_linear(0.0) == 0.0 and X.srgb_to_linear(1.0) == pytest.approx(1.0, rel=1e-6)
    assert X.srgb_to_linear(0.5) == pytest.approx(0.21404114, rel=1e-5) and X.srgb_to_linear(0.03) == pytest.approx(0.03 / 12.92, rel=1e-6)


def test_bunny_box_scene_file():
    """Hand-written scene around the reference's own test asset: <default>/$name, lookat, fovAxis, spectra, refs, named IORs, PLY."""
    sc = X.load_scene(os.path.join(MESHES, "bunny_box.xml"), params={"spp": 8})
    assert (sc.width, sc.height, sc.spp, sc.max_depth, sc.rr_depth) == (128, 96, 8, 6, 4)
    assert sc.sampler == S.SAMPLER_SOBOL and sc.filter == S.FILTER_BOX
    # fovAxis "smaller" on a 4:3 film = the y axis (sensor.cpp:243-246): xfov = 2 atan(tan(19 deg) * 4/3)
    assert sc.xfov == pytest.approx(math.degrees(2 * math.atan(math.tan(math.radians(19.0)) * 128 / 96)), rel=1e-6)
    assert len(sc.idx) == 69451 and len(sc.pos) == 35947 and sc.nrm is not None and not sc.shapes[0]["face_normals"]
    assert len(sc.analytic) == 7 and [a["type"] for a in sc.analytic] == [S.SHAPE_RECTANGLE] * 5 + [S.SHAPE_SPHERE, S.SHAPE_RECTANGLE]
    assert sc.analytic[5]["radius"] == pytest.approx(0.03) and np.allclose(sc.analytic[5]["to_world"][:3, 3], [0.1, 0.063, 0.05])
    assert len(sc.emitters) == 1 and sc.emitters[0]["shape"] == len(sc.shapes) + 6 and sc.analytic[6]["emitter"] == 0
    glass = sc.bsdfs[sc.analytic[5]["bsdf"]]
    assert glass["type"] == S.BSDF_DIELECTRIC and glass["eta"][0] == pytest.approx(1.5046 / 1.000277, rel=1e-6)
    red = sc.bsdfs[sc.analytic[3]["bsdf"]]["reflectance"]
    assert red == pytest.approx([X.srgb_to_linear(0xa8 / 255), X.srgb_to_linear(0x20 / 255), X.srgb_to_linear(0x18 / 255)], rel=1e-6)
    # camera: lookat(origin, target, up) puts the origin in the last column and looks down +z
    np.testing.assert_allclose(sc.cam_to_world[:3, 3], [0, 0.11, 0.42], atol=1e-7)
    d = np.array([-0.015, 0.1, 0]) - np.array([0, 0.11, 0.42]); np.testing.assert_allclose(sc.cam_to_world[:3, 2], d / np.linalg.norm(d), atol=1e-6)
    assert X.load_scene(os.path.join(MESHES, "bunny_box.xml")).spp == 16          # the <default>


MINIMAL = """<scene version="0.5.0"><integrator type="path"/>
<sensor type="perspective">{sensor}<film type="hdrfilm"><integer name="width" value="64"/><integer name="height" value="32"/>{film}</film></sensor>
{body}
<shape type="rectangle"><emitter type="area"><spectrum name="radiance" value="3"/></emitter></shape></scene>"""


def load_text(tmp_path, text, **kw):
    p = tmp_path / "s.xml"; p.write_text(text)
    return X.load_scene(str(p), **kw)


def test_defaults_and_conventions(tmp_path):
    sc = load_text(tmp_path, MINIMAL.format(sensor="", film="", body=""))
    # plugin defaults: integrator.cpp:193-219, sensor.cpp:157-159, sampler plugins (4 samples), film.cpp:89-92 (gaussian filter), shape.cpp (diffuse 0.5)
    assert (sc.max_depth, sc.rr_depth, sc.strict_normals, sc.hide_emitters) == (-1, 5, 0, 0)
    assert (sc.near, sc.far, sc.spp, sc.sampler, sc.filter) == (pytest.approx(1e-2), pytest.approx(1e4), 4, S.SAMPLER_INDEPENDENT, S.FILTER_GAUSSIAN)
    assert sc.bsdfs[0]["type"] == S.BSDF_DIFFUSE and sc.bsdfs[0]["reflectance"] == (0.5, 0.5, 0.5)
    assert sc.emitters[0]["radiance"] == (3.0, 3.0, 3.0)
    # default focal length 50mm on a 36x24 sensor diagonal (sensor.cpp:264-277), aspect 2
    diag = 2 * math.atan(math.sqrt(36 ** 2 + 24 ** 2) / 100.0); w = 2 * math.tan(diag / 2) / math.sqrt(1 + 1 / 4.0)
    assert sc.xfov == pytest.approx(math.degrees(2 * math.atan(w / 2)), rel=1e-6)
    sc = load_text(tmp_path, MINIMAL.format(sensor='<float name="fov" value="40"/><string name="fovAxis" value="diagonal"/>', film='<rfilter type="mitchell"><float name="B" value="0.2"/></rfilter>', body=""))
    w = 2 * math.tan(math.radians(20)) / math.sqrt(1 + 1 / 4.0)
    assert sc.xfov == pytest.approx(math.degrees(2 * math.atan(w / 2)), rel=1e-6)
    assert sc.filter == S.FILTER_MITCHELL and sc.filter_radius == pytest.approx(0.2) and sc.filter_stddev == pytest.approx(1 / 3)
    # transforms compose left to right in document order, each multiplying from the left (scenehandler.cpp:432-447)
    body = '<shape type="sphere"><transform name="toWorld"><scale value="2"/><translate x="1"/><rotate z="1" angle="90"/></transform><float name="radius" value="0.5"/></shape>'
    sc = load_text(tmp_path, MINIMAL.format(sensor="", film="", body=body))
    sph = sc.analytic[0]
    assert sph["radius"] == pytest.approx(1.0) and np.allclose(sph["to_world"][:3, 3], [0, 1, 0], atol=1e-6)      # scale moved into the radius (sphere.cpp:113-122)
    # emitter order: scene-level emitters first in document order, then area lights in shape order (scene.cpp:527-547, :589-623)
    body = ('<shape type="disk"><emitter type="area"><rgb name="radiance" value="1,2,3"/></emitter></shape><emitter type="point"><point name="position" x="1" y="2" z="3"/></emitter>'
            '<emitter type="constant"><spectrum name="radiance" value="0.5"/></emitter>')
    sc = load_text(tmp_path, MINIMAL.format(sensor="", film="", body=body))
    assert [e["type"] for e in sc.emitters] == [S.EMITTER_POINT, S.EMITTER_CONSTANT, S.EMITTER_AREA, S.EMITTER_AREA]
    assert sc.emitters[2]["radiance"] == (1.0, 2.0, 3.0) and sc.analytic[0]["emitter"] == 2 and sc.analytic[1]["emitter"] == 3
    np.testing.assert_array_equal(sc.emitters[0]["to_world"][:3, 3], [1, 2, 3])


def test_obj_materials_groups_and_instances(tmp_path):
    (tmp_path / "two.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nusemtl a\nf 1 2 3\nusemtl b\nf 1 2 4\n")
    body = ('<bsdf type="diffuse" id="A"><rgb name="reflectance" value="0.1"/></bsdf>'
            '<shape type="obj"><string name="filename" value="two.obj"/><ref name="a" id="A"/><bsdf name="b" type="conductor"><string name="material" value="Au"/></bsdf></shape>'
            '<shape type="shapegroup" id="grp"><shape type="cube"><ref id="A"/></shape></shape>'
            '<shape type="instance"><ref id="grp"/><transform name="toWorld"><translate x="5"/></transform></shape>'
            '<shape type="instance"><ref id="grp"/></shape>')
    sc = load_text(tmp_path, MINIMAL.format(sensor="", film="", body=body))
    assert [sh["group"] for sh in sc.shapes] == [0, 0, 1] and [sh["tri_count"] for sh in sc.shapes] == [1, 1, 12]
    assert sc.bsdfs[sc.shapes[0]["bsdf"]]["reflectance"] == pytest.approx((0.1, 0.1, 0.1)) and sc.bsdfs[sc.shapes[1]["bsdf"]]["type"] == S.BSDF_CONDUCTOR
    eta, k = S.CONDUCTOR_IOR["Au"]
    np.testing.assert_allclose(sc.bsdfs[sc.shapes[1]["bsdf"]]["eta"], np.float32(eta) / np.float32(1.000277), rtol=1e-6)       # conductor.cpp:176: divided by extEta (air)
    assert sc.shapes[2]["bsdf"] == sc.shapes[0]["bsdf"]                        # one <ref>erenced BSDF = one material
    assert len(sc.instances) == 2 and sc.instances[0]["to_world"][0, 3] == 5 and sc.instances[0]["group"] == 0


@pytest.mark.parametrize("text,msg", [
    (MINIMAL.format(sensor="", film="", body='<shape type="hair"><string name="filename" value="x"/></shape>'), 'shape plugin "hair" is not supported'),
    (MINIMAL.format(sensor="", film="", body='<shape type="sphere"><bsdf type="irawan"/></shape>'), 'BSDF plugin "irawan" is not supported'),
    (MINIMAL.format(sensor="", film="", body='<shape type="sphere"><float name="radus" value="1"/></shape>'), "unused or unsupported property radus"),
    (MINIMAL.format(sensor="", film="", body='<shape type="sphere"><float name="radius" value="$r"/></shape>'), "undefined parameter"),
    (MINIMAL.format(sensor="", film="", body='<shape type="sphere"><ref id="nope"/></shape>'), "Referenced object 'nope' not found"),
    (MINIMAL.format(sensor="", film="", body='<shape type="obj"><string name="filename" value="missing.obj"/></shape>'), "could not be found"),
    (MINIMAL.format(sensor="", film="", body='<foo/>'), 'Unhandled tag "foo"'),
    (MINIMAL.format(sensor='<float name="fov" value="30"/><string name="focalLength" value="35mm"/>', film="", body=""), "either a focal length"),
    (MINIMAL.format(sensor='<transform name="toWorld"><scale value="2"/></transform>', film="", body=""), "Scale factors in the camera-to-world"),
    (MINIMAL.format(sensor="", film="", body='<emitter type="sunsky"/>'), 'emitter plugin "sunsky" is not supported'),
    (MINIMAL.format(sensor="", film="", body='<subsurface type="dipole" id="m"/>'), "subsurface"),
    (MINIMAL.replace('type="path"', 'type="bdpt"').format(sensor="", film="", body=""), 'integrator "bdpt" is not supported'),
    (MINIMAL.format(sensor="", film="", body='<shape type="sphere"><transform name="toWorld"><scale x="1" value="2"/></transform></shape>'), "both xyz and value"),
    ("<scene version='0.5.0'><integrator type='path'/><sensor type='perspective'/><shape type='sphere'/></scene>", "no emitters"),
    ("<scene", "XML parse error"),
])
def test_errors_name_the_problem(tmp_path, text, msg):
    with pytest.raises(X.SceneError, match=msg):
        load_text(tmp_path, text)


def test_images_for_environment_maps(tmp_path):
    img = (np.arange(4 * 8 * 3, dtype=np.float32).reshape(4, 8, 3) / 7.0)
    np.save(tmp_path / "e.npy", img)
    with open(tmp_path / "e.pfm", "wb") as f:
        f.write(b"PF\n8 4\n-1.0\n"); f.write(img[::-1].astype("<f4").tobytes())
    np.testing.assert_array_equal(X.load_image(str(tmp_path / "e.npy")), img)
    np.testing.assert_array_equal(X.load_image(str(tmp_path / "e.pfm")), img)
    # Radiance RGBE, flat scanlines: (128, 64, 32, 129) = (1.0, 0.5, 0.25)
    with open(tmp_path / "e.hdr", "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 2\n"); f.write(bytes([128, 64, 32, 129] * 4))
    np.testing.assert_allclose(X.load_image(str(tmp_path / "e.hdr")), np.tile([1.0, 0.5, 0.25], (2, 2, 1)), rtol=1e-6)
    with pytest.raises(X.SceneError, match="not readable"):
        X.load_image(str(tmp_path / "e.tiff"))
    body = '<emitter type="envmap"><string name="filename" value="e.pfm"/><float name="scale" value="2"/></emitter>'
    sc = load_text(tmp_path, MINIMAL.format(sensor="", film="", body=body))
    assert sc.envmap["scale"] == 2.0 and sc.envmap["rgb"].shape == (4, 8, 3) and sc.emitters[0]["type"] == S.EMITTER_ENVMAP
    np.testing.assert_array_equal(sc.envmap["rgb"], img.astype(np.float16).astype(np.float32))


def test_bitmap_texture_from_an_image_file(tmp_path):
    """<texture type="bitmap"> from an image: the MIP pyramid is built at load time like BitmapTexture does (scenes.build_mip_pyramid, pinned against the
    reference's pyramids in tests/test_oracle_golden.py) -- here from the base image of the reference-built 48x40 pyramid, so the result must equal it."""
    pyr = S.load_texture_pyramid()
    np.save(tmp_path / "tex.npy", pyr["base"])
    body = ('<shape type="cube"><bsdf type="diffuse"><texture type="bitmap" name="reflectance"><string name="filename" value="tex.npy"/>'
            '<string name="wrapModeV" value="repeat"/><float name="uvscale" value="2"/><string name="filterType" value="trilinear"/></texture></bsdf></shape>')
    sc = load_text(tmp_path, MINIMAL.format(sensor="", film="", body=body))
    t = sc.textures[0]
    assert t["type"] == S.TEXTURE_BITMAP and t["filter"] == S.MIP_TRILINEAR and (t["uscale"], t["vscale"]) == (2.0, 2.0) and t["n_levels"] == len(pyr["levels"])
    off = 0
    for (w, h, ref), lv in zip(pyr["levels"], sc.texture_levels):
        assert (int(lv[0]), int(lv[1]), int(lv[2])) == (w, h, off)
        np.testing.assert_array_equal(sc.texture_texels[off:off + w * h * 3], ref); off += w * h * 3
    assert sc.bsdfs[0]["texture"] == 0 and sc.uv is not None


def test_sampler_override(tmp_path):
    text = MINIMAL.format(sensor='<sampler type="ldsampler"><integer name="sampleCount" value="64"/><integer name="dimension" value="8"/></sampler>', film="", body="")
    with pytest.raises(X.SceneError, match='sampler "ldsampler" is not supported'):
        load_text(tmp_path, text)
    sc = load_text(tmp_path, text, sampler="sobol")
    assert sc.sampler == S.SAMPLER_SOBOL and sc.spp == 64
    with pytest.raises(X.SceneError, match="override"):
        load_text(tmp_path, text, sampler="halton")


def test_mask_with_alpha_channel_texture(tmp_path):
    """The classic leaf card: <bsdf type="mask"> whose opacity is the alpha channel of an image (BitmapTexture `channel`, bitmap.cpp:261-266) over a twosided diffuse."""
    rgba = np.zeros((8, 8, 4), np.float32); rgba[..., :3] = (0.2, 0.6, 0.1); rgba[2:6, 2:6, 3] = 1.0
    np.save(tmp_path / "leaf.npy", rgba)
    body = ('<shape type="rectangle"><bsdf type="mask"><texture type="bitmap" name="opacity"><string name="filename" value="leaf.npy"/><string name="channel" value="a"/>'
            '<string name="filterType" value="nearest"/><string name="wrapMode" value="clamp"/></texture>'
            '<bsdf type="twosided"><bsdf type="diffuse"><rgb name="reflectance" value="0.2, 0.6, 0.1"/></bsdf></bsdf></bsdf></shape>')
    body = body.replace('<shape type="rectangle">', '<shape type="cube">')       # (textured materials go on meshes: analytic shapes take constants)
    sc = load_text(tmp_path, MINIMAL.format(sensor="", film="", body=body))
    mask = sc.bsdfs[sc.shapes[0]["bsdf"]]
    assert mask["type"] == S.BSDF_MASK and mask["texture"] == 0 and sc.bsdfs[mask["distr"]]["type"] == S.BSDF_DIFFUSE and sc.bsdfs[mask["distr"]]["twosided"] == 1
    t = sc.textures[0]
    assert t["type"] == S.TEXTURE_BITMAP and t["filter"] == S.MIP_NEAREST and t["wrap_u"] == S.WRAP_CLAMP and t["n_levels"] == 1
    lvl0 = sc.texture_texels[:8 * 8 * 3].reshape(8, 8, 3)
    np.testing.assert_array_equal(lvl0[..., 0], rgba[..., 3]); np.testing.assert_array_equal(lvl0[..., 1], rgba[..., 3])
    assert t["color0"] == pytest.approx((0.25, 0.25, 0.25))                  # the average opacity (16 of 64 texels)
    with pytest.raises(X.SceneError, match='Channel "q" not found'):
        load_text(tmp_path, MINIMAL.format(sensor="", film="", body=body.replace('value="a"', 'value="q"')))


def test_blackbody_spectra_match_reference(tmp_path):
    """<blackbody temperature="..K" scale=".."/> (scenehandler.cpp:618-631) against the reference's own conversion of BlackBodySpectrum
    (tests/golden/blackbody_rgb.npz, from oracle/_ref/harness `blackbody`): within 5e-5 of the largest channel."""
    g = np.load(os.path.join(HERE, "golden", "blackbody_rgb.npz"))
    for t, rgb in zip(g["temperature"], g["rgb"]):
        mine = np.array(X.blackbody_to_rgb(float(t)))
        assert np.abs(mine - rgb).max() / rgb.max() < 5e-5, (t, mine, rgb)
    p = tmp_path / "bb.xml"
    p.write_text("""<scene version="0.5.0"><integrator type="path"/><sensor type="perspective"><sampler type="independent"/><film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/></film></sensor>
<shape type="rectangle"><emitter type="area"><blackbody name="radiance" temperature="$temp" scale="1e-4"/></emitter></shape></scene>""")
    sc = X.load_scene(str(p), params={"temp": "5000K"})
    want = np.array(X.blackbody_to_rgb(5000.0, 1e-4))
    assert np.allclose(sc.emitters[0]["radiance"], want, rtol=1e-6) and abs(want[0] - 1.52633) < 1e-3
    with pytest.raises(X.SceneError):
        X.load_scene(str(p), params={"temp": "warm"})


@pytest.mark.parametrize("kw", [dict(), dict(global_fog=True, integrator=S.INTEGRATOR_VOLPATH)])
def test_media_round_trip(kw, tmp_path):
    """Participating media in scene files: <medium type="homogeneous"> with its <phase>, `interior` / `exterior` references on meshes and analytic shapes, the
    sensor's medium, `null` BSDFs, the volumetric integrators -- written by export_scene, read back, and traced to the same radiance by the oracle."""
    import oracle
    oracle.build()
    sc = S.fog_box(48, 48, 4, **kw)
    sc.xfov = float(np.float32(sc.xfov)); sc.sample_to_camera = S.sample_to_camera(sc.xfov, sc.near, sc.far, sc.width / sc.height)
    sc2 = X.load_scene(X.export_scene(sc, str(tmp_path), name="fog"))
    assert sc2.integrator == sc.integrator and len(sc2.media) == len(sc.media) and (sc2.sensor_medium >= 0) == (sc.sensor_medium >= 0)
    assert sorted(map(repr, sc2.media)) == sorted(map(repr, sc.media))                  # (the file's order of first use may differ from the generator's)
    pairs = np.stack([np.arange(400) % 48, (np.arange(400) * 7) % 48, np.arange(400) % 4], 1).astype(np.uint32)
    a = oracle.Oracle(sc).render_samples(pairs)["li"]; b = oracle.Oracle(sc2).render_samples(pairs)["li"]
    assert (a.view(np.uint32) == b.view(np.uint32)).all()


def test_media_conventions(tmp_path):
    """What the reference's constructors derive (src/medium/homogeneous.cpp:156-226, src/medium/materials.h:88-192, src/librender/shape.cpp:47-75): sigmaT / albedo,
    `scale`, the sampling weight from the largest albedo (at least 1/2), `single` picking the smallest sigma_t, a `null` BSDF for a medium transition without one."""
    head = ('<scene version="0.5.0"><integrator type="volpath"><integer name="maxDepth" value="4"/></integrator>'
            '<sensor type="perspective"><transform name="toWorld"><lookat origin="0,0,-5" target="0,0,0" up="0,1,0"/></transform><sampler type="independent"><integer name="sampleCount" value="2"/></sampler>'
            '<film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/></film></sensor>'
            '<emitter type="point"><rgb name="intensity" value="1,1,1"/></emitter>')
    def load(body):
        p = tmp_path / "m.xml"; p.write_text(head + body + "</scene>"); return X.load_scene(str(p))
    sc = load('<medium type="homogeneous" id="fog"><rgb name="sigmaT" value="2, 4, 8"/><rgb name="albedo" value="0.5, 0.25, 0.75"/><float name="scale" value="0.5"/><string name="strategy" value="single"/>'
              '<phase type="hg"><float name="g" value="0.3"/></phase></medium><shape type="sphere"><ref name="interior" id="fog"/></shape>')
    m = sc.media[0]
    np.testing.assert_allclose(m["sigma_s"], [0.5, 0.5, 3.0], rtol=1e-6); np.testing.assert_allclose(m["sigma_a"], [0.5, 1.5, 1.0], rtol=1e-6)
    assert m["strategy"] == S.MEDIUM_SINGLE and m["sampling_density"] == pytest.approx(1.0) and m["medium_sampling_weight"] == pytest.approx(0.75) and m["phase"] == S.PHASE_HG and m["g"] == pytest.approx(0.3)
    assert sc.integrator == S.INTEGRATOR_VOLPATH and sc.bsdfs[sc.analytic[0]["bsdf"]]["type"] == S.BSDF_NULL and list(sc.shape_media[0]) == [0, -1]
    with pytest.raises(X.SceneError, match="maximum"):
        load('<shape type="sphere"><medium name="interior" type="homogeneous"><rgb name="sigmaS" value="1,1,1"/><rgb name="sigmaA" value="1,1,1"/><string name="strategy" value="maximum"/></medium></shape>')
    with pytest.raises(X.SceneError, match="interior"):
        load('<shape type="sphere"><medium type="homogeneous"><rgb name="sigmaS" value="1,1,1"/><rgb name="sigmaA" value="1,1,1"/></medium></shape>')
    with pytest.raises(X.SceneError, match="heterogeneous"):
        load('<shape type="sphere"><medium name="interior" type="heterogeneous"/></shape>')
