#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by running the REFERENCE itself (oracle/_ref/harness, built by
oracle/ref_build/Makefile from /root/reference).  Runs only in the build container; the fixtures it writes are data
(inputs + the reference's outputs) and are committed.  Usage: python tests/golden/make_golden.py"""
import importlib
import os
import subprocess
import sys
import tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
scenes = importlib.import_module("mitsuba-im_amd.scenes")
OUT = os.path.join(ROOT, "tests", "golden")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "harness")


def run(*args):
    subprocess.check_call([HARNESS] + [str(a) for a in args], cwd=os.path.join(ROOT, "oracle", "_ref"), timeout=600)


def golden_scenes():
    """name -> scene; small film sizes keep fixtures small, 1080p variants pin the Sobol m=11 index math."""
    return {
        "cornell_sobol": scenes.cornell_box(width=1920, height=1080, spp=8, sampler=scenes.SAMPLER_SOBOL),
        "cornell_indep": scenes.cornell_box(width=1920, height=1080, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=0),
        "cornell_small": scenes.cornell_box(width=96, height=54, spp=16, sampler=scenes.SAMPLER_SOBOL),
        "cornell_small_gauss": scenes.cornell_box(width=96, height=54, spp=4, sampler=scenes.SAMPLER_SOBOL, filter_kind=scenes.FILTER_GAUSSIAN),
        "closed_box": scenes.closed_box(width=64, height=64, spp=16),
        "veach_small": scenes.veach_mis(width=96, height=54, spp=16),
        "atrium_small": scenes.atrium(width=96, height=54, spp=16, detail=0.08, env_size=(64, 32)),
        # integrator switches: strictNormals (smooth-shaded columns make it bite), hideEmitters, early Russian roulette, unbounded depth
        "atrium_strict": scenes.atrium(width=96, height=54, spp=8, detail=0.08, env_size=(64, 32), strict_normals=True, rr_depth=2),
        "atrium_hide_indep": scenes.atrium(width=96, height=54, spp=8, detail=0.08, env_size=(64, 32), hide_emitters=True, sampler=scenes.SAMPLER_INDEPENDENT,
                                           max_depth=-1, rr_depth=3, seed=7),
        "cornell_hide": scenes.cornell_box(width=96, height=54, spp=8, hide_emitters=True, max_depth=3, rr_depth=1),
        # analytic shapes behind rayIntersect (rectangle / disk / sphere / cylinder) as geometry and as area lights
        "cbox_shapes": scenes.cbox_shapes(width=96, height=96, spp=16),
        "shape_lights": scenes.shape_lights(width=96, height=64, spp=16),
        "cbox_shapes_strict_indep": scenes.cbox_shapes(width=96, height=96, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=3, strict_normals=True, hide_emitters=True, rr_depth=2, disk_cap=False),
        # scene-level emitters: point + spot next to the area light; constant environment + directional light
        "cbox_lights": scenes.cbox_lights(width=96, height=96, spp=16),
        "open_constant": scenes.open_constant(width=96, height=64, spp=16),
        "open_constant_hide_indep": scenes.open_constant(width=96, height=64, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=11, hide_emitters=True, rr_depth=2),
        # smooth BSDFs: dielectric, conductor, plastic (linear / nonlinear), twosided(conductor)
        "cbox_materials": scenes.cbox_materials(width=96, height=96, spp=16),
        "cbox_materials_strict_indep": scenes.cbox_materials(width=96, height=96, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=4, strict_normals=True, hide_emitters=True, rr_depth=3, max_depth=-1),
        # shapegroup + instance (two groups, 16 placements with rotation / non-uniform scale), smooth-shaded and rough-conductor members
        "instanced_garden": scenes.instanced_garden(width=96, height=64, spp=16),
        # roughdielectric (extra sampler dimension per bounce) + difftrans
        "cbox_translucent": scenes.cbox_translucent(width=96, height=96, spp=16),
        "cbox_translucent_indep": scenes.cbox_translucent(width=96, height=96, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=9, rr_depth=2, strict_normals=True),
        # roughplastic (rough-transmittance slices of the reference's data/microfacet tables as material input)
        "cbox_roughplastic": scenes.cbox_roughplastic(width=96, height=96, spp=16),
        # the other reconstruction filters (negative lobes: mitchell, catmullrom, lanczos)
        "cornell_small_tent": scenes.cornell_box(width=96, height=54, spp=4, filter_kind=scenes.FILTER_TENT),
        "cornell_small_mitchell": scenes.cornell_box(width=96, height=54, spp=4, filter_kind=scenes.FILTER_MITCHELL),
        "cornell_small_catmullrom": scenes.cornell_box(width=96, height=54, spp=4, filter_kind=scenes.FILTER_CATMULLROM),
        "cornell_small_lanczos": scenes.cornell_box(width=96, height=54, spp=4, filter_kind=scenes.FILTER_LANCZOS),
        # meshes with texture coordinates (UV tangents) + procedural textures
        "textured_room": scenes.textured_room(width=96, height=64, spp=16),
        # bitmap textures: MIP pyramid (input data = the reference's own), EWA / trilinear / bilinear / nearest, wrap modes, ray differentials
        "bitmap_room": scenes.bitmap_room(width=96, height=64, spp=16),
        # camera rays straight into a detailed environment map: EWA-filtered lookups with the sensor ray's differentials
        # (footprints shrink with 1 / sqrt(spp), integrator.cpp:145-146: 1-2 spp on a small film under a 1024 x 512 map gives minification)
        "sky_view": scenes.sky_view(width=60, height=44, spp=1, env_size=(1024, 512)),      # not a power of two: the reference's responsive driver walks (W+2b) x (H+2b) pixels (integrator.cpp:338-339) and Sobol pixel 64 would alias pixel 0
        "sky_view_indep": scenes.sky_view(width=60, height=44, spp=2, env_size=(1024, 512), sampler=scenes.SAMPLER_INDEPENDENT, seed=2),
        # Sobol sampler with a scramble value (sobol.cpp:92-102: frame number -> sampleTEA; XORed into every sample, flips the pixel bits of look_up)
        "cornell_scramble": scenes.cornell_box(width=96, height=54, spp=8, sampler=scenes.SAMPLER_SOBOL, seed=7),
        # the rest of MicrofacetDistribution on the Veach plates: anisotropic Beckmann / GGX, sampleVisible = false, Phong and Ashikhmin-Shirley
        "veach_microfacets": scenes.veach_mis(width=96, height=54, spp=16, microfacets=scenes.VEACH_MICROFACETS),
        "veach_microfacets_2": scenes.veach_mis(width=96, height=54, spp=8, microfacets=scenes.VEACH_MICROFACETS_2, sampler=scenes.SAMPLER_INDEPENDENT, seed=3),
        # roughdielectric over the rest of MicrofacetDistribution: anisotropic sphere (tangent from its own parameterisation), all-normal sampling with
        # Walter's widened sampling distribution (roughdielectric.cpp:409-414), Phong
        "cbox_translucent_mf": scenes.cbox_translucent(width=96, height=96, spp=16, frost_kw=dict(alpha=0.08, alpha_v=0.3, distr=scenes.DISTR_GGX, sample_visible=False),
                                                      slab_kw=dict(alpha=0.2, distr=scenes.DISTR_BECKMANN, sample_visible=False)),
        "cbox_translucent_mf2": scenes.cbox_translucent(width=96, height=96, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=5, frost_kw=dict(alpha=0.25, alpha_v=0.1, distr=scenes.DISTR_BECKMANN, sample_visible=True),
                                                       slab_kw=dict(alpha=0.3, distr=scenes.DISTR_PHONG)),
        # textures on plastic.diffuseReflectance / roughplastic.diffuseReflectance / difftrans.transmittance (lobe weights from the texture's average)
        "textured_plastics": scenes.textured_plastics(width=96, height=64, spp=16),
        "textured_plastics_smooth": scenes.textured_plastics(width=96, height=64, spp=8, rough=False, sampler=scenes.SAMPLER_INDEPENDENT, seed=6),
        # thindielectric panes: ENull transmission keeps the path "unscattered" (hideEmitters hides the sky through the glass, not the area light)
        "glass_pane": scenes.glass_pane(width=96, height=64, spp=16),
        "glass_pane_hide_indep": scenes.glass_pane(width=96, height=64, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=8, hide_emitters=True, rr_depth=2),
        # mask: cut-out screen (checkerboard opacity), tinted half-transparent sheet over plastic, grid-masked rough conductor; ENull pass-through
        "masked_room": scenes.masked_room(width=96, height=64, spp=16),
        "masked_room_hide_indep": scenes.masked_room(width=96, height=64, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=12, hide_emitters=True, rr_depth=2),
        # textures on analytic shapes (their own uv parameterisations): checkerboard rectangle, grid plastic sphere, bitmap cylinder, checkerboard disk, masked rectangle
        "textured_shapes": scenes.textured_shapes(width=96, height=96, spp=16),
        # a crop window of a larger frame (Film cropOffsetX/Y, cropWidth/Height): the camera maps the rendered film onto its part of the full frame
        "cornell_crop": scenes.set_crop_window(scenes.cornell_box(width=60, height=36, spp=8), 192, 108, 70, 40),      # (60 wide, not 64: see sky_view)
        # roughplastic sampling all normals instead of the visible ones (sampleVisible = false)
        "cbox_roughplastic_phong": scenes.cbox_roughplastic(width=96, height=96, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=22, phong=True),
        "cbox_roughplastic_allnormals": scenes.cbox_roughplastic(width=96, height=96, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=21, sample_visible=False),
        # the BSDF adapters: bumpmap (bitmap displacement under a scale texture; grid displacement by finite differences), normalmap, mixturebsdf (2 and 3 children,
        # weights rescaled, twosided), bumpmap(mixture), mask(bumpmap)
        "layered_room": scenes.layered_room(width=96, height=64, spp=16),
        "layered_room_procedural": scenes.layered_room(width=96, height=64, spp=16, procedural_maps=True),
        "layered_room_strict_indep": scenes.layered_room(width=96, height=64, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=13, strict_normals=True, rr_depth=2),
        # a scene FILE: hand-written XML around the reference's own test asset (data/tests/bunny.ply, 69451 triangles, generated vertex normals), read by
        # mitsuba-im_amd/xml_scene.py + meshio.py and handed to the reference flattened
        "bunny_box": importlib.import_module("mitsuba-im_amd.xml_scene").load_scene(os.path.join(OUT, "meshes", "bunny_box.xml")),
        # volpath_simple over homogeneous media: smoke cube behind a `null` boundary (isotropic, balance), glass block with a forward-scattering
        # interior (hg, single), `null` sphere of haze (hg, manual); _global: the sensor sits in a thin fog that fills the room
        "fog_box": scenes.fog_box(width=96, height=96, spp=16),
        # the reference's `sunsky` emitter (compound: rasterised sky + sun -> envmap; sunRadiusScale = 0 -> envmap + directional): drop-in fixtures only, the
        # integrator sees the expanded elements (scene.cpp:530-539)
        "sunsky_terrace": scenes.sunsky_terrace(width=96, height=64, spp=16),
        "sunsky_terrace_dirsun": scenes.sunsky_terrace(width=96, height=64, spp=16, sun_radius_scale=0.0),
        "fog_box_global": scenes.fog_box(width=96, height=96, spp=8, global_fog=True, sampler=scenes.SAMPLER_INDEPENDENT, seed=14, rr_depth=2),
        "fog_box_global_hide": scenes.fog_box(width=96, height=96, spp=8, global_fog=True, hide_emitters=True, strict_normals=True, max_depth=5),
        # the same rooms through `volpath` (multiple importance sampling; emitters found through index-matched boundaries)
        "fog_mis": scenes.fog_box(width=96, height=96, spp=16, integrator=scenes.INTEGRATOR_VOLPATH),
        "fog_mis_global": scenes.fog_box(width=96, height=96, spp=8, global_fog=True, sampler=scenes.SAMPLER_INDEPENDENT, seed=15, rr_depth=2, integrator=scenes.INTEGRATOR_VOLPATH),
        # the volumetric loops under an environment map (sky seen through media, emitter sampling of the map attenuated by the media)
        "fog_sky": scenes.fog_sky(width=96, height=64, spp=16),
        "fog_sky_simple": scenes.fog_sky(width=96, height=64, spp=16, integrator=scenes.INTEGRATOR_VOLPATH_SIMPLE),
        "fog_sky_global_hide": scenes.fog_sky(width=96, height=64, spp=8, global_fog=True, hide_emitters=True, sampler=scenes.SAMPLER_INDEPENDENT, seed=16, rr_depth=2),
        "fog_constant": scenes.fog_sky(width=96, height=64, spp=16, constant_env=True),
        "fog_constant_simple_indep": scenes.fog_sky(width=96, height=64, spp=8, constant_env=True, integrator=scenes.INTEGRATOR_VOLPATH_SIMPLE, sampler=scenes.SAMPLER_INDEPENDENT, seed=17, rr_depth=2),
        # a thin glass pane (thindielectric: ENull transmission) in the room: transmittance walks and the emitter search pass through it, attenuated
        "fog_pane": scenes.fog_box(width=96, height=96, spp=16, pane=True),
        "fog_pane_mis": scenes.fog_box(width=96, height=96, spp=8, pane=True, global_fog=True, integrator=scenes.INTEGRATOR_VOLPATH, sampler=scenes.SAMPLER_INDEPENDENT, seed=18),
        # the BSDF adapters (mixturebsdf / bumpmap / normalmap) inside volumetric renders: the layered room filled with fog, a `null` sphere of haze over the mound
        "fog_layered": scenes.layered_room(width=96, height=64, spp=16, fog=scenes.INTEGRATOR_VOLPATH_SIMPLE),
        # `mask` inside volumetric renders: its ENull lobe (1 - opacity, textured) in the transmittance walks and the emitter search; volpath_simple goes through the pdf-less sample overload
        "fog_masked": scenes.masked_room(width=96, height=64, spp=16, fog=scenes.INTEGRATOR_VOLPATH_SIMPLE),
        "fog_masked_mis": scenes.masked_room(width=96, height=64, spp=8, fog=scenes.INTEGRATOR_VOLPATH, sampler=scenes.SAMPLER_INDEPENDENT, seed=31, hide_emitters=True),
        "fog_dusty": scenes.fog_box(width=96, height=96, spp=16, pane=True, dusty=True),
        "fog_dusty_mis": scenes.fog_box(width=96, height=96, spp=8, pane=True, dusty=True, global_fog=True, integrator=scenes.INTEGRATOR_VOLPATH, sampler=scenes.SAMPLER_INDEPENDENT, seed=41),
        "fog_layered_procedural": scenes.layered_room(width=96, height=64, spp=8, fog=scenes.INTEGRATOR_VOLPATH, procedural_maps=True),      # (for the drop-in test: no bitmap textures)
        "fog_layered_mis": scenes.layered_room(width=96, height=64, spp=8, fog=scenes.INTEGRATOR_VOLPATH, sampler=scenes.SAMPLER_INDEPENDENT, seed=23, strict_normals=True),
        "fog_mis_global_hide": scenes.fog_box(width=96, height=96, spp=8, global_fog=True, hide_emitters=True, strict_normals=True, max_depth=5, integrator=scenes.INTEGRATOR_VOLPATH),
        # round 3 (appended, so the pair streams of the scenes above stay what they were): a `collimated` beam among the emitters (a 0-D emitter: never sampled successfully)
        "cbox_collimated": scenes.cbox_collimated(width=96, height=96, spp=16),
        # `roughdiffuse` (Oren-Nayar): the full model and the qualitative approximation
        "cbox_roughdiffuse": scenes.cbox_roughdiffuse(width=96, height=96, spp=16),
        "cbox_roughdiffuse_strict_indep": scenes.cbox_roughdiffuse(width=96, height=96, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=17, strict_normals=True),
        # modified `phong`
        "cbox_phong": scenes.cbox_phong(width=96, height=96, spp=16),
        "cbox_phong_strict_indep": scenes.cbox_phong(width=96, height=96, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=19, strict_normals=True),
        # `ward` in its three variants
        "cbox_ward": scenes.cbox_ward(width=96, height=96, spp=16),
        "cbox_ward_strict_indep": scenes.cbox_ward(width=96, height=96, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=21, strict_normals=True),
        "ward_room": scenes.ward_room(width=96, height=64, spp=16),
        # `coating` over diffuse / rough conductor / smooth conductor, with and without absorption
        "cbox_coating": scenes.cbox_coating(width=96, height=96, spp=16),
        "cbox_roughcoating": scenes.cbox_roughcoating(width=96, height=96, spp=16),      # `roughcoating`: Beckmann / GGX / Phong interfaces
        "blend_room": scenes.blend_room(width=96, height=64, spp=16),      # `blendbsdf` with textured and constant weights
        "cbox_coating_strict_indep": scenes.cbox_coating(width=96, height=96, spp=8, sampler=scenes.SAMPLER_INDEPENDENT, seed=25, strict_normals=True),
    }


HARNESS_STRICT = os.path.join(ROOT, "oracle", "_ref_strict", "harness")


def _harness(exe, *args, timeout=3600):
    subprocess.check_call([exe] + [str(a) for a in args], cwd=os.path.dirname(exe), timeout=timeout, stdout=subprocess.DEVNULL)


def develop(film, sc):
    """raw 5-channel ImageBlock sums incl. border -> H x W x 3 image (sum / weight), float32"""
    b = (film.shape[0] - sc.height) // 2
    f = film[b:film.shape[0] - b, b:film.shape[1] - b]
    return (f[..., :3] / np.maximum(f[..., 4:5], 1e-20)).astype(np.float32)


def strict_fixtures(only=None):
    """The SAME reference sources compiled WITHOUT -ffast-math (`make -C oracle/ref_build FAST=0 OUT=oracle/_ref_strict`): per-sample Li, path depth and
    number of sampler values for the pairs of every small golden scene -> tests/golden/strict/<name>.npz.  Against this build the restatement takes the same
    path for every sample and most scenes agree bit for bit (tests/test_oracle_golden.py::test_li_samples_vs_strict_reference): what separates it from the
    shipped -ffast-math build is compiler re-association inside the reference, not the algorithm."""
    os.makedirs(os.path.join(OUT, "strict"), exist_ok=True)
    tmp = tempfile.mkdtemp()
    for name, sc in golden_scenes().items():
        if only and name not in only:
            continue
        pairs = np.load(os.path.join(OUT, name + "_samples.npz"))["pairs"]
        path = os.path.join(tmp, name + ".miscene"); scenes.save_scene(sc, path)
        ppath = os.path.join(tmp, name + "_pairs.bin"); pairs.tofile(ppath); base = os.path.join(tmp, name)
        _harness(HARNESS_STRICT, path, "samples", ppath, base)
        extra = {}
        if name in ("cornell_small", "veach_small", "atrium_small", "instanced_garden", "bunny_box", "closed_box"):
            _harness(HARNESS_STRICT, path, "image", 8, base); extra["image"] = develop(np.load(base + "_film.npy"), sc)
        np.savez_compressed(os.path.join(OUT, "strict", name + ".npz"), li=np.load(base + "_li.npy"), depth=np.load(base + "_depth.npy"), nvals=np.load(base + "_nsamples.npy"), **extra)
        print("strict", name, flush=True)


# converged low-resolution images of the three BASELINE scene classes: 240 x 135, Sobol, enough samples per pixel that the
# reference's OWN two builds (-ffast-math as shipped / strict IEEE) agree below the 1e-4 relative-L2 tolerance of the north star
CONVERGED = {"S1_cornell": ("cornell_box", dict(), 1024), "S2_veach": ("veach_mis", dict(), 32768), "S3_atrium": ("atrium", dict(detail=0.08, env_size=(64, 32)), 32768),
             "S4_fog": ("fog_box", dict(global_fog=True, integrator=scenes.INTEGRATOR_VOLPATH), 2048)}      # the volumetric loop (volpath) over the fog_box room


def converged_scene(key):
    gen, kw, spp = CONVERGED[key]
    return getattr(scenes, gen)(240, 135, spp, **kw)


def converged_fixtures(only=None):
    os.makedirs(os.path.join(OUT, "converged"), exist_ok=True)
    tmp = tempfile.mkdtemp()
    for key in CONVERGED:
        if only and key not in only:
            continue
        sc = converged_scene(key); path = os.path.join(tmp, key + ".miscene"); scenes.save_scene(sc, path); imgs = {}
        for tag, exe in (("fast", HARNESS), ("strict", HARNESS_STRICT)):
            base = os.path.join(tmp, key + "_" + tag); _harness(exe, path, "image", 8, base, timeout=7200)
            imgs[tag] = develop(np.load(base + "_film.npy"), sc)
        a, b = imgs["fast"].astype(np.float64), imgs["strict"].astype(np.float64)
        print("converged", key, sc.spp, "spp: reference fast-math vs strict build rel-L2 = %.3g" % np.sqrt(((a - b) ** 2).sum() / (a ** 2).sum()), flush=True)
        np.savez_compressed(os.path.join(OUT, "converged", key + ".npz"), fast=imgs["fast"], strict=imgs["strict"], spp=np.int64(sc.spp))


def main():
    if "--strict" in sys.argv:
        i = sys.argv.index("--strict"); return strict_fixtures(set(sys.argv[i + 1].split(",")) if len(sys.argv) > i + 1 else None)
    if "--converged" in sys.argv:
        i = sys.argv.index("--converged"); return converged_fixtures(set(sys.argv[i + 1].split(",")) if len(sys.argv) > i + 1 else None)
    only = None                                          # --only a,b: regenerate just these scenes (the pair streams of all scenes are still drawn, in order)
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    tmp = tempfile.mkdtemp()
    if only is None:
        run("tables", OUT)
    rng = np.random.default_rng(20251004)
    for name, sc in golden_scenes().items():
        path = os.path.join(tmp, name + ".miscene")
        scenes.save_scene(sc, path)
        n = 2048
        pairs = np.stack([rng.integers(0, sc.width, n), rng.integers(0, sc.height, n), rng.integers(0, sc.spp, n)], 1).astype(np.uint32)
        pairs[:8] = [[0, 0, 0], [sc.width - 1, sc.height - 1, sc.spp - 1], [sc.width // 2, sc.height // 2, 0], [1, 0, 1],
                     [0, 1, 2], [sc.width - 1, 0, 3], [0, sc.height - 1, 1], [sc.width // 3, sc.height // 3, sc.spp - 1]]
        if only is not None and name not in only:
            continue
        ppath = os.path.join(tmp, name + "_pairs.bin"); pairs.tofile(ppath)
        base = os.path.join(tmp, name)
        if name.startswith("sunsky"):                   # only the responsive target: neither the oracle nor the standalone front end builds this emitter
            run(path, "responsive", "path", -1, base + "_resp")
            np.savez_compressed(os.path.join(OUT, name + "_responsive.npz"), target=np.load(base + "_resp_target.npy"), meta=np.load(base + "_resp_meta.npy"))
            continue
        run(path, "samples", ppath, base)
        np.savez_compressed(os.path.join(OUT, name + "_samples.npz"), pairs=pairs,
                            li=np.load(base + "_li.npy"), pos=np.load(base + "_pos.npy"), ray=np.load(base + "_ray.npy"),
                            depth=np.load(base + "_depth.npy"), nvals=np.load(base + "_nsamples.npy"),
                            vals=np.load(base + "_svalues.npy")[:512])
        if name in ("cornell_sobol", "closed_box", "veach_small", "atrium_small", "cbox_shapes", "shape_lights", "cbox_lights", "cbox_collimated", "cbox_roughdiffuse", "cbox_phong", "cbox_ward", "ward_room", "cbox_coating", "blend_room", "cbox_roughcoating", "open_constant", "cbox_materials", "instanced_garden", "cbox_translucent", "cbox_roughplastic", "textured_room", "bitmap_room", "veach_microfacets", "veach_microfacets_2", "cbox_translucent_mf", "cbox_translucent_mf2", "glass_pane", "masked_room", "textured_shapes", "cornell_crop", "layered_room"):
            run(path, "hits", 97 if sc.width > 1000 else 5, base + "_hits.npy")
            run(path, "camera", base)
            run(path, "units", base)
            np.savez_compressed(os.path.join(OUT, name + "_units.npz"), hits=np.load(base + "_hits.npy"),
                                camrays=np.load(base + "_camrays.npy"), filter=np.load(base + "_filter.npy"),
                                warp=np.load(base + "_warp.npy"), triaccel=np.load(base + "_triaccel.npy"),
                                emitter=np.load(base + "_emitter.npy"), bsdf=np.load(base + "_bsdf.npy"))
        if name in ("fog_box", "fog_box_global", "fog_mis", "fog_mis_global", "fog_sky", "fog_masked", "fog_masked_mis", "fog_pane_mis", "fog_layered_procedural", "cornell_small", "atrium_small", "cbox_shapes", "cbox_lights", "cbox_collimated", "cbox_roughdiffuse", "cbox_phong", "cbox_ward", "ward_room", "cbox_coating", "blend_room", "cbox_roughcoating", "open_constant", "cbox_materials", "veach_small", "instanced_garden", "cbox_translucent", "textured_room", "sky_view", "veach_microfacets", "textured_plastics_smooth", "glass_pane", "masked_room", "cornell_crop", "layered_room", "layered_room_procedural"):
            # the reference's own `path` through the RESPONSIVE interface (ImageOrderIntegrator -> ClassicSamplingIntegrator), one thread:
            # the target the drop-in plugin must reproduce (tests/test_gpu_dropin.py)
            run(path, "responsive", {1: "volpath_simple", 2: "volpath"}.get(sc.get("integrator", 0), "path"), -1, base + "_resp")
            np.savez_compressed(os.path.join(OUT, name + "_responsive.npz"), target=np.load(base + "_resp_target.npy"), meta=np.load(base + "_resp_meta.npy"))
        if sc.width < 200:
            run(path, "image", 8, base)
            stats = open(base + "_stats.txt").read()
            np.savez_compressed(os.path.join(OUT, name + "_image.npz"), film=np.load(base + "_film.npy"), stats=np.array(stats))
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
