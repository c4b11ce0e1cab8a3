"""ctypes binding of libmi355pt.so (C-ABI: include/mi355pt.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback on the product path."""
import ctypes as C
import os
import struct
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355PT_LIB") or os.path.join(_HERE, "libmi355pt.so")   # MI355PT_LIB: A/B another build of the same library
SOBOL_PATH = os.path.join(_HERE, "data", "sobol_tables.bin")


class MiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mi355pt error {code}: {msg}")
        self.code = code


class MiShape(C.Structure):
    _fields_ = [("first_tri", C.c_uint32), ("tri_count", C.c_uint32), ("first_vert", C.c_uint32), ("vert_count", C.c_uint32),
                ("bsdf", C.c_int32), ("emitter", C.c_int32), ("flags", C.c_uint32), ("pad", C.c_uint32)]


class MiMaterial(C.Structure):
    _fields_ = [("type", C.c_uint32), ("flags", C.c_uint32), ("distr", C.c_uint32), ("alpha", C.c_float),
                ("reflectance", C.c_float * 3), ("eta", C.c_float * 3), ("k", C.c_float * 3), ("specular", C.c_float * 3)]


class MiEmitter(C.Structure):
    _fields_ = [("type", C.c_uint32), ("shape", C.c_int32), ("radiance", C.c_float * 3), ("weight", C.c_float), ("cutoff", C.c_float), ("beam", C.c_float),
                ("to_world", C.c_float * 16)]


class MiAnalytic(C.Structure):
    _fields_ = [("type", C.c_uint32), ("bsdf", C.c_int32), ("emitter", C.c_int32), ("flags", C.c_uint32),
                ("to_world", C.c_float * 16), ("to_object", C.c_float * 16), ("radius", C.c_float), ("length", C.c_float), ("pad", C.c_float * 2)]


class MiTexture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("color0", C.c_float * 3), ("color1", C.c_float * 3), ("line_width", C.c_float),
                ("uoffset", C.c_float), ("voffset", C.c_float), ("uscale", C.c_float), ("vscale", C.c_float),
                ("wrap_u", C.c_uint32), ("wrap_v", C.c_uint32), ("filter", C.c_uint32), ("max_anisotropy", C.c_float), ("first_level", C.c_uint32), ("n_levels", C.c_uint32)]


class MiInstance(C.Structure):
    _fields_ = [("group", C.c_uint32), ("pad", C.c_uint32 * 3), ("to_world", C.c_float * 16), ("to_object", C.c_float * 16)]


class MiRenderParams(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("rr_depth", C.c_int32), ("strict_normals", C.c_uint32), ("hide_emitters", C.c_uint32),
                ("sampler", C.c_uint32), ("spp", C.c_uint32), ("seed", C.c_uint64), ("device", C.c_uint32), ("planes_per_batch", C.c_uint32), ("opacity", C.c_uint32), ("integrator", C.c_uint32)]


class MiTile(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32)]


class MiStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("path_length_sum", C.c_uint64), ("samples", C.c_uint64),
                ("render_ms", C.c_double), ("extend_ms", C.c_double), ("shade_ms", C.c_double), ("shadow_ms", C.c_double), ("other_ms", C.c_double),
                ("extend_launches", C.c_uint64), ("extend_rays", C.c_uint64), ("extend_launches_all", C.c_uint64)]


EXPORTS = ["mi_last_error", "mi_set_sobol_tables", "mi_load_sobol_tables", "mi_scene_create", "mi_scene_destroy", "mi_scene_set_triangles",
           "mi_scene_set_analytic", "mi_scene_set_instances", "mi_scene_set_media", "mi_scene_set_materials", "mi_scene_set_material_tables", "mi_scene_set_textures", "mi_scene_set_texture_data", "mi_scene_set_emitters", "mi_scene_set_envmap", "mi_scene_set_envmap_filter", "mi_scene_set_camera", "mi_scene_set_film",
           "mi_scene_commit", "mi_scene_ray_intersect", "mi_scene_clone", "mi_render_merge_film", "mi_render_create", "mi_render_destroy", "mi_render_run", "mi_render_run_rows", "mi_render_clear", "mi_render_cancel",
           "mi_render_film_size", "mi_render_read_film", "mi_render_read_film_device", "mi_render_samples", "mi_render_stats",
           "mi_render_set_profiling", "mi_debug_intersect", "mi_debug_intersect_inst", "mi_debug_sobol", "mi_debug_camera_rays", "mi_debug_sincosf", "mi_debug_libm"]
HOST_EXPORTS = ["mi_host_last_error", "mi_host_create", "mi_host_create_devices", "mi_host_create_ex", "mi_host_destroy", "mi_host_preprocess", "mi_host_render", "mi_host_cancel", "mi_host_statistics"]


def build(force=False):
    """Compile the HIP extension in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")] + (["-B"] if force else []))
    return LIB_PATH


class Lib:
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise MiError(-1, f"{path} is missing: build it with `make -C mitsuba-im_amd/csrc` (there is no CPU fallback)")
        L = C.CDLL(path)
        self.L = L
        L.mi_last_error.restype = C.c_char_p
        vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float
        L.mi_set_sobol_tables.argtypes = [vp, u32, vp, vp]
        L.mi_scene_create.argtypes = [C.POINTER(vp)]
        L.mi_scene_destroy.argtypes = [vp]; L.mi_scene_destroy.restype = None
        L.mi_scene_set_triangles.argtypes = [vp, vp, vp, vp, vp, u32, u32, vp, u32]
        L.mi_scene_set_analytic.argtypes = [vp, vp, u32]
        L.mi_scene_set_instances.argtypes = [vp, vp, u32]
        L.mi_scene_set_media.argtypes = [vp, vp, u32, vp, u32, C.c_int32]
        L.mi_scene_set_material_tables.argtypes = [vp, vp, u32]
        L.mi_scene_set_textures.argtypes = [vp, vp, u32]
        L.mi_scene_set_texture_data.argtypes = [vp, vp, u32, vp, u64]
        L.mi_scene_set_materials.argtypes = [vp, vp, u32]
        L.mi_scene_set_emitters.argtypes = [vp, vp, u32]
        L.mi_scene_set_envmap.argtypes = [vp, vp, u32, u32, vp, f32]
        L.mi_scene_set_envmap_filter.argtypes = [vp, C.c_int32]
        L.mi_scene_set_camera.argtypes = [vp, vp, vp, f32, f32]
        L.mi_scene_set_film.argtypes = [vp, u32, u32, u32, f32, f32]
        L.mi_scene_commit.argtypes = [vp, u32]
        L.mi_scene_clone.argtypes = [vp, u32, C.POINTER(vp)]
        L.mi_scene_ray_intersect.argtypes = [vp, vp, u64, vp]
        L.mi_render_merge_film.argtypes = [vp, vp]
        L.mi_render_create.argtypes = [vp, C.POINTER(MiRenderParams), C.POINTER(vp)]
        L.mi_render_destroy.argtypes = [vp]; L.mi_render_destroy.restype = None
        L.mi_render_run.argtypes = [vp, MiTile, u32, u32]
        L.mi_render_run_rows.argtypes = [vp, MiTile, u32, u32, u32]
        L.mi_render_clear.argtypes = [vp]
        L.mi_render_cancel.argtypes = [vp]; L.mi_render_cancel.restype = None
        L.mi_render_film_size.argtypes = [vp, i32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
        L.mi_render_read_film.argtypes = [vp, i32, vp]
        L.mi_render_read_film_device.argtypes = [vp, i32, vp]
        L.mi_render_samples.argtypes = [vp, vp, u64, vp]
        L.mi_render_stats.argtypes = [vp, C.POINTER(MiStats)]
        L.mi_render_set_profiling.argtypes = [vp, i32]
        L.mi_debug_intersect.argtypes = [vp, vp, u64, i32, vp]
        L.mi_debug_intersect_inst.argtypes = [vp, vp, u64, i32, vp, vp]
        L.mi_debug_sobol.argtypes = [vp, vp, u64, u32, vp, vp]
        L.mi_debug_camera_rays.argtypes = [vp, vp, u64, vp]
        L.mi_debug_sincosf.argtypes = [vp, u64, vp]
        L.mi_debug_libm.argtypes = [i32, vp, vp, u64, vp]

    def check(self, rc):
        if rc != 0:
            raise MiError(rc, self.L.mi_last_error().decode())


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = Lib()
        load_sobol_tables(_LIB)
    return _LIB


def load_sobol_tables(L, path=SOBOL_PATH):
    """mitsuba-im_amd/data/sobol_tables.bin: Sobol' direction matrices (first 128 dimensions) and the van-der-Corput
    matrices for m = 1..16 -- numeric table DATA of the reference's sampler (src/samplers/sobolseq.cpp:33,106537,107241)."""
    with open(path, "rb") as f:
        assert f.read(8) == b"MISOBOL1"
        dims, rows = struct.unpack("<2I", f.read(8))
        m32 = np.frombuffer(f.read(dims * 52 * 4), dtype="<u4").copy()
        vdc = np.frombuffer(f.read(rows * 52 * 8), dtype="<u8").copy()
        vdci = np.frombuffer(f.read(rows * 52 * 8), dtype="<u8").copy()
    L.check(L.L.mi_set_sobol_tables(m32.ctypes.data, dims, vdc.ctypes.data, vdci.ctypes.data))


def _p(a):
    return None if a is None else a.ctypes.data


def device_sincosf(x):
    """glibcSincosf of pt_device.h on an array of floats -> (sin, cos)."""
    L = lib(); a = np.ascontiguousarray(x, np.float32).reshape(-1); out = np.zeros((len(a), 2), np.float32)
    L.check(L.L.mi_debug_sincosf(_p(a), len(a), _p(out))); return out[:, 0], out[:, 1]


LIBM_FUNCTIONS = {"expf": 0, "logf": 1, "powf": 2, "tanf": 3, "atanf": 4, "atan2f": 5, "acosf": 6}


def device_libm(name, x, y=None):
    """The device's restatement of glibc's `name` (libm_glibc.h, as the kernels call it) on arrays of floats."""
    L = lib(); a = np.ascontiguousarray(x, np.float32).reshape(-1); out = np.zeros(len(a), np.float32)
    b = None if y is None else np.ascontiguousarray(y, np.float32).reshape(-1)
    L.check(L.L.mi_debug_libm(LIBM_FUNCTIONS[name], _p(a), _p(b) if b is not None else None, len(a), _p(out))); return out


class Scene:
    """mi_scene handle filled from a flattened scene (mitsuba-im_amd/scenes.py)."""

    def __init__(self, sc, device=0):
        L = lib(); self.L = L; self.sc = sc
        h = C.c_void_p(); L.check(L.L.mi_scene_create(C.byref(h))); self.h = h
        shapes = (MiShape * len(sc.shapes))()
        for i, s in enumerate(sc.shapes):
            shapes[i] = MiShape(s["first_tri"], s["tri_count"], s["first_vert"], s["vert_count"], s["bsdf"], s["emitter"], (s["face_normals"] & 1) | ((s.get("has_uv", 0) & 1) << 1), s.get("group", 0))
        mats = (MiMaterial * len(sc.bsdfs))()
        for i, b in enumerate(sc.bsdfs):
            m = MiMaterial(b["type"], (b["twosided"] & 1) | ((b["sample_visible"] & 1) << 1) | ((b.get("nonlinear", 0) & 1) << 2) | ((b.get("aniso", 0) & 1) << 3) | ((b.get("texture", -1) + 1) << 8), b["distr"], b["alpha"])
            m.reflectance[:] = b["reflectance"]; m.eta[:] = b["eta"]; m.k[:] = b["k"]; m.specular[:] = b["specular"]
            mats[i] = m
        ems = (MiEmitter * max(1, len(sc.emitters)))()
        for i, e in enumerate(sc.emitters):
            em = MiEmitter(e["type"], e["shape"]); em.radiance[:] = e["radiance"]; em.weight = e["weight"]
            em.cutoff, em.beam = e.get("cutoff", 20.0), e.get("beam", 15.0)
            em.to_world[:] = np.asarray(e.get("to_world", np.eye(4)), np.float32).reshape(-1).tolist(); ems[i] = em
        L.check(L.L.mi_scene_set_triangles(h, _p(sc.pos), _p(sc.nrm), _p(sc.uv), _p(sc.idx), len(sc.pos), len(sc.idx), C.cast(shapes, C.c_void_p), len(sc.shapes)))
        recs = sc.get("analytic") or []
        if recs:
            an = (MiAnalytic * len(recs))()
            for i, a in enumerate(recs):
                r = MiAnalytic(a["type"], a["bsdf"], a["emitter"], a["flags"])
                r.to_world[:] = a["to_world"].reshape(-1).tolist(); r.to_object[:] = a["to_object"].reshape(-1).tolist()
                r.radius, r.length = a["radius"], a["length"]
                an[i] = r
            L.check(L.L.mi_scene_set_analytic(h, C.cast(an, C.c_void_p), len(recs)))
        insts = sc.get("instances") or []
        if insts:
            arr = (MiInstance * len(insts))()
            for i, a in enumerate(insts):
                r = MiInstance(a["group"]); r.to_world[:] = a["to_world"].reshape(-1).tolist(); r.to_object[:] = a["to_object"].reshape(-1).tolist(); arr[i] = r
            L.check(L.L.mi_scene_set_instances(h, C.cast(arr, C.c_void_p), len(insts)))
        media = sc.get("media") or []
        if media:                                       # mi_medium has the layout of the oracle's record (oracle/binding.py OrcMedium): 6 floats, uint, 2 floats, uint, float, uint
            buf = np.zeros(len(media), dtype=[("sigma_a", np.float32, 3), ("sigma_s", np.float32, 3), ("strategy", np.uint32), ("sampling_density", np.float32),
                                              ("medium_sampling_weight", np.float32), ("phase", np.uint32), ("g", np.float32), ("pad", np.uint32)])
            for i, m in enumerate(media):
                buf[i] = (m["sigma_a"], m["sigma_s"], m["strategy"], m["sampling_density"], m["medium_sampling_weight"], m["phase"], m["g"], 0)
            sm = np.ascontiguousarray(sc.shape_media, np.int32)
            L.check(L.L.mi_scene_set_media(h, buf.ctypes.data_as(C.c_void_p), len(media), sm.ctypes.data_as(C.c_void_p), len(sm), int(sc.sensor_medium)))
        L.check(L.L.mi_scene_set_materials(h, C.cast(mats, C.c_void_p), len(sc.bsdfs)))
        texs = sc.get("textures") or []
        if texs:
            ta = (MiTexture * len(texs))()
            for i, t in enumerate(texs):
                r = MiTexture(t["type"]); r.color0[:] = t["color0"]; r.color1[:] = t["color1"]; r.line_width = t["line_width"]
                r.uoffset, r.voffset, r.uscale, r.vscale = t["uoffset"], t["voffset"], t["uscale"], t["vscale"]
                r.wrap_u, r.wrap_v, r.filter, r.max_anisotropy, r.first_level, r.n_levels = t.get("wrap_u", 1), t.get("wrap_v", 1), t.get("filter", 3), t.get("max_anisotropy", 20.0), t.get("first_level", 0), t.get("n_levels", 0); ta[i] = r
            L.check(L.L.mi_scene_set_textures(h, C.cast(ta, C.c_void_p), len(texs)))
            if sc.get("texture_levels") is not None:
                L.check(L.L.mi_scene_set_texture_data(h, _p(sc.texture_levels), len(sc.texture_levels), _p(sc.texture_texels), len(sc.texture_texels)))
        if sc.get("material_tables") is not None:
            L.check(L.L.mi_scene_set_material_tables(h, _p(sc.material_tables), len(sc.material_tables)))
        L.check(L.L.mi_scene_set_emitters(h, C.cast(ems, C.c_void_p), len(sc.emitters)))
        if sc.envmap is not None:
            rgb = np.ascontiguousarray(sc.envmap["rgb"], np.float32); tw = np.ascontiguousarray(sc.envmap["to_world"], np.float32)
            L.check(L.L.mi_scene_set_envmap(h, _p(rgb), rgb.shape[1], rgb.shape[0], _p(tw), float(sc.envmap["scale"])))
            if sc.get("env_texture", 0):
                L.check(L.L.mi_scene_set_envmap_filter(h, int(sc.env_texture) - 1))
        s2c = np.ascontiguousarray(sc.sample_to_camera, np.float32); c2w = np.ascontiguousarray(sc.cam_to_world, np.float32)
        L.check(L.L.mi_scene_set_camera(h, _p(s2c), _p(c2w), sc.near, sc.far))
        L.check(L.L.mi_scene_set_film(h, sc.width, sc.height, sc.filter, sc.filter_radius, sc.filter_stddev))
        L.check(L.L.mi_scene_commit(h, device))

    def close(self):
        if getattr(self, "h", None):
            self.L.L.mi_scene_destroy(self.h); self.h = None

    def __del__(self):
        self.close()

    def ray_intersect(self, rays8):
        """Scene::rayIntersect for a batch of rays -> structured array of mi_intersection records."""
        rays8 = np.ascontiguousarray(rays8, np.float32).reshape(-1, 8)
        dt = np.dtype([("valid", "<u4"), ("t", "<f4"), ("p", "<f4", 3), ("ng", "<f4", 3), ("ns", "<f4", 3), ("s", "<f4", 3), ("tt", "<f4", 3), ("uv", "<f4", 2), ("wi", "<f4", 3), ("bary", "<f4", 2),
                       ("prim", "<u4"), ("instance", "<i4"), ("material", "<i4"), ("emitter", "<i4")])
        out = np.zeros(len(rays8), dt)
        self.L.check(self.L.L.mi_scene_ray_intersect(self.h, _p(rays8), len(rays8), out.ctypes.data)); return out

    # unit-level device entry points
    def intersect(self, rays8, any_hit=False, with_instance=False):
        rays8 = np.ascontiguousarray(rays8, np.float32).reshape(-1, 8); out = np.zeros((len(rays8), 4), np.float32)
        if with_instance:
            inst = np.zeros(len(rays8), np.int32)
            self.L.check(self.L.L.mi_debug_intersect_inst(self.h, _p(rays8), len(rays8), int(any_hit), _p(out), _p(inst))); return out, inst
        self.L.check(self.L.L.mi_debug_intersect(self.h, _p(rays8), len(rays8), int(any_hit), _p(out))); return out

    def sobol(self, px_py_k, ndims):
        a = np.ascontiguousarray(px_py_k, np.uint32).reshape(-1, 3); idx = np.zeros(len(a), np.uint64); vals = np.zeros((len(a), ndims), np.float32)
        self.L.check(self.L.L.mi_debug_sobol(self.h, _p(a), len(a), ndims, _p(idx), _p(vals))); return idx, vals

    def camera_rays(self, pos2):
        a = np.ascontiguousarray(pos2, np.float32).reshape(-1, 2); out = np.zeros((len(a), 8), np.float32)
        self.L.check(self.L.L.mi_debug_camera_rays(self.h, _p(a), len(a), _p(out))); return out


class Render:
    """mi_render handle: the integrator instance (MonteCarloIntegrator properties + sampler)."""

    def __init__(self, scene, max_depth=None, rr_depth=None, sampler=None, spp=None, seed=None, device=0, planes_per_batch=0,
                 strict_normals=None, hide_emitters=None, opacity=False, integrator=None):
        sc = scene.sc; L = scene.L; self.L = L; self.scene = scene
        p = MiRenderParams(sc.max_depth if max_depth is None else max_depth, sc.rr_depth if rr_depth is None else rr_depth,
                           sc.strict_normals if strict_normals is None else int(strict_normals),
                           sc.hide_emitters if hide_emitters is None else int(hide_emitters),
                           sc.sampler if sampler is None else sampler, sc.spp if spp is None else spp,
                           sc.seed if seed is None else seed, device, planes_per_batch, int(opacity), int(sc.get("integrator", 0) or 0) if integrator is None else int(integrator))
        self.params = p
        h = C.c_void_p(); L.check(L.L.mi_render_create(scene.h, C.byref(p), C.byref(h))); self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.L.mi_render_destroy(self.h); self.h = None

    def __del__(self):
        self.close()

    def run(self, tile=None, s0=0, s1=None, row_stride=1):
        sc = self.scene.sc
        t = MiTile(0, 0, sc.width, sc.height) if tile is None else MiTile(*tile)
        self.L.check(self.L.L.mi_render_run_rows(self.h, t, row_stride, s0, self.params.spp if s1 is None else s1))

    def clear(self):
        self.L.check(self.L.L.mi_render_clear(self.h))

    def cancel(self):
        self.L.L.mi_render_cancel(self.h)

    def set_profiling(self, on):
        self.L.check(self.L.L.mi_render_set_profiling(self.h, int(on)))

    def film_shape(self, layout=0):
        h, w, c, b = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.L.check(self.L.L.mi_render_film_size(self.h, layout, C.byref(h), C.byref(w), C.byref(c), C.byref(b)))
        return h.value, w.value, c.value, b.value

    def read_film(self, layout=0):
        h, w, c, _ = self.film_shape(layout); out = np.zeros((h, w, c), np.float32)
        self.L.check(self.L.L.mi_render_read_film(self.h, layout, _p(out))); return out

    def read_film_device(self, layout, device_ptr):
        self.L.check(self.L.L.mi_render_read_film_device(self.h, layout, C.c_void_p(device_ptr)))

    def samples(self, pairs):
        a = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 3); out = np.zeros((len(a), 3), np.float32)
        self.L.check(self.L.L.mi_render_samples(self.h, _p(a), len(a), _p(out))); return out

    def stats(self):
        s = MiStats(); self.L.check(self.L.L.mi_render_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in MiStats._fields_}


class HostIntegrator:
    """The product path as a reference user sees it: the C++ host mirror mi355::MIPathTracerHIP (csrc/integrator_host.h) behind its C shim (include/mi355pt_host.h).
    face "classic" = Integrator::render as Scene::render drives it (no target, no controls: one submission); "responsive" = ResponsiveIntegrator::render with a target
    film and a progress callback between submissions (src/mitsuba/im_render.cpp:103-222).  devices = HIP devices the film rows are spread over (entries may repeat)."""
    _CB = C.CFUNCTYPE(C.c_int, C.c_double, C.c_void_p)

    def __init__(self, scene, spp=None, devices=(0,), planes_per_batch=0, integrator=None, preview_interval_ms=-1.0, max_depth=None):
        L = scene.L.L; self.L = L; self.scene = scene; sc = scene.sc
        L.mi_host_create_ex.restype = C.c_void_p; L.mi_host_last_error.restype = C.c_char_p; L.mi_host_statistics.restype = C.c_char_p
        L.mi_host_create_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_int, C.c_double]
        L.mi_host_preprocess.argtypes = [C.c_void_p, C.c_void_p]; L.mi_host_destroy.argtypes = [C.c_void_p]; L.mi_host_statistics.argtypes = [C.c_void_p]
        L.mi_host_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), self._CB, C.c_void_p, C.c_int, C.c_int]
        dev = (C.c_uint32 * len(devices))(*devices)
        self.spp = sc.spp if spp is None else spp
        self.h = L.mi_host_create_ex(sc.max_depth if max_depth is None else max_depth, sc.rr_depth, sc.strict_normals, sc.hide_emitters, sc.sampler, self.spp, sc.seed, dev, len(devices),
                                     planes_per_batch, int(sc.get("integrator", 0) or 0) if integrator is None else int(integrator), float(preview_interval_ms))
        if not self.h:
            raise RuntimeError(L.mi_host_last_error().decode())
        if L.mi_host_preprocess(self.h, scene.h) != 0:
            raise RuntimeError(L.mi_host_last_error().decode())
        self.cont = C.c_int(1); self.abort = C.c_int(0); self.calls = 0

    def render(self, face="classic", target=None):
        """returns the integrator's return code (0 = all sample planes done)"""
        if face == "classic":
            return self.L.mi_host_render(self.h, None, None, None, self._CB(), None, 0, 1)

        def progress(spp, user):
            self.calls += 1; return 0
        cb = self._CB(progress)
        return self.L.mi_host_render(self.h, target.ctypes.data if target is not None else None, C.byref(self.cont), C.byref(self.abort), cb, None, 0, 1)

    def statistics(self):
        return (self.L.mi_host_statistics(self.h) or b"").decode()

    def close(self):
        if getattr(self, "h", None):
            self.L.mi_host_destroy(self.h); self.h = None

    def __del__(self):
        self.close()
