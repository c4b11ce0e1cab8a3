"""Quick GPU bring-up check (not a test): GPU vs oracle on samples + small film, and a first timing."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mi = importlib.import_module("mitsuba-im_amd")
import oracle
S = mi.scenes
rng = np.random.default_rng(7)
for name, sc in [("cornell_sobol", S.cornell_box(1920, 1080, 8)), ("cornell_indep", S.cornell_box(1920, 1080, 8, sampler=S.SAMPLER_INDEPENDENT)),
                 ("closed_box", S.closed_box())]:
    n = 20000
    pairs = np.stack([rng.integers(0, sc.width, n), rng.integers(0, sc.height, n), rng.integers(0, sc.spp, n)], 1).astype(np.uint32)
    orc = oracle.Oracle(sc); ref = orc.render_samples(pairs)["li"]
    gs = mi.Scene(sc); r = mi.Render(gs)
    got = r.samples(pairs)
    eq = (got.view(np.uint32) == ref.view(np.uint32)).all(1)
    print(name, "bit-exact samples: %d / %d" % (eq.sum(), n), "max abs diff", np.abs(got - ref).max(), flush=True)
    if not eq.all():
        bad = np.where(~eq)[0][:5]; print("  first bad:", pairs[bad], got[bad], ref[bad])
sc = S.cornell_box(96, 54, 16)
orc = oracle.Oracle(sc); film_o, cnt_o = orc.render_image(threads=8)
gs = mi.Scene(sc); r = mi.Render(gs); r.run(); film_g = r.read_film(0); st = r.stats()
print("small film: max abs diff", np.abs(film_g - film_o).max(), "rel L2", np.linalg.norm(film_g[..., :3] - film_o[..., :3]) / np.linalg.norm(film_o[..., :3]))
print("counters oracle", cnt_o, "gpu", st["rays"], st["shadow_rays"], st["path_length_sum"])
sc = S.cornell_box(1920, 1080, 64)
gs = mi.Scene(sc); r = mi.Render(gs)
r.run(s1=8)
for spp in (8, 64):
    r.clear(); t = time.time(); r.run(s1=spp); dt = time.time() - t; st = r.stats()
    print("1080p x %d spp: %.1f ms wall, %.1f ms device -> %.1f Msamples/s; rays/sample %.2f shadow/sample %.2f" % (
        spp, dt * 1e3, st["render_ms"], 1920 * 1080 * spp / st["render_ms"] / 1e3, st["rays"] / st["samples"], st["shadow_rays"] / st["samples"]), flush=True)
r.set_profiling(True); r.clear(); r.run(s1=16); st = r.stats()
print("profile 16spp: total %.1f ms extend %.1f shade %.1f shadow %.1f other %.1f (launches %d)" % (st["render_ms"], st["extend_ms"], st["shade_ms"], st["shadow_ms"], st["other_ms"], st["extend_launches"]))
