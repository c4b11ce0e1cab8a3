import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mi = importlib.import_module("mitsuba-im_amd")
t = time.time(); sc = mi.scenes.atrium(3840, 2160, 64); t1 = time.time(); gs = mi.Scene(sc); t2 = time.time()
print("atrium: %d triangles; generate %.1fs, commit (BVH build + upload) %.2fs" % (len(sc.idx), t1 - t, t2 - t1))
r = mi.Render(gs); r.run(s1=2); r.set_profiling(True); r.clear(); r.run(s1=8); st = r.stats(); n = 3840 * 2160 * 8
print("atrium 4K depth 8: %.1f Msamples/s | ms total %.1f extend %.1f shade %.1f shadow %.1f other %.1f | rays/sample %.2f shadow %.2f" % (
    n / st["render_ms"] / 1e3, st["render_ms"], st["extend_ms"], st["shade_ms"], st["shadow_ms"], st["other_ms"], st["rays"] / st["samples"], st["shadow_rays"] / st["samples"]))
