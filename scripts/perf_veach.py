import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mi = importlib.import_module("mitsuba-im_amd")
sc = mi.scenes.veach_mis(1920, 1080, 512)
gs = mi.Scene(sc); r = mi.Render(gs)
r.run(s1=8); r.set_profiling(True); r.clear(); r.run(s1=32); st = r.stats(); n = 1920 * 1080 * 32
print("veach 1080p depth 12: %.1f Msamples/s | ms total %.1f extend %.1f shade %.1f shadow %.1f other %.1f | rays/sample %.2f" % (
    n / st["render_ms"] / 1e3, st["render_ms"], st["extend_ms"], st["shade_ms"], st["shadow_ms"], st["other_ms"], st["rays"] / st["samples"]))
