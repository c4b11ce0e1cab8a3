import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mi = importlib.import_module("mitsuba-im_amd")
detail = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
W, H = (3840, 2160) if len(sys.argv) < 3 else (1920, 1080)
t = time.time(); sc = mi.scenes.atrium(W, H, 64, detail=detail); t1 = time.time(); gs = mi.Scene(sc); t2 = time.time()
r = mi.Render(gs); r.run(s1=2); r.set_profiling(True); r.clear(); r.run(s1=8); st = r.stats(); n = W * H * 8
print("atrium detail %.2f: %d tris, commit %.2fs | %dx%d depth 8: %.1f Msamples/s | ms total %.1f extend %.1f shade %.1f shadow %.1f other %.1f | rays/sample %.2f shadow %.2f | %.2f Grays/s closest" % (
    detail, len(sc.idx), t2 - t1, W, H, n / st["render_ms"] / 1e3, st["render_ms"], st["extend_ms"], st["shade_ms"], st["shadow_ms"], st["other_ms"], st["rays"] / st["samples"], st["shadow_rays"] / st["samples"],
    st["rays"] / 2 / max(st["extend_ms"], 1e-9) / 1e6), flush=True)
