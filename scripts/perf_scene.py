"""Throughput of one synthetic scene: python scripts/perf_scene.py <generator> [width height spp [json kwargs]]  (e.g. cbox_materials 1920 1080 32; atrium 3840 2160 16 '{"detail": 0.25}')"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mi = importlib.import_module("mitsuba-im_amd")
name = sys.argv[1]; W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920; H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080; spp = int(sys.argv[4]) if len(sys.argv) > 4 else 32
import json
kw = json.loads(sys.argv[5]) if len(sys.argv) > 5 else {}
sc = getattr(mi.scenes, name)(W, H, spp, **kw)
gs = mi.Scene(sc); r = mi.Render(gs)
r.run(s1=min(8, spp)); r.set_profiling(True); r.clear(); r.run(); st = r.stats(); n = W * H * spp
print("%s%s %dx%d x%d [%d tris]: %.1f Msamples/s | ms total %.1f extend %.1f shade %.1f shadow %.1f other %.1f | rays/sample %.2f" % (
    name, kw or "", W, H, spp, len(sc.idx), n / st["render_ms"] / 1e3, st["render_ms"], st["extend_ms"], st["shade_ms"], st["shadow_ms"], st["other_ms"], st["rays"] / st["samples"]), flush=True)
