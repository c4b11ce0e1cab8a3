"""Quick perf probe: SCENE (default cbox_shapes) at 1080p, per-stage HIP-event times (ms per 1080p plane-batch set)."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mi = importlib.import_module("mitsuba-im_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sc = getattr(mi.scenes, os.environ.get("SCENE", "cbox_shapes"))(1920, 1080, spp)
gs = mi.Scene(sc); r = mi.Render(gs, planes_per_batch=int(os.environ.get("PLANES", "0")))
r.run(s1=8); r.set_profiling(True)
r.clear(); r.run(s1=spp); st = r.stats()
n = 1920 * 1080 * spp
print("grid=%s packet=%s planes=%s: %.1f Msamples/s | ms: total %.1f extend %.1f shade %.1f shadow %.1f other %.1f" % (
    "seg=%s e=%s s=%s sh=%s" % tuple(os.environ.get(k, "-") for k in ("MI355PT_SEGMENTS", "MI355PT_GRID_EXTEND", "MI355PT_GRID_SHADE", "MI355PT_GRID_SHADOW")), os.environ.get("MI355PT_NO_PACKET", "0") != "1", os.environ.get("PLANES", "auto"),
    n / st["render_ms"] / 1e3, st["render_ms"], st["extend_ms"], st["shade_ms"], st["shadow_ms"], st["other_ms"]), flush=True)
