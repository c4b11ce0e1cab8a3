"""BVH depth + throughput of a BVH scene (SCENE env: atrium / bunny), for the traversal-stack A/B (MI355PT_STACK24=1 skips the 20-entry variant)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mi = importlib.import_module("mitsuba-im_amd")
which = os.environ.get("SCENE", "atrium"); spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if which == "bunny":
    X = importlib.import_module("mitsuba-im_amd.xml_scene"); sc = X.load_scene(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "meshes", "bunny_box.xml")); sc.width, sc.height = 1920, 1080
    sc.sample_to_camera = mi.scenes.sample_to_camera(sc.xfov, sc.near, sc.far, sc.width / sc.height); sc.spp = spp
else:
    sc = mi.scenes.atrium(3840, 2160, spp)
gs = mi.Scene(sc); r = mi.Render(gs, spp=spp)
r.run(s1=min(4, spp)); r.set_profiling(True); r.clear(); r.run(s1=spp); st = r.stats()
n = sc.width * sc.height * spp
print("%s stack24=%s: %.1f Msamples/s | ms: total %.1f extend %.1f shade %.1f shadow %.1f" % (which, os.environ.get("MI355PT_STACK24", "0"), n / st["render_ms"] / 1e3, st["render_ms"], st["extend_ms"], st["shade_ms"], st["shadow_ms"]), flush=True)
