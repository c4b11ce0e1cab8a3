"""Run the drop-in plugin through the reference's responsive driver on one golden scene and keep its output (debugging aid)."""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tests.golden.make_golden import golden_scenes
S = importlib.import_module("mitsuba-im_amd.scenes")
name = sys.argv[1]; out = os.path.join(ROOT, "gpurun_out", name + "_hip")
os.makedirs(os.path.dirname(out), exist_ok=True)
path = out + ".miscene"; S.save_scene(golden_scenes()[name], path)
h = os.path.join(ROOT, "oracle", "_ref", "harness")
subprocess.run([h, path, "responsive", "path_hip", "-1", out], cwd=os.path.dirname(h), check=True, timeout=300)
os.remove(path)
