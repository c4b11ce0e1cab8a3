"""fast-arithmetic kernels vs the strict ones (quick look before writing the tests)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mi = importlib.import_module("mitsuba-im_amd")
rng = np.random.default_rng(3)
for name, sc in [("cornell", mi.scenes.cornell_box(1920, 1080, 8)), ("veach", mi.scenes.veach_mis(96, 54, 16)), ("atrium", mi.scenes.atrium(96, 54, 16, detail=0.08, env_size=(64, 32)))]:
    n = 50000; pairs = np.stack([rng.integers(0, sc.width, n), rng.integers(0, sc.height, n), rng.integers(0, sc.spp, n)], 1).astype(np.uint32)
    gs = mi.Scene(sc); ref = mi.Render(gs).samples(pairs); got = mi.Render(gs, fast_math=True).samples(pairs)
    err = np.abs(got - ref).max(1) / (np.abs(ref).max(1) + 1e-6)
    print(name, "fast vs precise: median %.1e p99 %.1e max %.1e  >1e-4: %d of %d  >1e-2: %d" % (np.median(err), np.quantile(err, .99), err.max(), (err > 1e-4).sum(), n, (err > 1e-2).sum()), flush=True)
sc = mi.scenes.cornell_box(480, 270, 64); gs = mi.Scene(sc)
a = mi.Render(gs); a.run(); fa = a.read_film(2); b = mi.Render(gs, fast_math=True); b.run(); fb = b.read_film(2)
print("cornell 480x270x64spp developed image rel L2 fast vs precise: %.2e" % (np.linalg.norm(fa - fb) / np.linalg.norm(fa)))
