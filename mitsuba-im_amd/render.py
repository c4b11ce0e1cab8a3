"""Command-line front end: render a Mitsuba XML scene on the MI355X path tracer (what `mitsuba scene.xml` does for the `path` integrator).

    python -m mitsuba-im_amd.render scene.xml [-o out.exr|out.pfm|out.npy|out.png] [-D name=value ...] [--spp N] [--sampler sobol|independent] [--device K]

The image written is the developed film (sum / weight, linear RGB, as HDRFilm::develop would hand to its writer); `.exr` (FLOAT channels, ZIP), `.pfm` and `.npy`
keep the linear values; `.png` / `.jpg` get ldrfilm's default sRGB encoding (imageio.py).  There is no CPU fallback: without the HIP library / a GPU this exits with an error.
"""
import argparse
import sys
import time

import numpy as np


def write_image(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    if path.endswith(".npy"):
        np.save(path, rgb)
    elif path.endswith(".pfm"):
        with open(path, "wb") as f:
            f.write(b"PF\n%d %d\n-1.0\n" % (rgb.shape[1], rgb.shape[0]))
            f.write(rgb[::-1].astype("<f4").tobytes())
    elif path.endswith(".exr"):
        from . import imageio
        imageio.write_exr(path, rgb)
    elif path.endswith((".png", ".jpg", ".jpeg")):
        from . import imageio
        imageio.write_ldr(path, rgb)
    else:
        raise SystemExit(f"unsupported output format: {path} (.exr, .pfm, .npy, .png, .jpg)")


def main(argv=None):
    from . import xml_scene
    from .api import Scene, Render, MiError
    ap = argparse.ArgumentParser(prog="python -m mitsuba-im_amd.render", description=__doc__.split("\n\n")[0])
    ap.add_argument("scene")
    ap.add_argument("-o", "--output", default=None)
    ap.add_argument("-D", dest="defines", action="append", default=[], metavar="name=value")
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--sampler", choices=["sobol", "independent"], default=None, help="replace the scene's sampler plugin (keeps its sampleCount)")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    params = {}
    for d in a.defines:
        if "=" not in d:
            raise SystemExit(f"-D expects name=value, got {d!r}")
        k, v = d.split("=", 1); params[k] = v
    try:
        t0 = time.perf_counter()
        sc = xml_scene.load_scene(a.scene, params, sampler=a.sampler)
        if a.spp is not None:
            sc.spp = a.spp
        t1 = time.perf_counter()
        scene = Scene(sc, device=a.device)
        render = Render(scene, device=a.device)
        t2 = time.perf_counter()
        render.run()
        rgb = render.read_film(2)
        t3 = time.perf_counter()
    except (xml_scene.SceneError, MiError, OSError) as e:
        print(f"error: {e}", file=sys.stderr)
        return 1
    out = a.output or (a.scene.rsplit(".", 1)[0] + ".exr")          # hdrfilm's default fileFormat (src/films/hdrfilm.cpp: openexr)
    write_image(out, rgb)
    n = sc.width * sc.height * sc.spp
    print(f"{sc.name}: {sc.width}x{sc.height}, {sc.spp} spp, {len(sc.idx)} triangles; load {t1 - t0:.2f} s, upload+BVH {t2 - t1:.2f} s, "
          f"render {t3 - t2:.3f} s ({n / (t3 - t2) / 1e6:.1f} Msamples/s) -> {out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
