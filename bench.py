#!/usr/bin/env python3
"""bench.py -- Msamples/s of the unidirectional path-tracing hot path on MI355X (BASELINE.json metric).

Workload (config[1] of BASELINE.json): synthetic Cornell box (mitsuba-im_amd/scenes.py), 1920x1080, Sobol sampler, 256 spp,
maxDepth 8, rrDepth 5, box filter.  One "step" = one complete render of that frame (clear film -> trace all sample planes ->
accumulate film -> read the raw film into a device tensor -> framebuffer reduce to rank 0).  Scene upload / BVH build happen
once before the timed region; inputs are resident in HBM when timing starts.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): weak scaling -- every rank traces the same number of camera
samples as the single-GPU job: the film rows are interleaved over the ranks (mitsuba-im_amd/dist.py) and the sample count grows to
256*N spp, so each rank still traces 1920*1080*256 samples; value = all samples of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on the render stream inside the timed region)
and `cpu_baseline` (the reference itself -- oracle/_ref/harness -- or, where that build is absent, the oracle port).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def cpu_baseline(scenes, width, height, max_depth):
    """Reference CPU path tracer (or the oracle port) on a bounded sample of the same workload, host cores of this box."""
    cores = min(os.cpu_count() or 1, 16)
    spp = 2
    sc = scenes.cornell_box(width, height, spp, sampler=scenes.SAMPLER_SOBOL, max_depth=max_depth)
    sample = f"Cornell box {width}x{height}, sobol, {spp} spp of 256, maxDepth {max_depth} ({width * height * spp} samples)"
    harness = os.path.join(ROOT, "oracle", "_ref", "harness")
    if os.path.exists(harness):
        try:
            with tempfile.TemporaryDirectory() as tmp:
                path = os.path.join(tmp, "s.miscene"); scenes.save_scene(sc, path)
                subprocess.run([harness, path, "image", str(cores), os.path.join(tmp, "o")], cwd=os.path.dirname(harness),
                               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
                stats = dict(l.split(" ", 1) for l in open(os.path.join(tmp, "o_stats.txt")).read().splitlines()[:4])
                return {"value": round(float(stats["msamples_per_s"]), 4), "unit": "Msamples/s", "cores": cores, "kind": "reference",
                        "sample": sample + "; SamplingIntegrator::renderBlock over 32x32 blocks"}
        except Exception as e:  # fall through to the port
            print(f"[bench] reference harness failed ({e}); using the oracle port", file=sys.stderr)
    import oracle
    oracle.build()
    orc = oracle.Oracle(sc)
    t = time.perf_counter(); orc.render_image(threads=cores); dt = time.perf_counter() - t
    return {"value": round(width * height * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port", "sample": sample}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--max-depth", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fast-math", action="store_true", help="use the fast-arithmetic kernel build (fused multiply-add, approximate divide/sqrt); default: strict IEEE kernels, bit-identical to the oracle")
    ap.add_argument("--no-stage-timing", action="store_true", help="skip the per-kernel HIP events (roofline becomes null)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    mi = importlib.import_module("mitsuba-im_amd")
    mi_dist = importlib.import_module("mitsuba-im_amd.dist")
    S = mi.scenes

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 through `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    use_dist = world > 1 or os.environ.get("MI355PT_FORCE_DIST") == "1"      # the latter: exercise the RCCL calls on a single GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    total_spp = args.spp * world                       # weak scaling: per-rank samples fixed
    sc = S.cornell_box(args.width, args.height, total_spp, sampler=S.SAMPLER_SOBOL, max_depth=args.max_depth)
    scene = mi.Scene(sc, device=local)
    render = mi.Render(scene, device=local, fast_math=args.fast_math)
    render.set_profiling(not args.no_stage_timing)
    tile, row_stride = mi_dist.interleaved_rows(sc.width, sc.height, rank, world)     # rank k owns rows k, k + N, ...
    fh, fw, fc, _ = render.film_shape(0)
    film = torch.empty((fh, fw, fc), dtype=torch.float32, device="cuda")

    def step():
        render.clear()
        render.run(tile=tile, s0=0, s1=total_spp, row_stride=row_stride)
        render.read_film_device(0, film.data_ptr())
        if use_dist:
            dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)      # the one exchange step of the path (mitsuba-im_amd/dist.py: reduce_film)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage = {"extend_ms": 0.0, "shade_ms": 0.0, "shadow_ms": 0.0, "other_ms": 0.0, "render_ms": 0.0, "extend_launches": 0, "extend_launches_all": 0}
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = render.stats()
        for k in stage:
            stage[k] += st[k]
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda"); dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dt = float(tmax.item())

    st = render.stats()                                 # counters accumulate since the last clear = one step
    samples_all = sc.width * sc.height * total_spp
    value = samples_all * args.steps / dt / 1e6
    out = {
        "metric": "Msamples/sec at 1920x1080 path-trace, max depth 8", "value": round(value, 2), "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "arithmetic": "fast (fma, approximate divide/sqrt)" if args.fast_math else "strict IEEE (bit-identical to the oracle)",
        "config": {"workload": f"Cornell box (synthetic, 32 triangles), {sc.width}x{sc.height}, sobol sampler, {total_spp} spp "
                               f"({args.spp} per GPU-share), maxDepth {args.max_depth}, rrDepth 5, box filter, "
                               f"1xMI355X wavefront path tracer per rank", "parallelism": f"tiles{world}"},
    }
    if rank == 0:
        rays, shadow = st["rays"], st["shadow_rays"]; n = st["samples"] or 1
        out["counters"] = {"rays_per_sample": round(rays / n, 4), "shadow_rays_per_sample": round(shadow / n, 4),
                           "avg_path_length": round(st["path_length_sum"] / n, 4)}
        if not args.no_stage_timing and stage["extend_launches"]:
            # algorithmic bytes per kernel (DESIGN.md "bytes per unit"): extend 48 B/ray; shade 68 B read per ray + 68 B written per
            # surviving path + 48 B per shadow record; shadow 48 B per record.  Counters are per step; stage times are summed over steps.
            survivors = max(rays - n, 0)
            per_step = {"extend": 48.0 * rays, "shade": 68.0 * rays + 68.0 * survivors + 48.0 * shadow, "shadow": 48.0 * shadow}
            ms = {"extend": stage["extend_ms"], "shade": stage["shade_ms"], "shadow": stage["shadow_ms"]}
            dom = max(ms, key=ms.get)
            # batches alternate between two path pools / HIP streams; the stage events sit on the first stream, so `ms` covers `timed` of
            # the `total` launches (all launches move the same bytes on average): bytes per launch = step bytes / total launches
            shrink = (args.max_depth - 1) / args.max_depth if dom == "shadow" else 1.0
            launches = stage["extend_launches"] * shrink; total = stage["extend_launches_all"] * shrink
            bytes_per_launch = per_step[dom] * args.steps / max(total, 1)
            achieved = bytes_per_launch * launches / (ms[dom] * 1e-3) / 1e9
            # measured HBM bytes per launch of that kernel: rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) of this very
            # workload, committed under profiles/ (PMC cannot be collected from inside the timed run); valid for the default 1080p batches only
            traffic = None
            try:
                if (args.width, args.height, args.max_depth) == (1920, 1080, 8):
                    pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]["k_" + dom]
                    traffic = round(pmc["hbm_bytes_per_launch"], 1)
            except Exception:
                traffic = None
            out["roofline"] = {"bound": "hbm", "kernel": "k_" + dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "avg_launch_us": round(ms[dom] * 1e3 / max(launches, 1), 2), "launches": int(launches),
                               "algorithmic_bytes_per_launch": round(bytes_per_launch, 1), "streams": 2 if total > launches else 1,
                               "note": "launch durations are measured while the other stream's kernels share the CUs"}
            seg_bytes = 288.0 * rays + 16.0 * n
            out["pipeline"] = {"bytes_per_sample": round(seg_bytes / n, 1), "achieved_GBs": round(seg_bytes * args.steps / (stage["render_ms"] * 1e-3) / 1e9, 2),
                               "frac_of_hbm_peak": round(seg_bytes * args.steps / (stage["render_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                               "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in stage.items() if k.endswith("_ms")}}
        else:
            out["roofline"] = None
        out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(S, args.width, args.height, args.max_depth)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
