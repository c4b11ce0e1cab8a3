"""CPU, world_size 2, gloo: the N>1 path -- row-band tiles per rank, global-coordinate sampling, one sum-reduce of the raw film --
must reproduce the single-rank film.  The renderer plugged into mitsuba-im_amd/dist.py here is the oracle (checker); on the GPU
box bench.py plugs in the HIP path."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mi_dist = importlib.import_module("mitsuba-im_amd.dist"); scenes = importlib.import_module("mitsuba-im_amd.scenes")
    import oracle
    sc = scenes.cornell_box(64, 37, 4)          # odd height: ragged bands
    orc = oracle.Oracle(sc); b = orc.border
    film = np.zeros((sc.height + 2 * b, sc.width + 2 * b, 5), np.float32)

    def render_tile(tile, s0, s1):
        f, _ = orc.render_image(s0, s1, tile[1], tile[3], threads=1); film[...] += f

    tile = mi_dist.render_sharded(render_tile, sc.width, sc.height, rank, world, 0, sc.spp)
    t = torch.from_numpy(film); mi_dist.reduce_film(t, dist, dst=0)
    # second layout: rows interleaved over the ranks (what bench.py uses on the GPUs)
    film2 = np.zeros_like(film); (tx0, ty0, tx1, ty1), stride = mi_dist.interleaved_rows(sc.width, sc.height, rank, world)
    for y in range(ty0, ty1, stride):
        f, _ = orc.render_image(0, sc.spp, y, y + 1, threads=1); film2 += f
    t2 = torch.from_numpy(film2); mi_dist.reduce_film(t2, dist, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "film_interleaved.npy"), t2.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "film.npy"), t.numpy())
    np.save(os.path.join(out_dir, f"tile{rank}.npy"), np.array(tile))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_tiles_reduce_to_single_rank_film(tmp_path):
    sys.path.insert(0, ROOT)
    import importlib
    import oracle
    scenes = importlib.import_module("mitsuba-im_amd.scenes")
    oracle.build()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sc = scenes.cornell_box(64, 37, 4)
    full, _ = oracle.Oracle(sc).render_image(threads=2)
    got = np.load(tmp_path / "film.npy")
    t0, t1 = np.load(tmp_path / "tile0.npy"), np.load(tmp_path / "tile1.npy")
    assert t0[1] == 0 and t0[3] == t1[1] and t1[3] == sc.height
    assert np.allclose(got, full, rtol=1e-6, atol=1e-7)
    inner = (got[1:-1, 1:-1].view(np.uint32) == full[1:-1, 1:-1].view(np.uint32))
    assert inner.mean() > 0.999
    got2 = np.load(tmp_path / "film_interleaved.npy")
    assert np.allclose(got2, full, rtol=1e-6, atol=1e-7) and (got2[1:-1, 1:-1].view(np.uint32) == full[1:-1, 1:-1].view(np.uint32)).mean() > 0.99


def test_bench_launcher_self_spawns_ranks():
    """`python bench.py --gpus 2` called plainly must start its own ranks (the driver's scaling run calls it that way): the parent spawns two
    children with RANK / WORLD_SIZE set, they rendezvous (gloo here, RCCL on the GPUs), interleave the film rows, reduce, and rank 0 prints ONE JSON line.
    --dry-run stops before the GPU work, everything up to it is the code path of the real run."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo", "--scaling", "strong"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rows_covered_once"] and out["scaling"] == "strong"
    # under an external launcher (torch.distributed.run sets WORLD_SIZE) the script must NOT spawn again
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo"],
                       env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
