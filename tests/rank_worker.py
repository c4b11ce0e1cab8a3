"""One rank of the N > 1 layout of the hot path, as a process of its own: `python tests/rank_worker.py RANK WORLD PORT OUT_DIR [DEVICE]`.

The rank renders ITS rows (row k, k + N, ... of the film: mitsuba-im_amd/dist.py interleaved_rows) through the HIP path (C-ABI, mi_render_run_rows), then takes part
in the path's one exchange step, the sum-reduce of the raw film onto rank 0 (the reference's merge: Film::put(block) under a mutex, src/librender/renderproc.cpp:142-149).
tests/test_gpu_ranks.py starts two of these on ONE card (the test box has a single GPU, and RCCL refuses two ranks on one device), so the exchange runs over gloo on
host tensors; bench.py uses RCCL on device tensors.  Everything before the exchange -- a rank > 0 driving the kernels with a row offset and stride, a scene replica
per process, global-coordinate sampling -- is the code the 8-GPU run executes."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    device = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    mi = importlib.import_module("mitsuba-im_amd"); mi_dist = importlib.import_module("mitsuba-im_amd.dist")
    from tests.test_gpu_ranks import rank_scene
    sc = rank_scene(mi)
    scene = mi.Scene(sc, device=device); render = mi.Render(scene, device=device)
    tile, stride = mi_dist.interleaved_rows(sc.width, sc.height, rank, world)
    render.clear(); render.run(tile=tile, s0=0, s1=sc.spp, row_stride=stride)
    film = torch.from_numpy(np.ascontiguousarray(render.read_film(0)))
    st = render.stats()
    own = film.clone()
    mi_dist.reduce_film(film, dist, dst=0)
    counters = torch.tensor([st["samples"], st["rays"], st["shadow_rays"], st["path_length_sum"]], dtype=torch.int64)
    dist.reduce(counters, dst=0, op=dist.ReduceOp.SUM)
    np.save(os.path.join(out_dir, f"own{rank}.npy"), own.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "film.npy"), film.numpy())
        with open(os.path.join(out_dir, "counters.json"), "w") as f:
            json.dump(dict(zip(("samples", "rays", "shadow_rays", "path_length_sum"), counters.tolist())), f)
    dist.barrier(); dist.destroy_process_group()
    render.close(); scene.close()


if __name__ == "__main__":
    main()
