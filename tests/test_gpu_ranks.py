"""GPU (MI355X): the N > 1 layout of the hot path with the HIP path inside every rank (tests/test_distributed_gloo.py covers the same layout on CPU with the oracle
plugged in).  Two or three rank PROCESSES share the box's one card; each builds its own scene replica, renders its interleaved rows through the C-ABI
(mi_render_run_rows with a row offset = its rank and a stride = the world size) and joins the one sum-reduce of the raw film (gloo here: RCCL refuses two ranks on
one device; bench.py on an 8-GPU node runs the same steps with the reduce over RCCL).  Checked against the unsplit film of one process: bit for bit under the box
filter wherever a film row received samples of one rank only (the sum adds zeros there), and the per-rank ray counters add up to the unsplit job's."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


FLOOR = 1.0        # measured on MI355X (gpurun_out/r03f/ranks.log -> profiles/r03_rank_processes.log): 1.000000 for 2 and for 3 ranks -- the reduced film IS the unsplit film


def rank_scene(mi):
    """non-diffuse materials + Sobol indices of a non-power-of-two film, odd height: ragged row shares"""
    return mi.scenes.veach_mis(322, 181, 8, max_depth=12)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("world", [2, 3])
def test_rank_processes_on_the_hip_path(mi, tmp_path, world):
    port = str(29600 + (os.getpid() + world) % 2000)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), str(r), str(world), port, str(tmp_path)],
                              cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    try:
        logs = [p.communicate(timeout=300)[0] for p in procs]
    finally:
        for p in procs:          # a rank that died before the rendezvous leaves the others waiting: end exactly the processes started here
            if p.poll() is None:
                p.kill()
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    sc = rank_scene(mi)
    gs = mi.Scene(sc); r = mi.Render(gs); r.clear(); r.run(s0=0, s1=sc.spp); full = r.read_film(0); st = r.stats()
    got = np.load(tmp_path / "film.npy")
    share = float((bits(got) == bits(full)).mean()); print(f"[rank films] world {world}: bit-equal share {share:.6f}")
    assert got.shape == full.shape and np.allclose(got, full, rtol=1e-6, atol=1e-7) and share >= FLOOR
    # every rank produced its rows and nothing else: its own film is zero on the others' rows, and no rank's share is empty
    b = (full.shape[0] - sc.height) // 2
    for k in range(world):
        own = np.load(tmp_path / f"own{k}.npy")[b:b + sc.height]
        assert own[k::world, :, 4].sum() > 0
        for j in range(world):
            if j != k:
                assert own[j::world, :, 4].sum() <= 1e-3 * own[k::world, :, 4].sum()      # only samples that sit exactly on a row edge reach a neighbour's row
    cnt = json.load(open(tmp_path / "counters.json"))
    assert cnt == {k: st[k] for k in cnt}


def test_bench_two_ranks_share_the_card():
    """bench.py's N = 2 path end to end on the one-GPU box: the self-spawning launcher, two ranks on the HIP path (MI355PT_SHARE_DEVICE=1 maps both to device 0),
    the film reduce (gloo through the host: the rehearsal backend), barrier + max-over-ranks timing, rank 0's JSON line with the fixed job as `value` and the weak job beside it."""
    env = dict(os.environ, MI355PT_SHARE_DEVICE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "1", "--spp", "8", "--width", "640",
                          "--height", "360", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0 and line["weak"]["value"] > 0 and line["weak"]["spp"] == 16
    assert "REHEARSAL" in line["config"]["workload"] and 3.5 < line["counters"]["rays_per_sample"] < 5.0
