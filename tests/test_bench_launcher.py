"""bench.py --gpus N must start N ranks by itself when no launcher set WORLD_SIZE.  Driven here on
CPU at world size 2 over gloo with bench.py's synthetic stand-in renderer (RT_BENCH_STANDIN=1: launcher, sample
partition, sum-reduce and resolve are the real code; nothing is measured), and under torch.distributed.run."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RT_BENCH_STANDIN"] = "1"
    return env


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_world_size"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["data"] == "standin-selftest" and d["standin_max_abs_err"] < 1e-12
    assert d["config"]["partition"] == "sample index mod 2"


def test_bench_single_rank_and_world_size_mismatch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["rccl_world_size"] == 1
    env = dict(_env(), WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


def test_bench_under_torch_distributed_run():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_world_size"] == 2


def test_bench_strong_scaling_splits_a_fixed_job():
    """--scaling strong: the job's spp are fixed and split over the ranks, ceil((spp - r) / N) for rank r — even (4 over
    2), uneven (5 over 3: 2 + 2 + 1) and with a rank that gets nothing (2 over 3); the mean is that of the single-rank
    job in every case (BASELINE config 4's 1/2/4/8-GPU runs; the split of rt_render_multi, src/camera.rs:316-324)."""
    for n, spp in ((2, 4), (3, 5), (3, 2)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
                            "--scaling", "strong", "--spp", str(spp)], env=_env(), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        d = _json_line(r.stdout)
        assert d["scaling"] == "strong" and d["n_gpus"] == n and d["rccl_world_size"] == n
        assert d["config"]["global_spp"] == spp and d["standin_max_abs_err"] < 1e-12


def test_bench_launcher_ends_the_other_ranks_when_one_fails():
    """A rank that dies before the rendezvous leaves the others waiting for it: the launcher terminates them and exits
    non-zero instead of sitting out the collective's timeout."""
    import time
    env = dict(_env(), RT_BENCH_STANDIN_FAIL_RANK="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "rank(s) failed" in r.stderr
    assert time.time() - t0 < 120
