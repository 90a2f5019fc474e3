"""bench.py --gpus N must start N ranks by itself when no launcher set WORLD_SIZE.  Driven here on
CPU at world size 2 over gloo with bench.py's synthetic stand-in renderer (RT_BENCH_STANDIN=1: launcher, sample
partition, sum-reduce and resolve are the real code; nothing is measured), and under torch.distributed.run."""
import json
import os
import socket
import subprocess
import sys

import pytest

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


@pytest.mark.gpu
def test_ranks_sharing_the_gpu_render_the_single_rank_image(tmp_path):
    """The N > 1 path with REAL renders on a one-GPU box (RT_BENCH_SHARE_GPU=1: every rank on device 0, gloo, the sum-reduce
    staged through host memory — RCCL refuses two ranks on one device): launcher, sample partition, per-rank renders
    through the C ABI, exchange, resolve.  Weak scaling at 2 ranks (6 + 6 samples) and strong scaling at 3 (12 = 4 + 4 + 4)
    and at 2 with an uneven split (7 = 4 + 3) must give the single-rank image of the same sample set: the same paths
    (equal segment counts) summed in another order (f64: 1e-12 relative)."""
    import numpy as np
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RT_BENCH_STANDIN")}
    common = ["--workload", "cover", "--width", "256", "--height", "144", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-alone"]

    def run(tag, extra, share):
        path = str(tmp_path / (tag + ".npy"))
        env = dict(base, RT_BENCH_DUMP_IMAGE=path)
        if share:
            env["RT_BENCH_SHARE_GPU"] = "1"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + common, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        return _json_line(r.stdout), np.load(path)

    for total, n, args in ((12, 2, ["--spp", "6"]), (12, 3, ["--spp", "12", "--scaling", "strong"]), (7, 2, ["--spp", "7", "--scaling", "strong"])):
        one, img1 = run("one%d" % total, ["--gpus", "1", "--spp", str(total)], False)
        d, img = run("n%d_%d" % (n, total), ["--gpus", str(n)] + args, True)
        assert d["n_gpus"] == n and d["config"]["global_spp"] == total and "rehearsal" in d and "rehearsal" not in one
        assert d["config"]["partition"].startswith("sample index mod %d" % n)
        assert abs(d["segments_per_sample"] - one["segments_per_sample"]) < 1e-12
        assert img.shape == img1.shape and np.all(np.isfinite(img))
        np.testing.assert_allclose(img, img1, rtol=1e-12, atol=0)
