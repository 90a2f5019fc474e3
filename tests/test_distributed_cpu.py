"""N>1 path on CPU: world_size-2 gloo run of rt_amd.distributed.render_distributed (the same
partition + sum-reduce + resolve logic bench.py drives over RCCL), with the oracle standing in for
the per-rank renderer.  Checks that sharding by sample index reproduces the single-process render of
the same global sample set."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, spp_per_rank, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_ffi as O
    from golden.make_golden import golden_inputs
    from rt_amd import scenes
    from rt_amd.distributed import render_distributed
    cam = golden_inputs()["mixed_cam"]
    orc = O.OracleScene(scenes.mixed_test_scene())
    seen = {}

    def render_sum(spp, offset, stride, seed, out_sum):
        seen["part"] = (spp, offset, stride)
        mean = orc.render(cam, spp, seed=seed, sample_offset=offset, sample_stride=stride, nthreads=2)
        out_sum += torch.from_numpy(mean.reshape(-1) * float(spp))
        return None

    out = torch.zeros(cam.height * cam.width * 3, dtype=torch.float64)
    mean, _ = render_distributed(render_sum, out, spp_per_rank, seed=77)
    assert seen["part"] == (spp_per_rank, rank, world)
    if rank == 0:
        full = orc.render(cam, spp_per_rank * world, seed=77, nthreads=2)
        np.save(out_path, np.stack([mean.numpy().reshape(full.shape), full]))
    else:
        assert mean is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_sample_partition_over_gloo(tmp_path, world):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), 3, out), nprocs=world, join=True)
    got, full = np.load(out)
    np.testing.assert_allclose(got, full, rtol=1e-13, atol=1e-16)


def test_single_process_path_needs_no_process_group():
    from rt_amd.distributed import render_distributed

    def render_sum(spp, offset, stride, seed, out_sum):
        assert (spp, offset, stride) == (5, 0, 1)
        out_sum += 10.0
        return "stats"

    mean, st = render_distributed(render_sum, torch.zeros(6, dtype=torch.float64), 5)
    assert st == "stats" and torch.equal(mean, torch.full((6,), 2.0, dtype=torch.float64))


def test_partition_weak_and_strong_cover_the_sample_set_exactly_once():
    """partition(): in both modes the ranks' index sets r, r+N, ... are disjoint and their union is 0 .. total-1 — the
    sample set of the single-GPU render the multi-GPU image must equal (src/camera.rs:316-324)."""
    from rt_amd.distributed import partition
    for scaling in ("weak", "strong"):
        for world in (1, 2, 3, 4, 8):
            for spp in (1, 2, 5, 8, 256, 1023, 1024):
                seen = []
                total_ = None
                for r in range(world):
                    mine, off, stride, total = partition(spp, r, world, scaling)
                    total_ = total
                    seen += [off + k * stride for k in range(mine)]
                assert total_ == (spp * world if scaling == "weak" else spp)
                assert sorted(seen) == list(range(total_)), (scaling, world, spp)
    import pytest
    with pytest.raises(ValueError):
        partition(4, 0, 2, "sideways")
