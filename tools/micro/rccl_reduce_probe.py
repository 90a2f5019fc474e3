"""Host and device time of the frame exchange (dist.reduce of the f64 sum buffer) with one RCCL rank.
   python tools/micro/rccl_reduce_probe.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for dtype in (torch.float64, torch.float32):
    x = torch.ones(1920 * 1080 * 3, dtype=dtype, device="cuda")
    for name, op in (("reduce", lambda: dist.reduce(x, dst=0)), ("all_reduce", lambda: dist.all_reduce(x)),
                     ("copy (no collective)", lambda: x.clone())):
        for _ in range(3):
            op()
        torch.cuda.synchronize()
        host, dev = [], []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            t0 = time.perf_counter()
            op()
            host.append(time.perf_counter() - t0)
            b.record()
            torch.cuda.synchronize()
            dev.append(a.elapsed_time(b))
        print(f"{str(dtype):14s} {name:22s} host call {1e3 * sum(host) / len(host):7.3f} ms   device {sum(dev) / len(dev):7.3f} ms", flush=True)
dist.destroy_process_group()
