"""Host logic of the BVH builders (bvh_build.cpp: binned SAH with worker threads, f32 / u16 / BVH4 / BVH8 node formats),
compiled WITHOUT HIP under AddressSanitizer + UBSan and checked structurally (tests/native/bvh_check.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bvh_builders_under_sanitizers(tmp_path):
    exe = str(tmp_path / "bvh_check")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
           os.path.join(ROOT, "tests", "native", "bvh_check.cpp"), os.path.join(ROOT, "rt_amd", "csrc", "bvh_build.cpp"), os.path.join(ROOT, "rt_amd", "csrc", "bvh_wide8.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None) if False else None
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "OK" in r.stdout
