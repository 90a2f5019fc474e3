#!/usr/bin/env python3
"""A/B harness for kernel experiments.
  build:  python tools/ab_variants.py build name1:-DFOO name2:"-DBAR -DBAZ" ...   (here, CPU)
  run:    python tools/ab_variants.py run [bench args]                            (on the GPU box)
Variants live in build/variants/<name>/librt_hip.so; `base` is the in-tree library."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "build", "variants")


def build(specs):
    import __graft_entry__ as g
    for spec in specs:
        name, _, flags = spec.partition(":")
        d = os.path.join(VDIR, name)
        os.makedirs(d, exist_ok=True)
        objs = []
        for s in g.SOURCES:
            o = os.path.join(d, os.path.splitext(s)[0] + ".o")
            cmd = [g._hipcc()] + g.HIPCC_FLAGS + flags.split() + (["-x", "hip"] if s.endswith(".cpp") else []) + \
                  ["-c", os.path.join(g.CSRC, s), "-o", o]
            subprocess.run(cmd, check=True, cwd=g.CSRC)
            objs.append(o)
        subprocess.run([g._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "librt_hip.so")] + objs, check=True)
        for o in objs:
            os.remove(o)
        print("built", name, flags)


def run(args):
    names = ["base"] + sorted(os.listdir(VDIR)) if os.path.isdir(VDIR) else ["base"]
    for rnd in range(2):
        for n in names:
            env = dict(os.environ)
            if n != "base":
                env["RT_HIP_LIB"] = os.path.join(VDIR, n, "librt_hip.so")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-alone"] + args, env=env,
                               capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(n, "FAILED", r.stderr[-400:])
                continue
            j = json.loads(line[-1])
            rf = j["roofline"]
            print(f"{n:24s} r{rnd} value={j['value']:8.1f} ms/step={j['ms_per_step']:8.2f} trace={j['trace_fraction_of_step'] * j['ms_per_step']:7.2f}ms "
                  f"shade={j['shade_fraction_of_step'] * j['ms_per_step']:7.2f}ms bounce_rays/s={rf['rays_per_s'] / 1e9:.3f}G "
                  f"packet_rays/s={j['roofline_packet']['rays_per_s'] / 1e9:.3f}G", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
