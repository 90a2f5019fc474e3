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
    # entries: base (in-tree library), r1 (the round-1 tree under build/old/r1, if present: the reference point on THIS box —
    # boxes differ by several per cent), every library under build/variants, and NAME=VALUE[,NAME=VALUE] environment settings
    # given before the bench arguments (up to a "--")
    envs = []
    if "--" in args:
        k = args.index("--")
        envs, args = args[:k], args[k + 1:]
    names = ["base"] + (["r1"] if os.path.isdir(os.path.join(ROOT, "build", "old", "r1")) else []) + \
            (sorted(os.listdir(VDIR)) if os.path.isdir(VDIR) else []) + ["env:" + e for e in envs]
    for rnd in range(2):
        for n in names:
            env = dict(os.environ)
            bench = os.path.join(ROOT, "bench.py")
            cwd = ROOT
            if n == "r1":
                cwd = os.path.join(ROOT, "build", "old", "r1")
                bench = os.path.join(cwd, "bench.py")
            elif n.startswith("env:"):
                for kv in n[4:].split(","):
                    a, _, b = kv.partition("=")
                    env[a] = b
            elif n != "base":
                env["RT_HIP_LIB"] = os.path.join(VDIR, n, "librt_hip.so")
            r = subprocess.run([sys.executable, bench, "--no-cpu-baseline", "--no-alone"] + args, env=env, cwd=cwd,
                               capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(n, "FAILED", r.stderr[-400:])
                continue
            j = json.loads(line[-1])
            rf = j["roofline"]
            print(f"{n:34s} r{rnd} value={j['value']:8.1f} ms/step={j['ms_per_step']:8.2f} "
                  f"bounce_rays/s={rf['rays_per_s'] / 1e9:.3f}G packet_rays/s={j['roofline_packet']['rays_per_s'] / 1e9:.3f}G "
                  f"nodes/seg={rf.get('nodes_per_segment', 0):.2f} prims/seg={rf.get('prims_per_segment', 0):.2f} build_s={j['scene']['bvh_build_s']:.2f}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
