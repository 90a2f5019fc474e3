#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory with the CPU oracle.

The reference ships no fixtures and cannot be run here (Rust, no toolchain), so these vectors are
outputs of oracle/rt_oracle.c (brute-force closest hit, f64) on seeded inputs — they pin the oracle
and the HIP path against regressions and against each other, not against the Rust binary
("parity unpinned", oracle/rt_oracle.h).   Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_ffi as O  # noqa: E402
from rt_amd import scenes  # noqa: E402
from rt_amd.api import Camera  # noqa: E402


def golden_inputs():
    """Shared by the generator and the tests: the seeded inputs of every fixture."""
    k = np.arange(3000)
    o = np.stack([-8.0 + 16.0 * scenes.hash_u01(101, k, c) for c in range(3)], -1)
    o[:, 2] = 0.2 + 6.0 * scenes.hash_u01(101, k, 7)
    d = np.stack([2.0 * scenes.hash_u01(102, k, c) - 1.0 for c in range(3)], -1)
    return {
        "cover_cam": scenes.cover_camera(64, 36, spp=1, max_depth=50, defocus=False),
        "mixed_rays": np.concatenate([o, d], -1),
        "mixed_cam": Camera.new((9.0, -7.0, 4.0), (0.0, 0.0, 0.8), (0.0, 0.0, 1.0), 10.0, 0.3, 32, 24, 2, 12, 35.0),
        "hf_cam": scenes.heightfield_camera(48, 27, 1, 50),
    }


def main():
    gi = golden_inputs()
    cover = O.OracleScene(scenes.cover_scene())
    mixed = O.OracleScene(scenes.mixed_test_scene())
    hf = O.OracleScene(scenes.heightfield_scene(quads=48))
    h0 = cover.trace_primary(gi["cover_cam"], 0, jitter=True, mode=O.BRUTE)
    h1 = cover.trace_primary(gi["cover_cam"], 0, jitter=False, mode=O.BRUTE)
    hm = mixed.hit(gi["mixed_rays"], mode=O.BRUTE)
    hh = hf.trace_primary(gi["hf_cam"], 3, jitter=True, mode=O.BRUTE)
    img, st = mixed.render(gi["mixed_cam"], 2, seed=3, mode=O.BRUTE, with_stats=True)
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"),
                        cover_jitter_prim=h0["prim"], cover_jitter_t=h0["t"], cover_debug_prim=h1["prim"], cover_debug_t=h1["t"],
                        mixed_prim=hm["prim"], mixed_t=hm["t"], hf_prim=hh["prim"], hf_t=hh["t"],
                        mixed_render=img, mixed_render_segments=np.array([int(st["ray_segments"])]))
    print("wrote golden_v1.npz")


if __name__ == "__main__":
    main()
