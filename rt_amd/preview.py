"""Progressive multi-pass rendering: the caller contract of the reference's preview loop
(`window::render_thread`, src/window.rs:224-325) without the window.

Per pass k the reference calls `camera.render_pixel(&world, x, y, n_k)` for every pixel
(src/window.rs:277) — i.e. Halton points 0..n_k are reused by every pass while the scatter
randomness is fresh — and blends `new * n_k/total + old * (1 - n_k/total)` into an 8-bit LINEAR
frame buffer (`as_rgb_linear`, :292-310).  Here a pass is one `render_image(spp=n_k,
sample_offset=0, seed=seed+k)`; the blend and the 8-bit quantisation are kept (quantise=True) or
replaced by a float running mean (quantise=False, what a new host should use).
`render` is injectable so the contract can be tested on CPU with a stand-in renderer.
"""
import time

import numpy as np

# num_samples_at_pass (src/window.rs:233-247) as run lengths: 237 passes, 40 055 spp in total
SCHEDULE_RUNS = [(1, 1), (2, 1), (4, 1), (8, 2), (16, 2), (32, 2), (64, 84), (128, 18), (256, 126)]
SCHEDULE = [n for n, reps in SCHEDULE_RUNS for _ in range(reps)]


def round_half_away(x):
    """f64::round: ties away from zero (numpy's rint rounds ties to even)."""
    r = np.rint(x)
    tie = np.abs(x - np.trunc(x)) == 0.5
    return np.where(tie, np.trunc(x) + np.sign(x), r)


def as_rgb_linear(color):
    """Vec3Ext::as_rgb_linear (src/vec3.rs:44-69): (c * 255).round(); the reference panics outside
    [0, 1], this clamps."""
    return round_half_away(np.clip(color, 0.0, 1.0) * 255.0).astype(np.uint8)


def progressive(render, width, height, n_passes, seed=0, quantise=True, on_pass=None):
    """render(spp, seed) -> (H, W, 3) float64 mean radiance of one pass.
    Returns the final frame (uint8 linear if quantise else float64) and per-pass stats."""
    passes = SCHEDULE[:n_passes]
    frame8 = np.zeros((height, width, 3), dtype=np.uint8)
    framef = np.zeros((height, width, 3), dtype=np.float64)
    total, stats, t_first = 0, [], time.time()
    for k, n in enumerate(passes):
        t0 = time.time()
        new = render(n, seed + k)
        total += n
        new_ratio = float(n) / float(total)
        old_ratio = 1.0 - new_ratio
        if quantise:
            old = frame8.astype(np.float64) / 255.0
            frame8 = as_rgb_linear(new * new_ratio + old * old_ratio)
        else:
            framef = new * new_ratio + framef * old_ratio
        dt, dt_all = time.time() - t0, time.time() - t_first
        st = {"pass": k + 1, "samples": n, "total_samples": total,
              "mrays_per_s": n * width * height / 1e6 / max(dt, 1e-9),          # src/window.rs:315-324: rays == camera samples
              "overall_mrays_per_s": total * width * height / 1e6 / max(dt_all, 1e-9)}
        stats.append(st)
        if on_pass:
            on_pass(st, frame8 if quantise else framef)
    return (frame8 if quantise else framef), stats


def render_with_preview(camera, world, n_passes=9, seed=0, quantise=True, on_pass=None):
    """window::render_with_preview's render thread on the GPU path (headless)."""
    return progressive(lambda n, s: camera.render_image(world, spp=n, seed=s), camera.width, camera.height,
                       n_passes, seed, quantise, on_pass)


def render_with_preview_device(camera, world, n_passes=9, seed=0, quantise=True, on_pass=None):
    """The same loop with the frame kept on the GPU: one rt_render_pass per sweep (render + blend + 8-bit linear RGBA
    read-back, no allocation and no float frame crossing PCIe per sweep).  Returns the (H, W, 3) uint8 frame (the float
    frame with quantise=False) and the per-sweep stats, like render_with_preview."""
    total, stats, t_first = 0, [], time.time()
    frame = None
    for k, n in enumerate(SCHEDULE[:n_passes]):
        t0 = time.time()
        total += n
        rgba, mean = camera.render_pass(world, n, total, seed=seed + k, quantise=quantise, with_mean=True)
        frame = rgba[..., :3] if quantise else mean
        dt, dt_all = time.time() - t0, time.time() - t_first
        st = {"pass": k + 1, "samples": n, "total_samples": total,
              "mrays_per_s": n * camera.width * camera.height / 1e6 / max(dt, 1e-9),
              "overall_mrays_per_s": total * camera.width * camera.height / 1e6 / max(dt_all, 1e-9)}
        stats.append(st)
        if on_pass:
            on_pass(st, frame)
    return frame, stats
