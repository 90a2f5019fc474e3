"""Sky coefficient sets handed across the boundary as `rt_sky`.

The reference builds `SkyState::new(&SkyParams::default())` (src/hittable.rs:35) from the
Hosek-Wilkie RGB dataset embedded in the `hw-skymodel` crate.  That crate (and its dataset) is not
under /root/reference and cannot be fetched, so the true 27+3 floats are unobtainable here
(SURVEY.md §8c).  They are therefore *inputs* of the boundary (a Rust host passes
`SkyState::raw()`); synthetic runs use the placeholder set below.

SKY_SYNTH_V1 is NOT the Hosek-Wilkie coefficient set.  It is a fixed, hand-picked set with the same
shape (9 params per channel, order as consumed by hw_skymodel::rgb::SkyState::radiance, plus one
radiance scale per channel) chosen so that the tonemapped sky stays inside [0.35, 0.85] for every
direction, i.e. images stay in the [0,1] range the reference's 8-bit conversion requires.
"""
import numpy as np

from . import _abi

SKY_SYNTH_V1_PARAMS = np.array([
    [-0.60, -0.20, 0.60, 0.80, -1.0, 0.30, 0.010, 0.55, 0.50],   # R
    [-0.58, -0.18, 0.70, 0.75, -1.1, 0.32, 0.010, 0.60, 0.50],   # G
    [-0.55, -0.15, 0.90, 0.65, -1.2, 0.35, 0.008, 0.70, 0.50],   # B
], dtype=np.float32)
SKY_SYNTH_V1_RADIANCES = np.array([1.6, 1.9, 2.6], dtype=np.float32)


def sky_synth_v1(sun=(0.0, 0.0, 1.0)):
    """rt_sky record with the placeholder coefficients and World::build's sun direction
    (src/hittable.rs:38: Vec3::new(0,0,1).normalize())."""
    s = _abi.one(_abi.sky_dt)
    s["params"][0] = SKY_SYNTH_V1_PARAMS.reshape(-1)
    s["radiances"][0] = SKY_SYNTH_V1_RADIANCES
    s["sun"][0] = sun
    return s
