"""GPU tests of the ABI entries around the render path: the device output stage (f4), the progressive sweep entry
(f2: rt_render_pass against rt_amd/preview.py's restatement of window::render_thread) and the single-process
multi-GPU render (rt_render_multi: worker threads + RCCL sum-reduce; one device on the test box)."""
import ctypes as C

import numpy as np
import pytest

from rt_amd import World, _abi, preview, scenes
from rt_amd._lib import RtError, check, lib
from rt_amd.api import encode_gamma8

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mixed():
    desc = scenes.mixed_test_scene()
    return desc, World.build(desc)


def test_device_output_stage_matches_host_routine(mixed, oracle):
    """rt_encode_gamma8_device on a device-resident frame == rt_encode_gamma8 on the host copy, byte for byte away from
    rounding boundaries of round(c^(1/2.2) * 255) (the two pow() are glibc's and the GPU's), never more than one apart,
    and == the oracle's orc_gamma8 of the same frame."""
    from hip_mem import DeviceBuffer, synchronize
    desc, world = mixed
    cam = scenes.cam1(160, 120, max_depth=8)
    img = cam.render_image(world, spp=6, seed=5)
    special = np.array([0.0, 1.0, -0.25, 7.5, np.nan, 0.5, 1e-300, 0.21763764082403103])
    frame = np.concatenate([img.reshape(-1), special])
    d_in = DeviceBuffer.from_array(frame)
    d_out = DeviceBuffer(frame.size)
    check(lib().rt_encode_gamma8_device(d_in.ptr, frame.size, d_out.ptr, 0, None))
    synchronize()
    got = d_out.to_array(np.uint8, frame.size)
    host = encode_gamma8(frame)
    orc8 = np.zeros(frame.size, dtype=np.uint8)
    oracle.lib().orc_gamma8(_abi.ptr(np.ascontiguousarray(np.nan_to_num(frame, nan=0.0))), frame.size, _abi.ptr(orc8))
    scaled = np.clip(np.nan_to_num(frame, nan=0.0), 0.0, 1.0) ** (1.0 / 2.2) * 255.0
    safe = np.abs(scaled - np.floor(scaled) - 0.5) > 1e-6
    assert safe.mean() > 0.999
    assert np.array_equal(got[safe], host[safe]) and np.array_equal(got[safe], orc8[safe])
    assert (np.abs(got.astype(int) - host.astype(int)) <= 1).all()
    assert got[-8:-3].tolist() == [0, 255, 0, 255, 0]           # 0, 1, negative, > 1, NaN: clamped, no panic


def test_render_pass_matches_the_preview_loop(mixed):
    """rt_render_pass sweep by sweep against preview.progressive over rt_render (same seeds): the float frame within
    rounding of the blend, the 8-bit linear frame equal except where the float value sits on a rounding boundary,
    in the reference's quantised accumulation and in the float one."""
    desc, world = mixed
    cam = scenes.cam1(96, 72, max_depth=8)
    for quantise in (True, False):
        frames = []
        ref, _ = preview.progressive(lambda n, s: cam.render_image(world, spp=n, seed=s), 96, 72, 6, seed=11, quantise=quantise,
                                     on_pass=lambda st, f: frames.append(np.array(f)))
        total = 0
        for k, n in enumerate(preview.SCHEDULE[:6]):
            total += n
            rgba, mean, st = cam.render_pass(world, n, total, seed=11 + k, quantise=quantise, with_mean=True, with_stats=True)
            assert int(st["samples"]) == 96 * 72 * n and (rgba[..., 3] == 255).all()
            if quantise:
                diff = np.abs(rgba[..., :3].astype(int) - frames[k].astype(int))
                assert diff.max() <= 1 and (diff != 0).mean() < 1e-3, (k, diff.max(), (diff != 0).mean())
            else:
                np.testing.assert_allclose(mean, frames[k], rtol=1e-12, atol=1e-15)
                assert np.abs(rgba[..., :3].astype(int) - preview.as_rgb_linear(frames[k]).astype(int)).max() <= 1
    dev, stats = preview.render_with_preview_device(cam, world, n_passes=4, seed=2, quantise=True)
    host, _ = preview.render_with_preview(cam, world, n_passes=4, seed=2, quantise=True)
    assert dev.shape == host.shape == (72, 96, 3) and np.abs(dev.astype(int) - host.astype(int)).max() <= 1
    assert [s["total_samples"] for s in stats] == [1, 3, 7, 15]


def test_render_pass_argument_checks(mixed):
    desc, world = mixed
    cam = scenes.cam1(32, 24, max_depth=4)
    L = lib()
    rgba = np.zeros((24, 32, 4), np.uint8)
    assert L.rt_render_pass(world._h, _abi.ptr(cam.rec), 0, 0, 0, 1, _abi.ptr(rgba), None, None) == _abi.RT_ERR_INVALID
    assert L.rt_render_pass(world._h, _abi.ptr(cam.rec), 4, 2, 0, 1, _abi.ptr(rgba), None, None) == _abi.RT_ERR_INVALID
    assert L.rt_render_pass(world._h, _abi.ptr(cam.rec), 1, 1, 0, 8, _abi.ptr(rgba), None, None) == _abi.RT_ERR_INVALID
    assert L.rt_render_pass(world._h, _abi.ptr(cam.rec), 1, 1, 0, 1, _abi.ptr(rgba), None, None) == _abi.RT_OK
    # the running total must continue where the frame stands
    assert L.rt_render_pass(world._h, _abi.ptr(cam.rec), 2, 5, 0, 1, _abi.ptr(rgba), None, None) == _abi.RT_ERR_INVALID
    assert L.rt_render_pass(world._h, _abi.ptr(cam.rec), 2, 3, 0, 1, None, None, None) == _abi.RT_OK


def test_render_multi_one_device_equals_rt_render(mixed):
    """rt_render_multi with one device goes through the worker thread, ncclCommInitAll and ncclReduce and must give
    rt_render's image bit for bit; the same samples split by hand over two 'ranks' sum to it."""
    desc, world = mixed
    cam = scenes.cam1(80, 60, max_depth=8)
    ref, rst = cam.render_image(world, spp=7, seed=9, with_stats=True)
    img, st = cam.render_image_multi([world], spp=7, seed=9, with_stats=True)
    assert np.array_equal(img, ref)
    assert int(st[0]["ray_segments"]) == int(rst["ray_segments"]) and int(st[0]["samples"]) == 80 * 60 * 7
    again = cam.render_image_multi([world], spp=7, seed=9)                 # the communicator is reused
    assert np.array_equal(again, ref)
    # what two devices would render: sample indices 0,2,4,6 and 1,3,5
    a = cam.render_image(world, spp=4, seed=9, sample_offset=0, sample_stride=2)
    b = cam.render_image(world, spp=3, seed=9, sample_offset=1, sample_stride=2)
    np.testing.assert_allclose((4.0 * a + 3.0 * b) / 7.0, ref, rtol=1e-12, atol=1e-15)
    L = lib()
    out = np.zeros((60, 80, 3))
    o = cam.opts(spp=7, seed=9)
    handles = (C.c_void_p * 2)(world._h, world._h)
    assert L.rt_render_multi(handles, 2, _abi.ptr(cam.rec), _abi.ptr(o), _abi.ptr(out), None) == _abi.RT_ERR_INVALID
    assert b"same device" in L.rt_last_error()
    assert L.rt_render_multi(handles, 0, _abi.ptr(cam.rec), _abi.ptr(o), _abi.ptr(out), None) == _abi.RT_ERR_INVALID


def test_render_multi_f32_exchange_and_reuse_of_the_scene_buffers(mixed):
    """RT_OPT_REDUCE_F32: the per-device sums cross the exchange as f32, everything before it is f64 — so
    with one device the frame is rt_render's to f32 rounding of the per-pixel SUM.  The sum buffers and the stream
    belong to the scene: a second call at another frame size (smaller, then larger) reuses / regrows them."""
    desc, world = mixed
    for w, h, spp in ((80, 60, 7), (48, 36, 5), (128, 96, 3)):
        cam = scenes.cam1(w, h, max_depth=8)
        ref = cam.render_image(world, spp=spp, seed=9)
        img = cam.render_image_multi([world], spp=spp, seed=9, reduce_f32=True)
        np.testing.assert_allclose(img, ref, rtol=2.0 ** -23, atol=1e-30)
        assert not np.array_equal(img, ref)                              # it did go through f32
        assert np.array_equal(cam.render_image_multi([world], spp=spp, seed=9), ref)
    L = lib()
    cam = scenes.cam1(80, 60, max_depth=8)
    o = cam.opts(spp=7, seed=9)
    o["flags"] = 0x80
    out = np.zeros((60, 80, 3))
    handles = (C.c_void_p * 1)(world._h)
    assert L.rt_render_multi(handles, 1, _abi.ptr(cam.rec), _abi.ptr(o), _abi.ptr(out), None) == _abi.RT_ERR_INVALID


def test_two_threads_render_two_scenes_while_a_third_traces(mixed, oracle):
    """include/rt_hip.h promises handles that are shareable across threads and calls serialised per device — what the
    reference's Arc<World> sharing between the render thread and the window thread needs (src/window.rs:48-49, :76-82).
    Two host threads render two scenes on the one device and a third runs closest-hit batches on the first scene (the
    allocation-free entry stays out: its scratch array is shared per device by contract): every result equals the
    serial one bit for bit."""
    import threading
    desc, world = mixed
    desc2 = scenes.mixed_test_scene(seed=5, n_spheres=30, quads=10)
    world2 = World.build(desc2)
    cam = scenes.cam1(128, 96, max_depth=12)
    kw = dict(spp=8, seed=3, max_paths_in_flight=128 * 96 * 2)       # four passes, several pipes
    k = np.arange(50_000)
    rays = np.zeros(len(k), _abi.ray_dt)
    rays["o"] = np.stack([-6.0 + 12.0 * scenes.hash_u01(91, k, c) for c in range(3)], -1)
    rays["d"] = np.stack([2.0 * scenes.hash_u01(92, k, c) - 1.0 for c in range(3)], -1)
    want = (cam.render_image(world, **kw), cam.render_image(world2, **kw), world.hit(rays))
    got, errors = {}, []

    def run(name, fn, rounds):
        try:
            for r in range(rounds):
                got[(name, r)] = fn()
        except Exception as e:                                        # noqa: BLE001
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=run, args=("a", lambda: cam.render_image(world, **kw), 3)),
               threading.Thread(target=run, args=("b", lambda: cam.render_image(world2, **kw), 3)),
               threading.Thread(target=run, args=("c", lambda: world.hit(rays), 6))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for r in range(3):
        assert np.array_equal(got[("a", r)], want[0]) and np.array_equal(got[("b", r)], want[1])
    for r in range(6):
        assert np.array_equal(got[("c", r)]["prim"], want[2]["prim"]) and np.array_equal(got[("c", r)]["t"], want[2]["t"])


def test_streams_the_host_made_before_do_not_change_the_image(monkeypatch):
    """A render with several passes in flight puts its pipes on streams of the scene's own and checks once per scene that
    they sit on different hardware queues (ensure_pipe_streams); how many streams the host made before decides how many
    idle streams that check parks.  None of it may show in the image."""
    desc = scenes.mixed_test_scene(seed=5, n_spheres=30, quads=10)
    cam = scenes.cover_camera(160, 90, spp=8, max_depth=20, defocus=True)
    kw = dict(spp=8, seed=9, max_paths_in_flight=160 * 90 * 2, with_stats=True)      # four passes
    base, st0 = cam.render_image(World.build(desc), **kw)
    assert int(st0["passes"]) == 4
    try:
        hip = C.CDLL("libamdhip64.so")
    except OSError:
        hip = None
    made = []
    for extra in (1, 2):
        if hip is not None:
            for _ in range(extra):
                s = C.c_void_p()
                assert hip.hipStreamCreate(C.byref(s)) == 0
                made.append(s)
        img, st = cam.render_image(World.build(desc), **kw)
        assert np.array_equal(img, base) and int(st["ray_segments"]) == int(st0["ray_segments"]), extra
    monkeypatch.setenv("RT_HIP_STREAM_PROBE", "0")
    img, _ = cam.render_image(World.build(desc), **kw)
    assert np.array_equal(img, base)
    for s in made:
        hip.hipStreamDestroy(s)


def test_pipelined_frames_equal_the_waited_ones(mixed):
    """RT_OPT_PIPELINED: rt_render_accumulate_device queues a frame behind the one before and returns; the caller's stream
    waits for the frame, rt_scene_sync for all of them.  Three frames (different seeds) queued back to back on a stream
    — each behind an asynchronous clear of its buffer on that stream, the work the frame's first sum has to wait for —
    must leave exactly the sums of the three waited renders, and the statistics of the calls plus rt_scene_sync's must
    add up to the waited ones.  Then: an entry point that needs the workspaces while a frame is queued (World.hit), a
    frame of another shape behind a queued one, and a scene destroyed with a frame still queued."""
    from hip_mem import DeviceBuffer, hip, synchronize
    desc, world = mixed
    L, h = lib(), hip()
    h.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    cam = scenes.cam1(128, 72, max_depth=12)
    npix, nval = 128 * 72, 128 * 72 * 3
    kw = dict(spp=16, max_paths_in_flight=2 * npix)              # 8 passes of 2 samples: four in flight, every pipe used twice

    def waited(seed, **over):
        buf = DeviceBuffer(nval * 8)
        buf.fill(0)
        st = _abi.one(_abi.stats_dt)
        o = cam.opts(seed=seed, **dict(kw, **over))
        check(L.rt_render_accumulate_device(world._h, _abi.ptr(cam.rec), _abi.ptr(o), buf.ptr, None, _abi.ptr(st)))
        return buf.to_array(np.float64, nval), st[0]

    ref = [waited(s) for s in (1, 2, 3)]
    assert int(ref[0][1]["passes"]) == 8
    stream = C.c_void_p()
    assert h.hipStreamCreate(C.byref(stream)) == 0
    bufs = [DeviceBuffer(nval * 8) for _ in range(3)]
    for b in bufs:
        b.fill(0xFF)                                             # garbage: the clear queued on the stream is what zeroes it
    sts = []
    for s, b in zip((1, 2, 3), bufs):
        assert h.hipMemsetAsync(b.ptr, 0, nval * 8, stream) == 0
        st = _abi.one(_abi.stats_dt)
        o = cam.opts(seed=s, pipelined=True, **kw)
        check(L.rt_render_accumulate_device(world._h, _abi.ptr(cam.rec), _abi.ptr(o), b.ptr, stream, _abi.ptr(st)))
        sts.append(st[0])
    sts.append(world.sync())
    assert h.hipStreamSynchronize(stream) == 0
    for (want, _), b in zip(ref, bufs):
        assert np.array_equal(b.to_array(np.float64, nval), want)
    for key in ("samples", "ray_segments", "primary_segments", "bounce_segments", "tail_segments", "passes"):
        assert sum(int(x[key]) for x in sts) == sum(int(st[key]) for _, st in ref), key
    assert int(world.sync()["ray_segments"]) == 0               # nothing queued: returns at once, all zero

    # an entry point that uses the workspaces while a frame is queued waits for it first; the frame still completes
    rays = np.zeros((64, 6))
    rays[:, 0:3] = (13.0, 2.0, 3.0)
    rays[:, 3:6] = np.random.default_rng(5).normal(size=(64, 3))
    hits_before = world.hit(rays)
    assert h.hipMemsetAsync(bufs[0].ptr, 0, nval * 8, stream) == 0
    o = cam.opts(seed=1, pipelined=True, **kw)
    check(L.rt_render_accumulate_device(world._h, _abi.ptr(cam.rec), _abi.ptr(o), bufs[0].ptr, stream, None))
    assert np.array_equal(world.hit(rays), hits_before)
    # a frame of another shape behind a queued one (waits for it, then runs on its own), then a waited frame
    assert h.hipMemsetAsync(bufs[1].ptr, 0, nval * 8, stream) == 0
    o = cam.opts(seed=2, pipelined=True, **kw)
    check(L.rt_render_accumulate_device(world._h, _abi.ptr(cam.rec), _abi.ptr(o), bufs[1].ptr, stream, None))
    assert h.hipMemsetAsync(bufs[2].ptr, 0, nval * 8, stream) == 0
    o = cam.opts(seed=3, pipelined=True, **dict(kw, spp=10))
    check(L.rt_render_accumulate_device(world._h, _abi.ptr(cam.rec), _abi.ptr(o), bufs[2].ptr, stream, None))
    want10, _ = waited(3, spp=10)                                # (a waited call: drains what is queued first)
    assert h.hipStreamSynchronize(stream) == 0
    assert np.array_equal(bufs[0].to_array(np.float64, nval), ref[0][0])
    assert np.array_equal(bufs[1].to_array(np.float64, nval), ref[1][0])
    assert np.array_equal(bufs[2].to_array(np.float64, nval), want10)
    # rt_render ignores the flag (it returns the finished image)
    img = cam.render_image(world, seed=1, pipelined=True, **kw)
    assert np.array_equal(img.reshape(-1) * 16.0, ref[0][0]) or np.allclose(img.reshape(-1) * 16.0, ref[0][0], rtol=1e-15, atol=0)
    # a scene destroyed with a frame still queued
    w2 = World.build(desc)
    b2 = DeviceBuffer(nval * 8)
    b2.fill(0)
    o = cam.opts(seed=4, pipelined=True, **kw)
    check(L.rt_render_accumulate_device(w2._h, _abi.ptr(cam.rec), _abi.ptr(o), b2.ptr, stream, None))
    w2.close()
    synchronize()
    assert h.hipStreamDestroy(stream) == 0
