"""Progressive-pass contract of the preview loop (src/window.rs:224-325) — SURVEY §8 row f2."""
import numpy as np
import pytest

from rt_amd import preview, scenes


def test_schedule_matches_reference_table():
    assert len(preview.SCHEDULE) == 237 and sum(preview.SCHEDULE) == 40055
    assert preview.SCHEDULE[:9] == [1, 2, 4, 8, 8, 16, 16, 32, 32] and preview.SCHEDULE[9] == 64


def test_blend_is_running_mean_and_quantises_like_the_reference():
    calls = []

    def render(n, seed):
        calls.append((n, seed))
        return np.full((2, 3, 3), 0.1 * len(calls))

    f, st = preview.progressive(render, 3, 2, 4, seed=5, quantise=False)
    assert calls == [(1, 5), (2, 6), (4, 7), (8, 8)]
    w = np.array([1, 2, 4, 8.0])
    assert np.allclose(f, (w * np.array([0.1, 0.2, 0.3, 0.4])).sum() / w.sum())
    assert [s["total_samples"] for s in st] == [1, 3, 7, 15]
    calls.clear()
    q, _ = preview.progressive(render, 3, 2, 2, quantise=True)
    step1 = np.rint(0.1 * 255) / 255.0
    assert q.dtype == np.uint8 and (q == np.rint((0.2 * (2 / 3) + step1 * (1 / 3)) * 255)).all()
    assert (preview.as_rgb_linear(np.array([-0.5, 0.5, 7.0])) == [0, 128, 255]).all()


def test_progressive_with_oracle_renderer(oracle):
    desc = scenes.mixed_test_scene()
    cam = scenes.cam1(24, 18, max_depth=8)
    orc = oracle.OracleScene(desc)
    f, st = preview.progressive(lambda n, s: orc.render(cam, n, seed=s, nthreads=4), 24, 18, 5, seed=3, quantise=False)
    ref = sum(n * orc.render(cam, n, seed=3 + k, nthreads=4) for k, n in enumerate([1, 2, 4, 8, 8])) / 23.0
    np.testing.assert_allclose(f, ref, rtol=1e-12, atol=1e-15)


@pytest.mark.gpu
def test_render_with_preview_on_gpu(oracle):
    from rt_amd import World
    desc = scenes.mixed_test_scene()
    cam = scenes.cam1(64, 48, max_depth=8)
    world = World.build(desc)
    f, st = preview.render_with_preview(cam, world, n_passes=5, seed=3, quantise=False)
    orc = oracle.OracleScene(desc)
    ref, _ = preview.progressive(lambda n, s: orc.render(cam, n, seed=s, nthreads=8), 64, 48, 5, seed=3, quantise=False)
    np.testing.assert_allclose(f, ref, rtol=2e-6, atol=1e-12)
    q, _ = preview.render_with_preview(cam, world, n_passes=3, seed=3, quantise=True)
    assert q.dtype == np.uint8 and q.shape == (48, 64, 3)
