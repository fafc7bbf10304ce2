"""The event-count estimate of FeatureDetector::updateNumOfEvents (feature_detector.cpp:689-707):
cv::warpAffine with flags = WARP_INVERSE_MAP (nearest neighbour through OpenCV's 10-bit fixed-point
map) of the gradient images, L1 norm of 0.6 gradX cos + 0.6 gradY sin over the patch rect, truncated
to size_t.  OpenCV is absent: the oracle restates its published algorithm (PARITY UNPINNED) and is
checked here on the warps for which any nearest-neighbour map is exact; the device against the oracle."""
import numpy as np
import pytest


def scene(w=240, h=180, seed=3):
    rng = np.random.default_rng(seed)
    return rng.normal(0, 1.5, (h, w, 2))


def direct(grad, rect, shift, flow):
    """identity rotation, integer translation: warped(y, x) = grad(y + ty, x + tx), no rounding anywhere"""
    x0, y0, w, h = (int(round(v)) for v in rect)
    a = 0.6 * float(np.cos(np.float32(flow)))
    b = 0.6 * float(np.sin(np.float32(flow)))
    H, W = grad.shape[:2]
    s = 0.0
    for y in range(y0, y0 + h):
        for x in range(x0, x0 + w):
            X, Y = x + shift[0], y + shift[1]
            if 0 <= X < W and 0 <= Y < H:
                s += abs(grad[Y, X, 0] * a + grad[Y, X, 1] * b)
    return int(s)


@pytest.mark.parametrize("shift", [(0, 0), (3, -2), (-7, 11), (230, 0)])
def test_oracle_is_exact_for_identity_and_integer_translations(orc, shift):
    grad = scene()
    for rect, flow in (((30.0, 40.0, 25.0, 25.0), 0.3), ((100.5, 77.5, 25.0, 25.0), 2.1), ((7.0, 6.0, 25.0, 25.0), -1.0)):
        pose = np.array([1.0, 0.0, float(shift[0]), float(shift[1])])
        assert orc.estimate_num_events(grad, rect, pose, flow) == direct(grad, rect, shift, flow)


def test_oracle_fixed_point_map_properties(orc):
    """A rotation about a pixel by a small angle moves no source index inside half a pixel of the
    pivot; a half-pixel translation rounds up (X0 carries +512 before the shift by 10 bits)."""
    grad = scene()
    rect = (100.0, 80.0, 25.0, 25.0)
    # +0.5 px in x: floor(x + 0.5 + 0.5) = x + 1
    assert orc.estimate_num_events(grad, rect, np.array([1.0, 0.0, 0.5, 0.0]), 0.7) == direct(grad, rect, (1, 0), 0.7)
    # +0.499 px: stays
    assert orc.estimate_num_events(grad, rect, np.array([1.0, 0.0, 0.499, 0.0]), 0.7) == direct(grad, rect, (0, 0), 0.7)
    # the 10-bit grid: 0.4995 * 1024 = 511.49 -> 511, + 512 = 1023 -> still x; 0.49952 * 1024 = 511.51 -> 512 -> x + 1
    assert orc.estimate_num_events(grad, rect, np.array([1.0, 0.0, 0.4995, 0.0]), 0.7) == direct(grad, rect, (0, 0), 0.7)
    assert orc.estimate_num_events(grad, rect, np.array([1.0, 0.0, 0.49952, 0.0]), 0.7) == direct(grad, rect, (1, 0), 0.7)


@pytest.mark.gpu
def test_device_estimate_matches_the_oracle(ebo, orc):
    grad = scene(seed=9)
    rng = np.random.default_rng(4)
    n = 300
    rects = np.stack([rng.uniform(1, 210, n), rng.uniform(1, 150, n), np.full(n, 25.0), np.full(n, 25.0)], 1)
    rects[::7, :2] = np.floor(rects[::7, :2]) + 0.5  # exact .5 corners: round half to even
    th = rng.uniform(-0.4, 0.4, n)
    th[::5] = 0.0
    poses = np.stack([np.cos(th), np.sin(th), rng.uniform(-30, 30, n), rng.uniform(-30, 30, n)], 1)
    poses[::11, 2:] = np.round(poses[::11, 2:]) + 0.5
    flows = rng.uniform(-7, 7, n)
    with ebo.Context(image_w=240, image_h=180) as c:
        with pytest.raises(ebo.EboError) as ei:
            c.estimate_num_events(rects, poses, flows)
        assert ei.value.code == ebo.ERR_STATE
        c.optimizer_set_grad(grad[:, :, 0], grad[:, :, 1])
        got = c.estimate_num_events(rects, poses, flows)
        assert len(c.estimate_num_events(rects[:0], poses[:0], flows[:0])) == 0
    want = np.array([orc.estimate_num_events(grad, rects[i], poses[i], flows[i]) for i in range(n)], dtype=np.uint64)
    assert np.array_equal(got, want)
    assert want.max() > 300 and want.min() < want.max()
