"""Patch::warpImage (patch.cpp:132-154): predictedNabla_ = -gradX'(patch_) cos(flowDir_) - gradY'(patch_)
sin(flowDir_), the gradient images warped by cv::warpAffine(..., warp_.matrix2x3(), cv::WARP_INVERSE_MAP)
(nearest neighbour through OpenCV's fixed-point map).  The reference holds ONE test of it,
patch_test.cpp:62-108 (warpImageTest): two axis-aligned lines drawn by cv::line, a 45 degree warp, and
the bar `predictedNabla(i, j) <= 0` on both diagonals.  OpenCV is absent: the oracle restates the
published algorithm; cv::line of thickness 1 between two axis-aligned end points is the pixels between
them (restated in `line`).  The scenario through the oracle (CPU) and through the HIP path (GPU)."""
import numpy as np
import pytest


def line(img, p0, p1, value):
    """cv::line(img, p0, p1, value) for an axis-aligned segment, thickness 1, end points inclusive"""
    (x0, y0), (x1, y1) = p0, p1
    assert x0 == x1 or y0 == y1
    img[min(y0, y1):max(y0, y1) + 1, min(x0, x1):max(x0, x1) + 1] = value


def reference_scene(n):
    """patch_test.cpp:66-70 on an n x n image: gradX = a vertical line through the centre, gradY a horizontal one"""
    c = n // 2
    gx, gy = np.zeros((n, n)), np.zeros((n, n))
    line(gx, (c, 0), (c, n - 1), 1.0)
    line(gy, (0, c), (n - 1, c), 1.0)
    return np.stack([gx, gy], axis=-1)


def rot(angle):
    """Sophus::SE2d::rot(angle).data(): (cos, sin, 0, 0)"""
    return np.array([np.cos(angle), np.sin(angle), 0.0, 0.0])


def check_diagonals(image, n_inner):
    """patch_test.cpp:82-91: for i, j in 1..9: on either diagonal the predicted nabla is <= 0"""
    for i in range(1, n_inner):
        for j in range(1, n_inner):
            if i == j or i == n_inner - j:
                assert image[i, j] <= 0


def test_reference_warp_image_scenario_oracle(orc):
    """The reference's own set-up: an 11 x 11 image and Patch({5, 5}, extent 5) -- the rect (0, 0, 11, 11)
    touches the border (patch_.x + patch_.width >= cols), so warpImage returns before it assigns
    (:145-150) and predictedNabla_ keeps the zeros of Patch::init (patch.cpp:28): the bar holds."""
    grad = reference_scene(11)
    rect = (5.0 - 5, 5.0 - 5, 11.0, 11.0)  # patch.cpp:12
    out = orc.patch_warp_image(grad, rect, rot(np.pi / 4), np.float32(np.pi / 4))
    assert out is None
    check_diagonals(np.zeros((11, 11)), 10)


def test_warp_image_inside_the_border_oracle(orc):
    """The same scene two pixels larger, so that the rect clears the border and the warp runs: rotated by
    45 degrees about the origin and read back through the nearest-neighbour map, the two lines meet the
    patch as diagonals; with flowDir = 45 degrees every line pixel predicts -cos or -sin < 0, the rest 0."""
    grad = reference_scene(15)
    rect = (7.0 - 5, 7.0 - 5, 11.0, 11.0)
    flow = float(np.float32(np.pi / 4))
    out = orc.patch_warp_image(grad, rect, np.array([1.0, 0.0, 0.0, 0.0]), flow)
    assert out is not None and out.shape == (11, 11)
    # identity warp: the predicted nabla is -gx cos - gy sin of the rect itself
    want = -grad[2:13, 2:13, 0] * np.cos(flow) - grad[2:13, 2:13, 1] * np.sin(flow)
    np.testing.assert_allclose(out, want, rtol=0, atol=1e-16)
    assert (out <= 0).all() and (out < 0).sum() == 21  # the two lines cross the 11 x 11 rect, sharing one pixel
    # a rotation about the image centre c: x' = R (x - c) + c, i.e. the translation c - R c
    c = np.array([7.0, 7.0])
    ang = np.pi / 4
    R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    t = c - R @ c
    pose = np.array([np.cos(ang), np.sin(ang), t[0], t[1]])
    out = orc.patch_warp_image(grad, rect, pose, flow)
    assert (out <= 0).all()
    check_diagonals(out, 10)
    assert out[5, 5] < 0 and out[2, 2] < 0 and out[2, 8] < 0  # the rotated lines ARE the diagonals
    assert out[5, 1] == 0 and out[1, 5] == 0                  # ... and no longer the axes


@pytest.mark.gpu
def test_reference_warp_image_scenario_device(ebo, orc):
    """patch_test.cpp:62-108 through the HIP path (ebo_patch_warp_image): the early return of the reference's own
    set-up, and the scenario where the warp runs, equal to the oracle."""
    for n, corner in ((11, 5.0), (15, 7.0)):
        grad = reference_scene(n)
        rect = (corner - 5, corner - 5, 11.0, 11.0)
        pose = rot(np.pi / 4)
        flow = float(np.float32(np.pi / 4))
        with ebo.Context(image_w=n, image_h=n, patch_w=n, patch_h=n) as c:
            c.optimizer_set_grad(grad[:, :, 0].copy(), grad[:, :, 1].copy())
            got = c.patch_warp_image([rect], [pose], [flow])[0]
        want = orc.patch_warp_image(grad, rect, pose, flow)
        if n == 11:
            assert got is None and want is None
        else:
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-15)
            assert (got <= 0).all()
            check_diagonals(got, 10)


@pytest.mark.gpu
def test_device_warp_image_matches_the_oracle(ebo, orc):
    """300 random tracked patches in one launch: fractional rects (cv::Rect2d -> cv::Rect rounds half to even),
    rects on the border (skipped, as the reference returns early), warps that read outside the image
    (BORDER_CONSTANT 0), against the oracle pixel by pixel (the device's cos / sin may differ from glibc's in
    the last bit: 2 ulp of the largest gradient)."""
    rng = np.random.default_rng(5)
    W, H = 240, 180
    grad = rng.normal(0, 1.5, (H, W, 2))
    n = 300
    rects = np.stack([rng.uniform(-6, W - 18, n), rng.uniform(-6, H - 18, n), np.full(n, 25.0), np.full(n, 25.0)], 1)
    rects[::7, :2] = np.floor(rects[::7, :2]) + 0.5  # exact halves: round half to even
    rects[5] = (W - 25.0, 10.0, 25.0, 25.0)           # x + w == cols: border (>=)
    rects[6] = (W - 26.0, 10.0, 25.0, 25.0)           # one pixel inside
    ang = rng.uniform(-0.6, 0.6, n)
    poses = np.stack([np.cos(ang), np.sin(ang), rng.uniform(-30, 30, n), rng.uniform(-30, 30, n)], 1)
    flows = rng.uniform(-np.pi, np.pi, n)
    with ebo.Context(image_w=W, image_h=H) as c:
        c.optimizer_set_grad(grad[:, :, 0].copy(), grad[:, :, 1].copy())
        got = c.patch_warp_image(rects, poses, flows)
    skipped = 0
    for i in range(n):
        want = orc.patch_warp_image(grad, rects[i], poses[i], flows[i])
        if want is None:
            assert got[i] is None
            skipped += 1
        else:
            np.testing.assert_allclose(got[i], want, rtol=0, atol=4e-15)
    assert got[5] is None and got[6] is not None
    assert 5 < skipped < n // 2
