"""oracle/cv2_resize.py (CPU restatement of cv2.resize for uint8): structural properties and independent float checks.

OpenCV is absent from the image (parity unpinned, see the oracle's header); these tests pin what can be pinned
without it: the documented special cases of cv2.resize and agreement, within the fixed-point rounding, with
independent float64 formulations of each filter.
"""
import numpy as np
import pytest

from oracle import cv2_resize as R


def _rng_img(h, w, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def test_same_size_is_a_copy_and_constant_images_stay_constant():
    a = _rng_img(33, 47)
    for interp in (R.INTER_LINEAR, R.INTER_CUBIC, R.INTER_AREA):
        assert np.array_equal(R.resize(a, (47, 33), interp), a)
    c = np.full((37, 53, 3), 137, np.uint8)
    for interp in (R.INTER_LINEAR, R.INTER_CUBIC, R.INTER_AREA):
        for d in [(20, 11), (106, 74), (53, 74), (64, 64), (26, 37)]:
            assert (R.resize(c, d, interp) == 137).all(), (interp, d)


def test_linear_at_exactly_half_size_is_the_2x2_area_average():
    """cv::resize re-routes INTER_LINEAR with scale (2, 2) to INTER_AREA: (a+b+c+d+2) >> 2 -- the 40x -> 20x tile."""
    a = _rng_img(512, 512)
    want = ((a.reshape(256, 2, 256, 2, 3).astype(np.int32).sum((1, 3)) + 2) >> 2).astype(np.uint8)
    assert np.array_equal(R.resize(a, (256, 256)), want)
    assert np.array_equal(R.resize(a, (256, 256), R.INTER_AREA), want)


def test_area_integer_ratio_rounds_half_to_even_except_2x2():
    a = np.zeros((4, 4, 3), np.uint8)
    a[0, 0] = 2                          # sum 2 over a 2x2 cell: (2+2)>>2 = 1 ; over a 4x4 cell: 2/16 -> 0
    assert R.resize(a, (2, 2), R.INTER_AREA)[0, 0, 0] == 1
    b = np.zeros((4, 4, 3), np.uint8)
    b[0, :2] = 4                         # 4x4 cell sum 8 -> 0.5 -> half-to-even -> 0
    assert R.resize(b, (1, 1), R.INTER_AREA)[0, 0, 0] == 0
    b[1, 0] = 16                         # sum 24 -> 1.5 -> 2
    assert R.resize(b, (1, 1), R.INTER_AREA)[0, 0, 0] == 2
    img = _rng_img(96, 64, 3)
    got = R.resize(img, (16, 32), R.INTER_AREA)       # 4 x 3 cells
    mean = img.reshape(32, 3, 16, 4, 3).astype(np.float64).mean((1, 3))
    assert np.abs(got.astype(np.float64) - mean).max() <= 0.5 + 1e-4


def test_area_enlarging_by_an_integer_factor_is_pixel_replication():
    a = _rng_img(10, 12)
    assert np.array_equal(R.resize(a, (24, 20), R.INTER_AREA), a.repeat(2, 0).repeat(2, 1))
    assert np.array_equal(R.resize(a, (36, 30), R.INTER_AREA), a.repeat(3, 0).repeat(3, 1))


def _bilinear_f64(a, dw, dh):
    h, w, _ = a.shape
    xs = np.clip((np.arange(dw) + 0.5) * (w / dw) - 0.5, 0, w - 1)
    ys = np.clip((np.arange(dh) + 0.5) * (h / dh) - 0.5, 0, h - 1)
    x0, y0 = np.floor(xs).astype(int), np.floor(ys).astype(int)
    x1, y1 = np.minimum(x0 + 1, w - 1), np.minimum(y0 + 1, h - 1)
    fx, fy = (xs - x0)[None, :, None], (ys - y0)[:, None, None]
    A = a.astype(np.float64)
    top = A[y0][:, x0] * (1 - fx) + A[y0][:, x1] * fx
    bot = A[y1][:, x0] * (1 - fx) + A[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


@pytest.mark.parametrize("shape,dsize", [((300, 300), (256, 256)), ((1024, 1024), (256, 256)), ((180, 200), (256, 256)),
                                          ((511, 513), (256, 256)), ((97, 131), (40, 300))])
def test_linear_is_non_antialiased_half_pixel_bilinear_within_fixed_point_rounding(shape, dsize):
    a = _rng_img(*shape, seed=5)
    got = R.resize(a, dsize, R.INTER_LINEAR).astype(np.float64)
    want = _bilinear_f64(a, *dsize)
    assert np.abs(got - want).max() <= 1.0, np.abs(got - want).max()          # 11-bit weights + two-stage shift
    assert abs((got - want).mean()) < 0.3                                      # the two floors of the vertical pass bias low


def _cubic_w(t, A=-0.75):
    t = np.abs(t)
    return np.where(t <= 1, ((A + 2) * t - (A + 3)) * t * t + 1, np.where(t < 2, ((A * t - 5 * A) * t + 8 * A) * t - 4 * A, 0.0))


def _bicubic_f64(a, dw, dh):
    h, w, _ = a.shape
    A = a.astype(np.float64)

    def axis(n_out, n_in):
        c = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        s = np.floor(c).astype(int)
        idx = np.stack([np.clip(s - 1 + j, 0, n_in - 1) for j in range(4)], 1)
        wts = np.stack([_cubic_w(c - (s - 1 + j)) for j in range(4)], 1)
        return idx, wts

    xi, xw = axis(dw, w)
    yi, yw = axis(dh, h)
    H = sum(A[:, xi[:, j]] * xw[:, j][None, :, None] for j in range(4))
    return sum(H[yi[:, k]] * yw[:, k][:, None, None] for k in range(4))


@pytest.mark.parametrize("mode", ["sse", "scalar"])
def test_cubic_uses_the_minus_three_quarters_kernel(mode):
    a = _rng_img(80, 100, seed=9)
    got = R.resize(a, (256, 200), R.INTER_CUBIC, cubic_vertical=mode).astype(np.float64)
    want = np.clip(_bicubic_f64(a, 256, 200), 0, 255)
    assert np.abs(got - want).max() <= 1.5


def test_cubic_vector_and_scalar_vertical_pass_differ_by_at_most_one_level():
    a = _rng_img(60, 75, seed=2)
    v = R.resize(a, (201, 163), R.INTER_CUBIC, cubic_vertical="sse").astype(int)
    s = R.resize(a, (201, 163), R.INTER_CUBIC, cubic_vertical="scalar").astype(int)
    assert np.abs(v - s).max() <= 1 and (v != s).mean() < 1e-3
    assert np.array_equal(v[:, -1], s[:, -1])              # 201 * 3 % 8 = 3 trailing elements take the scalar path


def _area_f64(a, dw, dh):
    """Exact box integration of the piecewise-constant image over each destination cell."""
    h, w, _ = a.shape

    def axis(n_out, n_in):
        sc = n_in / n_out
        M = np.zeros((n_out, n_in))
        for d in range(n_out):
            lo, hi = d * sc, min((d + 1) * sc, n_in)
            for s in range(int(np.floor(lo)), int(np.ceil(hi))):
                M[d, s] = max(0.0, min(hi, s + 1) - max(lo, s))
            M[d] /= M[d].sum()
        return M

    X, Y = axis(dw, w), axis(dh, h)
    rows = np.tensordot(Y, a.astype(np.float64), axes=(1, 0))            # [dh, w, c]
    return np.tensordot(rows, X, axes=(1, 1)).transpose(0, 2, 1)         # [dh, dw, c]


@pytest.mark.parametrize("shape,dsize", [((411, 300), (128, 100)), ((100, 100), (33, 77)), ((733, 1024), (500, 699))])
def test_general_area_is_box_integration_within_float32_rounding(shape, dsize):
    a = _rng_img(*shape, seed=4)
    got = R.resize(a, dsize, R.INTER_AREA).astype(np.float64)
    want = _area_f64(a, *dsize)
    assert np.abs(got - want).max() <= 0.5 + 2e-3


@pytest.mark.parametrize("interp,mode", [(R.INTER_LINEAR, "bilinear"), (R.INTER_CUBIC, "bicubic")])
@pytest.mark.parametrize("src_hw,dst_hw", [((256, 256), (224, 224)), ((97, 131), (224, 302)), ((300, 256), (131, 97)),
                                           ((64, 48), (200, 333)), ((512, 384), (300, 300))])
def test_against_torch_interpolate_an_independent_implementation_of_the_same_convention(interp, mode, src_hw, dst_hw):
    """torch.nn.functional.interpolate(align_corners=False, antialias=False) samples at the same half-pixel-centre positions
    as cv2.resize and its bicubic uses the same A = -0.75 kernel -- independent code (ATen), float arithmetic, borders by index
    clamping.  The restatement must agree with it up to the 8-bit paths' own rounding: OpenCV quantises the coefficients to
    11 bits (2^-11 per tap) and rounds twice (fixed-point intermediate, final cast); every pixel within 1 LSB (2 for bicubic,
    where four taps per axis accumulate the coefficient quantisation), and the mean absolute deviation near the rounding's
    own 0.25 LSB.  (Exact-halving and identity shapes are routed to other code paths by cv2.resize and are tested above.)"""
    import torch
    a = _rng_img(*src_hw, seed=5)
    got = R.resize(a, (dst_hw[1], dst_hw[0]), interp).astype(np.float64)
    x = torch.from_numpy(a).permute(2, 0, 1)[None].double()
    ref = torch.nn.functional.interpolate(x, size=dst_hw, mode=mode, align_corners=False, antialias=False)[0].permute(1, 2, 0)
    ref = ref.clamp(0, 255).numpy()
    d = np.abs(got - ref)
    assert d.max() <= (1.0 if interp == R.INTER_LINEAR else 2.0) + 1e-6, d.max()
    assert d.mean() <= 0.32, d.mean()
