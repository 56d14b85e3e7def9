"""CPU restatement of ``cv2.resize`` for 8-bit images (1..4 channels).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: opencv-python
(``opencv-python>=4.7.0``, /root/reference/pyproject.toml:35) is not installed in this
image and the reference holds no test vectors for it, so this file restates the published
algorithm of OpenCV 4.x ``modules/imgproc/src/resize.cpp`` (``cv::hal::resize`` and the
invokers it dispatches to) and is anchored on the reference's call sites:

  * ``cv2.resize(patch, (ps, ps))``  -> INTER_LINEAR
        atlas_patch/services/feature_embedding.py:95, atlas_patch/services/extraction.py:113
        (every tile of a slide whose level read is not ``patch_size``, e.g. a 40x slide at
        ``--target-mag 20`` with levels 1/4/16: 512 x 512 -> 256 x 256)
  * ``cv2.resize(arr, (out_w, out_h), interpolation=INTER_AREA | INTER_CUBIC | INTER_LINEAR)``
        atlas_patch/core/wsi/iwsi.py:305-321 (the 1.25x thumbnail)

What OpenCV does for ``depth == CV_8U`` (x86-64 wheels: IPP is skipped for 8-bit linear unless
"not exact" IPP is enabled, there is no HAL override, baseline SSE2/SSE3 universal intrinsics):

  scale_x = 1 / (dsize.width / ssize.width) in double (likewise y).
  INTER_LINEAR with scale exactly (2, 2) is re-routed to INTER_AREA.
  INTER_AREA, scale_x >= 1 and scale_y >= 1
      integer scales ("area fast"):  2 x 2 -> (a + b + c + d + 2) >> 2;
                                     else   -> cvRound(float(sum) * (1.f / area))   (float32, half-even)
      otherwise: ``computeResizeAreaTab`` (double arithmetic, float32 weights) per axis, then per
                 source row ``buf[dx] += S[sx] * alpha`` in table order, ``sum[dx] (+)= beta * buf[dx]``
                 in float32, cvRound + saturate at the end of each destination row.
  INTER_AREA with any scale < 1 = the bilinear code below with "area mode" coordinates.
  INTER_LINEAR: fx = float((dx + 0.5) * scale_x - 0.5), sx = floor(fx), fx -= sx; left / right clamps
      (sx < 0 -> sx = 0, fx = 0; sx >= w - 1 -> sx = w - 1, fx = 0);  weights
      short(cvRound(w * 2048)) for (1.f - fx, fx), each rounded on its own; rows are NOT clamped in
      the table -- the row pointers are (clip(sy, 0, h-1), clip(sy + 1, 0, h-1));
      horizontal: t = S[sx] * a0 + S[sx + 1] * a1 (int32);
      vertical:   ((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2.
  INTER_CUBIC: ``interpolateCubic`` (A = -0.75f, float32), short weights as above, taps
      sx - 1 .. sx + 2 / sy - 1 .. sy + 2 clamped to the image (replicate); horizontal int32 sums;
      vertical: the scalar code is ``saturate_u8((sum_k S_k * b_k + 2^21) >> 22)`` in int32, but the
      vector loop that serves all but the last ``(width * cn) % 8`` elements of a row converts to
      float32: ``round_half_even(S0*b0' + (S1*b1' + (S2*b2' + S3*b3')))`` with ``b' = b * 2^-22``,
      separate multiply and add (no FMA at the SSE baseline).  ``cubic_vertical="sse"`` (default)
      follows the vector loop + scalar tail, ``"scalar"`` the integer formula everywhere.  The two
      differ only on near-ties of the final rounding.

Every function here is written from the algorithm's description; no OpenCV source is present in
this repository or image.
"""
from __future__ import annotations

import math

import numpy as np

INTER_NEAREST = 0
INTER_LINEAR = 1
INTER_CUBIC = 2
INTER_AREA = 3

_COEF_BITS = 11
_COEF_SCALE = 1 << _COEF_BITS
_F32 = np.float32
_DBL_EPS = 2.220446049250313e-16


def _sat_short_round(v: np.float32) -> int:
    """saturate_cast<short>(float): cvRound (half to even) then clamp."""
    r = int(np.rint(_F32(v)))
    return max(-32768, min(32767, r))


def _cubic_coeffs(x: np.float32):
    """interpolateCubic (float32, A = -0.75f), operation order kept."""
    A = _F32(-0.75)
    x = _F32(x)
    one = _F32(1)
    xp1 = _F32(x + one)
    c0 = _F32(_F32(_F32(_F32(_F32(_F32(A * xp1) - _F32(5) * A) * xp1) + _F32(8) * A) * xp1) - _F32(4) * A)
    c1 = _F32(_F32(_F32(_F32(_F32(_F32(A + _F32(2)) * x) - _F32(A + _F32(3))) * x) * x) + one)
    omx = _F32(one - x)
    c2 = _F32(_F32(_F32(_F32(_F32(_F32(A + _F32(2)) * omx) - _F32(A + _F32(3))) * omx) * omx) + one)
    c3 = _F32(_F32(_F32(one - c0) - c1) - c2)
    return c0, c1, c2, c3


def _weights(f: np.float32, interp: int):
    if interp == INTER_CUBIC:
        c = _cubic_coeffs(f)
    else:
        c = (_F32(_F32(1) - _F32(f)), _F32(f))
    return [_sat_short_round(_F32(v * _F32(_COEF_SCALE))) for v in c]


def linear_tables(ssize_wh, dsize_wh, interp: int):
    """(xofs, ialpha [dw, k], yofs, ibeta [dh, k]) exactly as cv::hal::resize builds them for 8-bit input."""
    sw, sh = int(ssize_wh[0]), int(ssize_wh[1])
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    inv_x, inv_y = dw / sw, dh / sh
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    area_mode = interp == INTER_AREA
    cubic = interp == INTER_CUBIC
    ksize = 4 if cubic else 2
    xofs = np.zeros(dw, np.int64)
    ialpha = np.zeros((dw, ksize), np.int32)
    for dx in range(dw):
        if not area_mode:
            fx = _F32((dx + 0.5) * scale_x - 0.5)
            sx = int(math.floor(float(fx)))
            fx = _F32(fx - _F32(sx))
        else:
            sx = int(math.floor(dx * scale_x))
            fx = _F32((dx + 1) - (sx + 1) * inv_x)
            fx = _F32(0) if fx <= 0 else _F32(fx - _F32(math.floor(float(fx))))
        if sx < ksize // 2 - 1 and sx < 0 and not cubic:
            fx, sx = _F32(0), 0
        if sx + ksize // 2 >= sw and sx >= sw - 1 and not cubic:
            fx, sx = _F32(0), sw - 1
        xofs[dx] = sx
        ialpha[dx] = _weights(fx, interp)
    yofs = np.zeros(dh, np.int64)
    ibeta = np.zeros((dh, ksize), np.int32)
    for dy in range(dh):
        if not area_mode:
            fy = _F32((dy + 0.5) * scale_y - 0.5)
            sy = int(math.floor(float(fy)))
            fy = _F32(fy - _F32(sy))
        else:
            sy = int(math.floor(dy * scale_y))
            fy = _F32((dy + 1) - (sy + 1) * inv_y)
            fy = _F32(0) if fy <= 0 else _F32(fy - _F32(math.floor(float(fy))))
        yofs[dy] = sy
        ibeta[dy] = _weights(fy, interp)
    return xofs, ialpha, yofs, ibeta


def area_table(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab for one axis: list of (di, si, alpha float32) in table order."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, _F32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, _F32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, _F32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _round_sat_u8(v: np.ndarray) -> np.ndarray:
    """saturate_cast<uchar>(float): cvRound (half to even), clamp to [0, 255]."""
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def _resize_area_fast(src: np.ndarray, dw: int, dh: int, isx: int, isy: int) -> np.ndarray:
    h, w, cn = src.shape
    blocks = src[:dh * isy, :dw * isx].reshape(dh, isy, dw, isx, cn).astype(np.int32)
    s = blocks.sum(axis=(1, 3))
    if isx == 2 and isy == 2 and cn in (1, 3, 4):
        return ((s + 2) >> 2).astype(np.uint8)
    scale = _F32(_F32(1) / _F32(isx * isy))
    return _round_sat_u8(s.astype(np.float32) * scale)


def _resize_area_general(src: np.ndarray, dw: int, dh: int, scale_x: float, scale_y: float) -> np.ndarray:
    h, w, cn = src.shape
    xtab = area_table(w, dw, scale_x)
    ytab = area_table(h, dh, scale_y)
    # group the x table per destination column, preserving table order
    per_dx = [[] for _ in range(dw)]
    for di, si, a in xtab:
        per_dx[di].append((si, a))
    maxk = max(len(v) for v in per_dx)
    xsi = np.zeros((dw, maxk), np.int64)
    xal = np.zeros((dw, maxk), np.float32)
    xn = np.array([len(v) for v in per_dx])
    for d, v in enumerate(per_dx):
        for k, (si, a) in enumerate(v):
            xsi[d, k], xal[d, k] = si, a
    out = np.zeros((dh, dw, cn), np.uint8)
    S = src.astype(np.float32)

    def hrow(sy: int) -> np.ndarray:
        buf = np.zeros((dw, cn), np.float32)
        for k in range(maxk):
            live = (k < xn)[:, None]
            term = (S[sy][xsi[:, k]] * xal[:, k][:, None]).astype(np.float32)
            buf = np.where(live, (buf + term).astype(np.float32), buf)
        return buf

    # the table is ordered by destination row, source rows ascending; the row a destination cell shares with
    # its neighbour sits at the end of one group and the start of the next -> a one-row cache is enough
    cached_si, cached = -1, None
    sums: dict[int, np.ndarray] = {}
    for di, si, beta in ytab:
        if si != cached_si:
            cached_si, cached = si, hrow(si)
        t = (beta * cached).astype(np.float32)
        sums[di] = t if di not in sums else (sums[di] + t).astype(np.float32)
    for di, v in sums.items():
        out[di] = _round_sat_u8(v)
    return out


def _resize_generic(src: np.ndarray, dw: int, dh: int, interp: int, cubic_vertical: str) -> np.ndarray:
    h, w, cn = src.shape
    xofs, ialpha, yofs, ibeta = linear_tables((w, h), (dw, dh), interp)
    S = src.astype(np.int64)
    if interp == INTER_CUBIC:
        taps = [np.clip(xofs - 1 + j, 0, w - 1) for j in range(4)]
        H = sum(S[:, taps[j], :] * ialpha[:, j][None, :, None] for j in range(4))       # [h, dw, cn] int
        rows = [np.clip(yofs - 1 + k, 0, h - 1) for k in range(4)]
        R = [H[rows[k]] for k in range(4)]                                                 # [dh, dw, cn]
        b = [ibeta[:, k][:, None, None] for k in range(4)]
        acc = R[0] * b[0] + R[1] * b[1] + R[2] * b[2] + R[3] * b[3]
        acc32 = ((acc + (1 << 31)) % (1 << 32)) - (1 << 31)                                 # int32 wrap, as the C code
        scalar = np.clip((acc32 + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
        if cubic_vertical == "scalar":
            return scalar
        sc = _F32(1.0 / (_COEF_SCALE * _COEF_SCALE))
        bf = [(ibeta[:, k].astype(np.float32) * sc)[:, None, None] for k in range(4)]
        Rf = [r.astype(np.float32) for r in R]
        v = (Rf[3] * bf[3]).astype(np.float32)
        v = ((Rf[2] * bf[2]).astype(np.float32) + v).astype(np.float32)
        v = ((Rf[1] * bf[1]).astype(np.float32) + v).astype(np.float32)
        v = ((Rf[0] * bf[0]).astype(np.float32) + v).astype(np.float32)
        vec = _round_sat_u8(v)
        flat_vec, flat_sc = vec.reshape(dh, dw * cn), scalar.reshape(dh, dw * cn)
        nvec = (dw * cn // 8) * 8                   # v_int16 lanes at the 128-bit baseline
        out = flat_sc.copy()
        out[:, :nvec] = flat_vec[:, :nvec]
        return out.reshape(dh, dw, cn)
    x1 = np.minimum(xofs + 1, w - 1)
    H = S[:, xofs, :] * ialpha[:, 0][None, :, None] + S[:, x1, :] * ialpha[:, 1][None, :, None]
    r0, r1 = np.clip(yofs, 0, h - 1), np.clip(yofs + 1, 0, h - 1)
    b0, b1 = ibeta[:, 0][:, None, None], ibeta[:, 1][:, None, None]
    v = (((b0 * (H[r0] >> 4)) >> 16) + ((b1 * (H[r1] >> 4)) >> 16) + 2) >> 2
    return (v & 0xFF).astype(np.uint8)


def resize(src: np.ndarray, dsize, interpolation: int = INTER_LINEAR, *, cubic_vertical: str = "sse") -> np.ndarray:
    """``cv2.resize(src, dsize=(width, height), interpolation=...)`` for uint8 [h, w] / [h, w, cn]."""
    arr = np.asarray(src)
    assert arr.dtype == np.uint8 and arr.ndim in (2, 3)
    squeeze = arr.ndim == 2
    a = arr[:, :, None] if squeeze else arr
    h, w, cn = a.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    assert dw > 0 and dh > 0
    inv_x, inv_y = dw / w, dh / h
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    if (dw, dh) == (w, h):                  # cv::resize: "dsize == ssize -> src.copyTo(dst)"
        return arr.copy()
    isx, isy = int(np.rint(scale_x)), int(np.rint(scale_y))          # saturate_cast<int>(double) = cvRound
    area_fast = abs(scale_x - isx) < _DBL_EPS and abs(scale_y - isy) < _DBL_EPS
    interp = interpolation
    if interp == INTER_LINEAR and area_fast and isx == 2 and isy == 2:
        interp = INTER_AREA
    if interp == INTER_AREA and scale_x >= 1 and scale_y >= 1:
        out = _resize_area_fast(a, dw, dh, isx, isy) if area_fast else _resize_area_general(a, dw, dh, scale_x, scale_y)
    elif interp in (INTER_LINEAR, INTER_AREA, INTER_CUBIC):
        out = _resize_generic(a, dw, dh, interp, cubic_vertical)
    else:
        raise ValueError("unsupported interpolation")
    return out[:, :, 0] if squeeze else out
