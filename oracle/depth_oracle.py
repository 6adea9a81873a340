"""CPU restatement of the live-camera depth hole filling (TEST INFRASTRUCTURE ONLY):
``fill_depth`` (Utils.py:455-514) as ``predict_ros.py:38-41`` applies it to each depth frame.

The reference body is a chain of OpenCV calls; OpenCV is not available offline, so each call is restated from its
published behaviour (all float32 images):

  cv2.dilate / cv2.erode   flat structuring element anchored at its centre; BORDER_CONSTANT with
                           morphologyDefaultBorderValue(): pixels outside the image never win (-inf / +inf)
  cv2.morphologyEx(CLOSE)  dilate, then erode, same element
  cv2.medianBlur(5)        exact median of the 5x5 window, BORDER_REPLICATE (the float32 path is a sorting network)
  cv2.bilateralFilter      bilateralFilter_32f: BORDER_REFLECT_101 padding, taps with sqrt(i^2+j^2) <= radius = d/2,
                           colour weights from a (4096+2)-entry table of exp(-x^2/(2 sigma_c^2)) spanning the image's
                           [min, max] range, linearly interpolated; centre tap weight 1; `src.copyTo(dst)` when
                           max - min < FLT_EPSILON
  cv2.GaussianBlur(5, 0)   sigma <= 0 and ksize 5: the fixed kernel [1 4 6 4 1]/16, separable, BORDER_REFLECT_101

PARITY UNPINNED for these OpenCV conventions (no reference test or golden pins fill_depth either).  What IS pinned
(tests/test_fill_depth.py): the morphology and the median against scipy.ndimage's independent implementations,
bit for bit, and the HIP kernels against this file (bit-exact up to the median, float32-roundoff after the blur).
"""
import math

import numpy as np

DIAMOND5 = np.array([[0, 0, 1, 0, 0], [0, 1, 1, 1, 0], [1, 1, 1, 1, 1], [0, 1, 1, 1, 0], [0, 0, 1, 0, 0]], np.uint8)


def _shifted(img, dy, dx, fill):
    """img shifted so that out[y, x] = img[y + dy, x + dx], `fill` outside."""
    H, W = img.shape
    out = np.full_like(img, fill)
    ys, ye = max(0, -dy), min(H, H - dy)
    xs, xe = max(0, -dx), min(W, W - dx)
    if ys < ye and xs < xe:
        out[ys:ye, xs:xe] = img[ys + dy:ye + dy, xs + dx:xe + dx]
    return out


def dilate(img, kernel):
    """cv2.dilate(img, kernel) on a float32 image (anchor = centre, border ignored)."""
    kh, kw = kernel.shape
    out = np.full_like(img, -np.inf)
    for i in range(kh):
        for j in range(kw):
            if kernel[i, j]:
                out = np.maximum(out, _shifted(img, i - kh // 2, j - kw // 2, -np.inf))
    return out


def erode(img, kernel):
    kh, kw = kernel.shape
    out = np.full_like(img, np.inf)
    for i in range(kh):
        for j in range(kw):
            if kernel[i, j]:
                out = np.minimum(out, _shifted(img, i - kh // 2, j - kw // 2, np.inf))
    return out


def median5(img):
    """cv2.medianBlur(img, 5) for float32: exact median, BORDER_REPLICATE."""
    p = np.pad(img, 2, mode="edge")
    H, W = img.shape
    stack = np.stack([p[i:i + H, j:j + W] for i in range(5) for j in range(5)], 0)
    return np.sort(stack, axis=0)[12]


def bilateral5(img, sigma_color=1.5, sigma_space=2.0, d=5):
    """cv2.bilateralFilter(img, 5, sigma_color, sigma_space) restated (bilateralFilter_32f, scalar loop order)."""
    img = np.asarray(img, np.float32)
    mn, mx = float(img.min()), float(img.max())
    if abs(mn - mx) < np.finfo(np.float32).eps:
        return img.copy()
    gcc = -0.5 / (sigma_color * sigma_color)
    gsc = -0.5 / (sigma_space * sigma_space)
    radius = max(d // 2, 1)
    bins = 1 << 12
    length = np.float32(mx - mn)
    scale_index = np.float32(bins) / length
    i = np.arange(bins + 2, dtype=np.float64)
    val = i / np.float64(scale_index)
    lut = np.exp(val * val * gcc).astype(np.float32)
    p = np.pad(img, radius, mode="reflect")       # numpy 'reflect' == BORDER_REFLECT_101
    H, W = img.shape
    wsum = np.ones((H, W), np.float32)
    acc = img.copy()
    for di in range(-radius, radius + 1):
        for dj in range(-radius, radius + 1):
            r = math.sqrt(float(di) * di + float(dj) * dj)
            if r > radius or (di == 0 and dj == 0):
                continue
            sw = np.float32(math.exp(r * r * gsc))
            v = p[radius + di:radius + di + H, radius + dj:radius + dj + W]
            alpha = np.abs(v - img) * scale_index
            idx = np.floor(alpha).astype(np.int64)
            alpha = alpha - idx.astype(np.float32)
            w = sw * (lut[idx] + alpha * (lut[idx + 1] - lut[idx]))
            acc = acc + v * w
            wsum = wsum + w
    return (acc / wsum).astype(np.float32)


def gaussian5(img):
    """cv2.GaussianBlur(img, (5,5), 0) for float32: [1 4 6 4 1]/16 rows then columns, BORDER_REFLECT_101."""
    k = np.array([0.0625, 0.25, 0.375, 0.25, 0.0625], np.float32)
    H, W = img.shape
    p = np.pad(img, ((0, 0), (2, 2)), mode="reflect")
    tmp = np.zeros_like(img)
    for t in range(5):
        tmp = tmp + p[:, t:t + W] * k[t]
    p = np.pad(tmp, ((2, 2), (0, 0)), mode="reflect")
    out = np.zeros_like(img)
    for t in range(5):
        out = out + p[t:t + H, :] * k[t]
    return out


def fill_depth(depth, max_depth=2.0, extrapolate=False, blur_type="bilateral", stages=None):
    """Utils.py:455-514, statement by statement.  depth in metres (any float dtype) -> float32 metres.
    `stages`: optional dict that receives the intermediate images (tests)."""
    depth = depth.astype(np.float32)
    valid = depth > 0.1
    depth[valid] = np.float32(max_depth) - depth[valid]
    depth = dilate(depth, DIAMOND5)
    depth = erode(dilate(depth, np.ones((5, 5), np.uint8)), np.ones((5, 5), np.uint8))    # MORPH_CLOSE
    empty = depth < 0.1
    dil = dilate(depth, np.ones((7, 7), np.uint8))
    depth[empty] = dil[empty]
    if extrapolate:
        top = np.argmax(depth > 0.1, axis=0)
        vals = depth[top, range(depth.shape[1])]
        for c in range(depth.shape[1]):
            depth[0:top[c], c] = vals[c]
        empty = depth < 0.1
        dil = dilate(depth, np.ones((31, 31), np.uint8))
        depth[empty] = dil[empty]
    if stages is not None:
        stages["filled"] = depth.copy()
    depth = median5(depth)
    if stages is not None:
        stages["median"] = depth.copy()
    if blur_type == "bilateral":
        depth = bilateral5(depth, 1.5, 2.0, 5)
    elif blur_type == "gaussian":
        valid = depth > 0.1
        blurred = gaussian5(depth)
        depth[valid] = blurred[valid]
    valid = depth > 0.1
    depth[valid] = np.float32(max_depth) - depth[valid]
    return depth


def grab_depth(depth_mm, max_depth=2.0, extrapolate=False, blur_type="bilateral"):
    """predict_ros.py:38-41: uint16 mm frame -> filled uint16 mm frame."""
    d = fill_depth(np.asarray(depth_mm).astype(np.uint16) / 1e3, max_depth, extrapolate, blur_type)
    # regions beyond max_depth are still negative here (only their rim gets filled); the reference's astype wraps them
    # modulo 2^16 on x86-64 (-500 -> 65036): stated explicitly so that the result does not depend on NumPy's cast path
    return ((d * 1000).astype(np.int64) & 0xFFFF).astype(np.uint16)
