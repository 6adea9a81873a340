#!/usr/bin/env python3
"""Pin-on-first-contact script for the OpenCV rules the hot path depends on (TEST INFRASTRUCTURE ONLY).

OpenCV is not installed in the offline build container, so three pieces of the oracle restate OpenCV's published behaviour
and are marked "parity unpinned" (DESIGN.md section 4): the cv2.resize(INTER_NEAREST) source-index rule (Utils.py:343-344),
cv2.Rodrigues (datasets.py:173) and the fill_depth chain (Utils.py:455-514: cv2.dilate, morphologyEx(MORPH_CLOSE),
medianBlur, bilateralFilter, GaussianBlur).  Run THIS script once on any machine that has opencv-python (the reference's
docker image does: docker/dockerfile:25) -- it needs nothing from the repo but oracle/fixtures.py:

    python oracle/pin_opencv.py            # writes tests/golden/opencv_{resize,rodrigues,fill_depth}.npz
    python -m pytest tests/test_pinned_third_party.py        # oracle AND (with -m gpu) the HIP kernels vs the real cv2

and commit the three files: tests/test_pinned_third_party.py skips while they are absent and turns "unpinned" into pinned
the moment they exist.  The fill_depth statements below are the reference's own lines, in its order."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fixtures as Fx  # noqa: E402  (deliberately not `oracle.fixtures`: no package import, no torch-free guarantee needed)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
FILL_CASES = [(0, 120, 160, False, "bilateral"), (1, 120, 160, True, "bilateral"), (2, 480, 640, False, "bilateral"),
              (3, 240, 320, False, "gaussian"), (4, 240, 320, False, None), ("wall7", 240, 320, False, "bilateral")]


def fill_depth_stages(cv2, depth, max_depth=2.0, extrapolate=False, blur_type="bilateral"):
    """Utils.py:455-514, statement by statement, recording every intermediate image."""
    st = {}
    depth = depth.astype(np.float32)
    custom_kernel = np.array([[0, 0, 1, 0, 0], [0, 1, 1, 1, 0], [1, 1, 1, 1, 1], [0, 1, 1, 1, 0], [0, 0, 1, 0, 0]], dtype=np.uint8)
    valid_pixels = (depth > 0.1)
    depth[valid_pixels] = max_depth - depth[valid_pixels]
    depth = cv2.dilate(depth, custom_kernel); st["dilate_diamond"] = depth.copy()
    depth = cv2.morphologyEx(depth, cv2.MORPH_CLOSE, np.ones((5, 5), np.uint8)); st["close5"] = depth.copy()
    empty_pixels = (depth < 0.1)
    dilated = cv2.dilate(depth, np.ones((7, 7), np.uint8))
    depth[empty_pixels] = dilated[empty_pixels]; st["fill7"] = depth.copy()
    if extrapolate:
        top_row_pixels = np.argmax(depth > 0.1, axis=0)
        top_pixel_values = depth[top_row_pixels, range(depth.shape[1])]
        for c in range(depth.shape[1]):
            depth[0:top_row_pixels[c], c] = top_pixel_values[c]
        empty_pixels = depth < 0.1
        dilated = cv2.dilate(depth, np.ones((31, 31), np.uint8))
        depth[empty_pixels] = dilated[empty_pixels]; st["fill31"] = depth.copy()
    depth = cv2.medianBlur(depth, 5); st["median5"] = depth.copy()
    if blur_type == "bilateral":
        depth = cv2.bilateralFilter(depth, 5, 1.5, 2.0)
    elif blur_type == "gaussian":
        valid_pixels = (depth > 0.1)
        blurred = cv2.GaussianBlur(depth, (5, 5), 0)
        depth[valid_pixels] = blurred[valid_pixels]
    st["blur"] = depth.copy()
    valid_pixels = (depth > 0.1)
    depth[valid_pixels] = max_depth - depth[valid_pixels]
    st["out_m"] = depth
    st["out_mm"] = (depth * 1000).astype(np.uint16)                # predict_ros.py:41
    return st


def main():
    import cv2
    os.makedirs(OUT, exist_ok=True)
    # 1. INTER_NEAREST source index for every source size 1..2000 -> 176 (x direction and y direction)
    tx = np.zeros((2000, 176), np.int32); ty = np.zeros((2000, 176), np.int32)
    for src in range(1, 2001):
        ramp = np.arange(src, dtype=np.float32)
        tx[src - 1] = cv2.resize(ramp.reshape(1, src), (176, 1), interpolation=cv2.INTER_NEAREST).reshape(-1).astype(np.int32)
        ty[src - 1] = cv2.resize(ramp.reshape(src, 1), (1, 176), interpolation=cv2.INTER_NEAREST).reshape(-1).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "opencv_resize.npz"), index_x=tx, index_y=ty, version=cv2.__version__)
    # 2. Rodrigues on 10^4 float32 vectors (incl. theta ~ 0, theta ~ pi) as datasets.py:173 calls it
    rng = np.random.default_rng(11)
    r = rng.normal(0, 1.0, (10000, 3)).astype(np.float32)
    r[:100] *= np.float32(1e-9); r[100:200] *= np.float32(1e-4)
    r[200:300] = (r[200:300] / np.linalg.norm(r[200:300], axis=1, keepdims=True) * np.float32(np.pi - 1e-4)).astype(np.float32)
    r[300] = 0
    R = np.stack([cv2.Rodrigues(v)[0] for v in r])
    np.savez_compressed(os.path.join(OUT, "opencv_rodrigues.npz"), rvec=r, R=R, dtype=str(R.dtype), version=cv2.__version__)
    # 3. fill_depth chain
    out = {"version": cv2.__version__}
    for seed, H, W, extrap, blur in FILL_CASES:
        mm = Fx.depth_frame_with_far_wall(7) if seed == "wall7" else Fx.depth_frame_with_holes(seed, H, W)
        st = fill_depth_stages(cv2, mm / 1e3, 2.0, extrap, blur)
        for k, v in st.items():
            out["%s_%s" % (seed, k)] = v
    np.savez_compressed(os.path.join(OUT, "opencv_fill_depth.npz"), **out)
    print("wrote opencv_resize.npz, opencv_rodrigues.npz, opencv_fill_depth.npz to", OUT, "(OpenCV %s)" % cv2.__version__)


if __name__ == "__main__":
    main()
