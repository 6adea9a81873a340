"""TEST INFRASTRUCTURE ONLY -- the CPU oracle for the se(3)-TrackNet hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker / the timed CPU baseline -- never as a fallback for
the HIP path (the product raises when its HIP extension is missing).

Pinning status (see DESIGN.md "Oracle"):
  * network forward, depth offset / normalise / pack, bbox, pose composition:
    pinned against the reference's own Python code executed in the build
    container (``oracle/make_golden.py`` -> ``tests/golden/*.npz``).
  * ``cv2.resize(INTER_NEAREST)`` and ``cv2.Rodrigues``: the reference calls
    OpenCV (unpinned ``opencv-python``, docker/dockerfile:25) which is absent
    offline; these two rules are restated from OpenCV's published algorithm
    -> **parity unpinned** for those two functions.
"""
