"""The boundary is a C ABI: a pure-C host program (tests/c_abi/host_smoke.c, gcc + the shared library +
the HIP runtime, no Python in the process) loads a checkpoint dump, runs se3tn_infer and must produce
what the oracle computes."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from oracle import fixtures as Fx
from oracle import se3_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "iros20-6d-pose-tracking_amd")


def _build(tmp):
    exe = os.path.join(tmp, "host_smoke")
    cmd = ["gcc", "-O1", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "c_abi", "host_smoke.c"), "-o", exe, "-L" + LIBDIR, "-lse3tracknet",
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_c_host_compiles_and_links(tmp_path):
    """CPU: header is valid C, every used symbol resolves against the built library."""
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    assert os.path.isfile(_build(str(tmp_path)))


@pytest.mark.gpu
def test_c_host_matches_oracle(tmp_path):
    exe = _build(str(tmp_path))
    sd = O.make_state_dict(0)
    n = 2
    A, B = Fx.net_inputs(31, n)
    poses = np.stack([Fx.pose(80 + i, (0.01, 0.02, 0.7)) for i in range(n)])
    with open(tmp_path / "w.bin", "wb") as f:
        tens = [(k, v) for k, v in sd.items() if v.dtype.is_floating_point]
        f.write(struct.pack("<i", len(tens)))
        for k, v in tens:
            kb = k.encode()
            f.write(struct.pack("<i", len(kb))); f.write(kb)
            f.write(struct.pack("<i", v.dim())); f.write(struct.pack("<%dq" % v.dim(), *v.shape))
            f.write(v.numpy().astype("<f4").tobytes())
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<i", n)); f.write(A.numpy().tobytes()); f.write(B.numpy().tobytes()); f.write(poses.tobytes())
    out = subprocess.run([exe, str(tmp_path / "w.bin"), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith("ok n=2")
    # se3tn_fill_depth (no blur: exact) and se3tn_crop_raw called from C, against the oracles on the same synthetic frame
    from oracle import depth_oracle as D
    yy, xx = np.mgrid[0:64, 0:80]
    hd = np.where((xx * 7 + yy * 3) % 11 == 0, 0, 600 + 3 * xx + 2 * yy).astype(np.uint16)
    filled = D.grab_depth(hd, 2.0, False, None)
    rgb = np.stack([(xx + 2 * yy + 5 * c) & 255 for c in range(3)], -1).astype(np.uint8)
    bb = np.array([[-6, 10], [54, 10], [-6, 70], [54, 70]])          # (v, u) corners of the window (10, -6, 70, 54)
    _, crop_d = O.crop_bbox(rgb, filled, bb, (176, 176))
    first = out.stdout.splitlines()[0]
    fields = dict(tok.split("=") for tok in first.split() if "=" in tok and not tok.startswith(("version", "bbox0")))
    assert int(fields["fill_sum"]) == int(filled.astype(np.int64).sum()) and int(fields["fill_holes"]) == int((filled == 0).sum())
    assert int(fields["crop_sum"]) == int(crop_d.astype(np.int64).sum())
    raw = open(tmp_path / "out.bin", "rb").read()
    trans = np.frombuffer(raw[:n * 12], np.float32).reshape(n, 3)
    rot = np.frombuffer(raw[n * 12:n * 24], np.float32).reshape(n, 3)
    poseB = np.frombuffer(raw[n * 24:], np.float64).reshape(n, 4, 4)
    ref = O.forward(sd, A, B)
    assert np.abs(trans - ref["trans"].numpy()).max() < 1e-4 and np.abs(rot - ref["rot"].numpy()).max() < 1e-4
    for i in range(n):
        want = O.process_predict(poses[i], trans[i], rot[i])
        assert np.abs(poseB[i] - want).max() < 1e-12
    # se3tn_on_track from C: the same frame through the oracle (its own render of image A, then the inner functions)
    from oracle import ss_fast as SF
    line = [l for l in out.stdout.splitlines() if l.startswith("track ")][0]
    tf = dict(tok.split("=") for tok in line.split()[1:])
    ov = np.array([[0.05, 0, 0], [-0.05, 0, 0], [0, 0.05, 0], [0, -0.05, 0], [0, 0, 0.05], [0, 0, -0.05]], np.float32)
    on = (ov / 0.05).astype(np.float32)
    oc = np.array([[1.0, .2, .2], [.2, 1.0, .2], [.2, .2, 1.0], [.9, .9, .1], [.1, .9, .9], [.9, .1, .9]], np.float32)
    faces = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]])
    Kc = np.array([[100.0, 0, 40.0], [0, 100.0, 32.0], [0, 0, 1]])
    P0 = np.array([[0.8, -0.6, 0, 0.01], [0.6, 0.8, 0, -0.005], [0, 0, 1, 0.6], [0, 0, 0, 1]])
    mean = np.array([100, 110, 120, 900, 90, 95, 105, 850.0]); std = np.array([40, 45, 50, 300, 35, 42, 48, 280.0])
    bbA = O.compute_bbox(P0, Kc, 150.0, (1000, -1000, 1000))
    win = (int(bbA[:, 1].min()), int(bbA[:, 0].min()), int(bbA[:, 1].max()), int(bbA[:, 0].max()))
    rgbA, depthA = SF.render_vispy(ov, on, oc, faces, P0, Kc, win, numpy_rule="numpy1")
    w7 = 1 + np.arange(rgbA.size) % 7
    w5 = 1 + np.arange(depthA.size) % 5
    assert int(tf["imageA_rgb"]) == int((rgbA.reshape(-1).astype(np.int64) * w7).sum())          # image A: every byte
    assert int(tf["imageA_depth"]) == int((depthA.reshape(-1).astype(np.int64) * w5).sum())
    want, aux = O.on_track(sd, P0, rgb, hd, rgbA, depthA, Kc, 150.0, mean, std)
    assert [int(v) for v in tf["bbox"].split(",")] == aux["bbox"].reshape(-1).tolist()
    got = np.array([float(v) for v in tf["pose"].split(",")]).reshape(4, 4)
    assert np.abs(got - want).max() < 1e-5, np.abs(got - want).max()
    # se3tn_on_track_batch from C (n = 2, the same two-pair kernels as n = 1: bitwise the single call for pair 0; pair 1 vs the oracle)
    lb = [l for l in out.stdout.splitlines() if l.startswith("trackbatch ")][0]
    tb = dict(tok.split("=") for tok in lb.split()[1:])
    pose0 = np.array([float(v) for v in tb["pose0"].split(",")]).reshape(4, 4)
    pose1 = np.array([float(v) for v in tb["pose1"].split(",")]).reshape(4, 4)
    assert np.array_equal(pose0, got)
    P1 = np.array([[1, 0, 0, -0.02], [0, 0.8, -0.6, 0.015], [0, 0.6, 0.8, 0.55], [0, 0, 0, 1.0]])
    bbA = O.compute_bbox(P1, Kc, 150.0, (1000, -1000, 1000))
    win = (int(bbA[:, 1].min()), int(bbA[:, 0].min()), int(bbA[:, 1].max()), int(bbA[:, 0].max()))
    rgbA, depthA = SF.render_vispy(ov, on, oc, faces, P1, Kc, win, numpy_rule="numpy1")
    want1, aux1 = O.on_track(sd, P1, rgb, hd, rgbA, depthA, Kc, 150.0, mean, std)
    assert [int(v) for v in tb["bbox1"].split(",")] == aux1["bbox"].reshape(-1).tolist()
    assert np.abs(pose1 - want1).max() < 1e-5, np.abs(pose1 - want1).max()
