"""CPU restatement of the se(3)-TrackNet per-frame hot path (TEST INFRASTRUCTURE ONLY).

Every function cites the reference file:line it follows (paths relative to the
upstream repo wenbowen123/iros20-6d-pose-tracking).  torch-CPU fp32 is used for the
network (the north star defines parity against "the reference PyTorch-CPU path");
numpy for the byte / integer / float64 pre- and post-processing.

Pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which
oracle/make_golden.py produced by running the reference's own code.
cv2.resize(NEAREST) and cv2.Rodrigues are restated from OpenCV's published
algorithm: parity unpinned for those two (OpenCV is not available offline).
"""
from collections import OrderedDict
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm2d default, network_modules.py:64 / :96

# (name, C_in, C_out, kernel)   se3_tracknet.py:57-78
_CONVBN = [
    ("convA1", 4, 64, 7), ("convB1", 4, 64, 7),
    ("convAB1", 128, 256, 3),
    ("trans_conv1", 256, 512, 3), ("rot_conv1", 256, 512, 3),
]
_BLOCKS = [
    ("convA2", 64), ("convB2", 64), ("convB3", 64),
    ("convAB2", 256), ("trans_conv2", 512), ("rot_conv2", 512),
]
_HEADS = ["trans_out", "rot_out"]


def state_dict_spec():
    """Ordered (key, shape, dtype) list of Se3TrackNet.state_dict()
    (se3_tracknet.py:52-78, network_modules.py:59-66,86-101)."""
    spec = []

    def bn(prefix, c):
        spec.append((prefix + ".weight", (c,), torch.float32))
        spec.append((prefix + ".bias", (c,), torch.float32))
        spec.append((prefix + ".running_mean", (c,), torch.float32))
        spec.append((prefix + ".running_var", (c,), torch.float32))
        spec.append((prefix + ".num_batches_tracked", (), torch.int64))

    for name, cin, cout, k in _CONVBN:
        spec.append((name + ".0.weight", (cout, cin, k, k), torch.float32))
        spec.append((name + ".0.bias", (cout,), torch.float32))
        bn(name + ".1", cout)
    for name, c in _BLOCKS:
        for i in (1, 2):
            spec.append(("%s.conv%d.weight" % (name, i), (c, c, 3, 3), torch.float32))
            spec.append(("%s.conv%d.bias" % (name, i), (c,), torch.float32))
        for i in (1, 2):
            bn("%s.bn%d" % (name, i), c)
    for name in _HEADS:
        spec.append((name + ".0.weight", (3, 512), torch.float32))
        spec.append((name + ".0.bias", (3,), torch.float32))
    return spec


def make_state_dict(seed=0, head_gain=0.05):
    """Deterministic random-init weights on the reference's state_dict surface.

    NOT the reference's init: He-normal conv weights and *randomised* BN
    statistics / affine (default BN stats gamma=1,beta=0,mu=0,var=1 would make a
    BN-folding bug invisible -- SURVEY.md section 4).  Keys/shapes/dtypes are exactly
    Se3TrackNet.state_dict()'s (checked by make_golden.py with strict loading).
    """
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for key, shape, dtype in state_dict_spec():
        if dtype == torch.int64:
            sd[key] = torch.tensor(1000, dtype=torch.int64)
        elif key.endswith("running_var"):
            sd[key] = torch.rand(shape, generator=g) * 1.0 + 0.5
        elif key.endswith("running_mean"):
            sd[key] = torch.randn(shape, generator=g) * 0.2
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[key] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif len(shape) == 2:
            sd[key] = torch.randn(shape, generator=g) * (head_gain / math.sqrt(shape[1]))
        elif ".bn" in key or key.split(".")[-2] == "1":
            # BN affine: gamma ~ U(0.6, 1.4), beta ~ N(0, 0.1)
            if key.endswith("weight"):
                sd[key] = torch.rand(shape, generator=g) * 0.8 + 0.6
            else:
                sd[key] = torch.randn(shape, generator=g) * 0.1
        else:  # conv / fc bias
            sd[key] = torch.randn(shape, generator=g) * 0.05
    return sd


# ----------------------------------------------------------------------------------
# network  (se3_tracknet.py:81-112)
# ----------------------------------------------------------------------------------
def _bn(sd, p, x):
    # nn.BatchNorm2d in eval mode: (x-mu)/sqrt(var+eps)*gamma+beta
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _conv_bn_selu(sd, name, x, stride):
    # ConvBNReLU = Conv2d(pad=(k-1)//2) + BN + **SELU**  network_modules.py:59-66
    w = sd[name + ".0.weight"]
    x = F.conv2d(x, w, sd[name + ".0.bias"], stride=stride, padding=(w.shape[-1] - 1) // 2)
    return F.selu(_bn(sd, name + ".1", x))


def _basic_block(sd, name, x):
    # ResnetBasicBlock.forward  network_modules.py:103-120 (no downsample, stride 1)
    out = F.conv2d(x, sd[name + ".conv1.weight"], sd[name + ".conv1.bias"], padding=1)
    out = F.relu(_bn(sd, name + ".bn1", out))
    out = F.conv2d(out, sd[name + ".conv2.weight"], sd[name + ".conv2.bias"], padding=1)
    out = _bn(sd, name + ".bn2", out)
    return F.relu(out + x)


@torch.no_grad()
def forward(sd, A, B, intermediates=False):
    """Se3TrackNet.forward (se3_tracknet.py:81-112). A,B: float32 [N,4,H,W] NCHW.
    Returns dict(trans,rot[,feature,trans_logit,rot_logit, per-stage tensors])."""
    out = {}
    a = _conv_bn_selu(sd, "convA1", A, 2)
    if intermediates: out["stemA"] = a
    a = F.max_pool2d(a, 3, 2, 1)
    if intermediates: out["poolA"] = a
    a = _basic_block(sd, "convA2", a)
    b = _conv_bn_selu(sd, "convB1", B, 2)
    if intermediates: out["stemB"] = b
    b = F.max_pool2d(b, 3, 2, 1)
    b = _basic_block(sd, "convB2", b)
    b = _basic_block(sd, "convB3", b)
    ab = torch.cat((a, b), 1).contiguous()
    if intermediates: out["cat"] = ab
    ab = _conv_bn_selu(sd, "convAB1", ab, 2)
    if intermediates: out["ab1"] = ab
    ab = _basic_block(sd, "convAB2", ab)
    out["feature"] = ab
    for head in ("trans", "rot"):
        h = _conv_bn_selu(sd, head + "_conv1", ab, 2)
        if intermediates: out[head + "_c1"] = h
        h = _basic_block(sd, head + "_conv2", h)
        if intermediates: out[head + "_c2"] = h
        h = F.adaptive_avg_pool2d(h, 1).reshape(A.shape[0], -1)
        logit = F.linear(h, sd[head + "_out.0.weight"], sd[head + "_out.0.bias"])
        out[head + "_logit"] = logit
        out[head] = torch.tanh(logit).contiguous()
    return out


# ----------------------------------------------------------------------------------
# pre-processing  (Utils.py:302-359, data_augmentation.py:124-189, datasets.py:115-136)
# ----------------------------------------------------------------------------------
def compute_bbox(pose, K, scale_size=230, scale=(1, 1, 1)):
    """Utils.py:302-316. float64 projection of the 4 corners of an object-centred
    square, np.round (half-to-even) -> int32 [4,2] (v,u)."""
    obj_x = pose[0, 3] * scale[0]
    obj_y = pose[1, 3] * scale[1]
    obj_z = pose[2, 3] * scale[2]
    offset = scale_size / 2
    points = np.ndarray((4, 3), dtype=np.float64)
    points[0] = [obj_x - offset, obj_y - offset, obj_z]
    points[1] = [obj_x - offset, obj_y + offset, obj_z]
    points[2] = [obj_x + offset, obj_y - offset, obj_z]
    points[3] = [obj_x + offset, obj_y + offset, obj_z]
    vus = np.zeros((4, 2))
    vus[:, 1] = points[:, 0] * K[0, 0] / points[:, 2] + K[0, 2]
    vus[:, 0] = points[:, 1] * K[1, 1] / points[:, 2] + K[1, 2]
    return np.round(vus).astype(np.int32)


def resize_nearest_indices(dst, src):
    """OpenCV resizeNN source index table (imgproc/resize.cpp):
    ifx = 1/(dst/src) in double; sx = min(floor(x*ifx), src-1).  PARITY UNPINNED."""
    inv_scale = float(dst) / float(src)
    ifx = 1.0 / inv_scale
    return np.minimum(np.floor(np.arange(dst, dtype=np.float64) * ifx).astype(np.int64), src - 1)


def resize_nearest(img, dsize):
    """cv2.resize(img, (w,h), interpolation=cv2.INTER_NEAREST) restated."""
    dw, dh = dsize
    ys = resize_nearest_indices(dh, img.shape[0])
    xs = resize_nearest_indices(dw, img.shape[1])
    return img[ys][:, xs]


def crop_bbox(color, depth, boundingbox, output_size=(176, 176)):
    """Utils.py:320-359. zero-padded crop + NEAREST resize; depth -> uint16."""
    left = int(np.min(boundingbox[:, 1])); right = int(np.max(boundingbox[:, 1]))
    top = int(np.min(boundingbox[:, 0])); bottom = int(np.max(boundingbox[:, 0]))
    h, w, _ = color.shape
    crop_w = right - left
    crop_h = bottom - top
    color_crop = np.zeros((crop_h, crop_w, 3), dtype=color.dtype)
    depth_crop = np.zeros((crop_h, crop_w), dtype=np.float64)
    top_offset = abs(min(top, 0))
    bottom_offset = min(crop_h - (bottom - h), crop_h)
    right_offset = min(crop_w - (right - w), crop_w)
    left_offset = abs(min(left, 0))
    top = max(top, 0); left = max(left, 0)
    bottom = min(bottom, h); right = min(right, w)
    color_crop[top_offset:bottom_offset, left_offset:right_offset, :] = color[top:bottom, left:right, :]
    depth_crop[top_offset:bottom_offset, left_offset:right_offset] = depth[top:bottom, left:right]
    rgb = resize_nearest(color_crop, output_size)
    dep = resize_nearest(depth_crop, output_size).astype(np.uint16)
    return rgb * (rgb != 0), dep * (dep != 0)


OFFSET_RULE = "numpy1"   # the default of the library too (include/se3tracknet.h: SE3TN_OFFSET_RULE_NUMPY1)


def normalize_depth(depth, pose, offset_rule=None):
    """OffsetDepth.normalize_depth  data_augmentation.py:134-144.  `depth -= pose[2,3]*1000` subtracts a float64 SCALAR from a
    float32 array, and what that means depends on the NumPy generation:
      "numpy1"  value-based casting (every NumPy the reference can run on: it pins Python 3.6 => NumPy <= 1.19, docker/dockerfile:28,
                and Utils.py:307 `np.float` stops importing at 1.24): the scalar is cast to float32, the subtraction is a float32
                operation.  Pinned by tests/golden/preprocess_numpy1.npz (the reference's own class under NumPy 1.26.4,
                oracle/make_numpy1_golden.py);
      "numpy2"  NEP 50: float64 arithmetic, rounded to float32 once.  Pinned by tests/golden/preprocess.npz (made in this image's
                main interpreter).  <= 1 ulp(f32) from "numpy1"."""
    rule = offset_rule or OFFSET_RULE
    depth = depth.astype(np.float32)
    invalid = np.logical_or(depth <= 100, depth >= 2000)
    z_mm = np.float64(pose[2, 3]) * 1000
    if rule == "numpy1":
        z32 = np.float32(z_mm)
        depth = (depth + z32) if pose[2, 3] < 0 else (depth - z32)
        assert depth.dtype == np.float32
    elif rule == "numpy2":
        if pose[2, 3] < 0:
            depth = (depth.astype(np.float64) + z_mm).astype(np.float32)
        else:
            depth = (depth.astype(np.float64) - z_mm).astype(np.float32)
    else:
        raise ValueError(rule)
    depth[invalid] = 2000
    return depth


def normalize_channels(rgb_hwc_f32, depth_f32, mean, std):
    """NormalizeChannels.normalize_channels data_augmentation.py:160-164 followed by
    ToTensor (data_augmentation.py:179-185): float64 math, stored as float32 [4,H,W]."""
    mean = np.asarray(mean, np.float64); std = np.asarray(std, np.float64)
    rgb = rgb_hwc_f32.transpose(2, 0, 1)
    rgb = (rgb - mean[:3, None, None]) / std[:3, None, None]
    d = (depth_f32 - mean[3]) / std[3]
    buf = np.zeros((4,) + depth_f32.shape, np.float32)
    buf[0:3] = rgb
    buf[3] = d
    return buf


def process_data(rgbA, depthA, A_in_cam, rgbB, depthB, mean, std, offset_rule=None):
    """TrackDataset.processData with the eval posttransforms
    OffsetDepth -> NormalizeChannels -> ToTensor (datasets.py:115-136, predict.py:189).
    Both depths are offset by poseA (data_augmentation.py:130-131).
    Returns (dataA, dataB) float32 [4,H,W]."""
    dA = normalize_depth(depthA, A_in_cam, offset_rule)
    dB = normalize_depth(depthB, A_in_cam, offset_rule)
    a = normalize_channels(rgbA.astype(np.float32), dA, mean[:4], std[:4])
    b = normalize_channels(rgbB.astype(np.float32), dB, mean[4:], std[4:])
    return a, b


# ----------------------------------------------------------------------------------
# pose update  (datasets.py:159-175)
# ----------------------------------------------------------------------------------
def rodrigues(rvec):
    """cv2.Rodrigues(vec3)->3x3 restated from OpenCV calib3d (double math, result
    cast to the input depth).  PARITY UNPINNED."""
    src = np.asarray(rvec)
    r = src.astype(np.float64).reshape(3)
    theta = math.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
    if theta < np.finfo(np.float64).eps:
        R = np.eye(3)
    else:
        c = math.cos(theta); s = math.sin(theta); c1 = 1.0 - c
        it = 1.0 / theta
        x, y, z = r[0] * it, r[1] * it, r[2] * it
        rrt = np.array([[x * x, x * y, x * z], [x * y, y * y, y * z], [x * z, y * z, z * z]])
        rx = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
        R = c * np.eye(3) + c1 * rrt + s * rx
    return R.astype(src.dtype if src.dtype in (np.float32, np.float64) else np.float64)


def process_predict(A_in_cam, trans_pred, rot_pred, trans_normalizer=0.03,
                    rot_normalizer=5 * np.pi / 180):
    """TrackDataset.processPredict datasets.py:159-175.
    t_B = f32(trans)*f32(tn) + t_A (f64);  R_B = f32(Rodrigues(f32(rot)*f32(rn))) . R_A."""
    B = np.eye(4)
    t = np.asarray(trans_pred, np.float32) * np.float32(trans_normalizer)
    B[:3, 3] = t + A_in_cam[:3, 3]
    r = np.asarray(rot_pred, np.float32) * np.float32(rot_normalizer)
    R = rodrigues(r).reshape(3, 3)
    B[:3, :3] = R.dot(A_in_cam[:3, :3])
    return B


def on_track(sd, prev_pose, rgb, depth, rgbA, depthA, K, object_width, mean, std,
             trans_normalizer=0.03, rot_normalizer=5 * np.pi / 180, offset_rule=None):
    """Composition of the inner functions of Tracker.on_track (predict.py:217-296) with the
    rendered (rgbA, depthA) supplied by the caller (the renderer is out of scope)."""
    bb = compute_bbox(prev_pose, K, object_width, scale=(1000, 1000, 1000))
    rgbB, depthB = crop_bbox(rgb, depth, bb, (rgbA.shape[1], rgbA.shape[0]))
    a, b = process_data(rgbA, depthA, prev_pose, rgbB, depthB, mean, std, offset_rule)
    out = forward(sd, torch.from_numpy(a)[None], torch.from_numpy(b)[None])
    pose = process_predict(prev_pose, out["trans"][0].numpy(), out["rot"][0].numpy(),
                           trans_normalizer, rot_normalizer)
    return pose, dict(bbox=bb, dataA=a, dataB=b, trans=out["trans"][0].numpy(),
                      rot=out["rot"][0].numpy(), trans_logit=out["trans_logit"][0].numpy(),
                      rot_logit=out["rot_logit"][0].numpy())
