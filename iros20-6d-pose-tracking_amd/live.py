"""Live-camera front end: the reference's ``predict_ros.TrackerRos`` (predict_ros.py:19-66) without ROS.

The ROS node is transport (subscribers, tf broadcaster) around three pieces of arithmetic, kept here with the
reference's method names so that a rospy / ROS 2 / RealSense wrapper only has to forward messages:

    grab_depth(depth_mm)   uint16 mm frame -> hole-filled uint16 mm (fill_depth, predict_ros.py:38-41) -- se3tn_fill_depth
    grab_color(bgr)        BGR uint8 frame -> RGB (cv2.cvtColor(..., COLOR_BGR2RGB), :43-46)
    on_track()             Tracker.on_track on the latest pair, pose feedback (:48-60); returns what the node
                           publishes on tf: (translation [3], quaternion x,y,z,w, stamp) (:62-66)
"""
import numpy as np


def quaternion_from_matrix(matrix):
    """Rotation part of a 4x4 homogeneous matrix -> unit quaternion (w, x, y, z) the way the `transformations`
    package does it by default (isprecise=False), which is what the reference calls at predict_ros.py:63: the
    eigenvector of the largest eigenvalue of the symmetric 4x4 matrix K built from the rotation (Bar-Itzhack 2000),
    sign chosen so that w >= 0.  Robust to the slightly non-orthonormal R_B the pose update produces (its Rodrigues
    factor is rounded to float32)."""
    M = np.asarray(matrix, dtype=np.float64)[:4, :4]
    m00, m01, m02 = M[0, 0], M[0, 1], M[0, 2]
    m10, m11, m12 = M[1, 0], M[1, 1], M[1, 2]
    m20, m21, m22 = M[2, 0], M[2, 1], M[2, 2]
    K = np.array([[m00 - m11 - m22, 0.0, 0.0, 0.0],
                  [m01 + m10, m11 - m00 - m22, 0.0, 0.0],
                  [m02 + m20, m12 + m21, m22 - m00 - m11, 0.0],
                  [m21 - m12, m02 - m20, m10 - m01, m00 + m11 + m22]]) / 3.0
    w, V = np.linalg.eigh(K)          # eigh reads the lower triangle
    q = V[[3, 0, 1, 2], np.argmax(w)]
    return -q if q[0] < 0.0 else q


class LiveTracker:
    """predict_ros.py:19-66 `TrackerRos` minus the ROS plumbing."""

    def __init__(self, tracker, pose_init, max_depth=2.0, extrapolate=False, blur_type="bilateral"):
        self.tracker = tracker
        self.color = None
        self.depth = None
        self.cur_time = None
        self.A_in_cam = np.asarray(pose_init, np.float64).copy()
        self._fill = dict(max_depth=max_depth, extrapolate=extrapolate, blur_type=blur_type)

    def reset(self, pose_init):
        self.color = None
        self.depth = None
        self.cur_time = None
        self.A_in_cam = np.asarray(pose_init, np.float64).copy()

    def grab_depth(self, depth_mm):
        """depth_mm: HxW array in millimetres (what CvBridge 'passthrough' hands over, cast to uint16)."""
        self.depth = self.tracker.engine.fill_depth(np.asarray(depth_mm).astype(np.uint16), **self._fill)

    def grab_color(self, bgr, stamp=0.0):
        """bgr: HxWx3 uint8 as CvBridge 'bgr8' delivers it; stored as RGB."""
        self.cur_time = stamp
        self.color = np.ascontiguousarray(np.asarray(bgr)[:, :, ::-1])

    def on_track(self):
        if self.color is None or self.depth is None or self.cur_time is None:
            return None
        ob_in_cam = self.tracker.on_track(self.A_in_cam, self.color.astype(np.uint8), self.depth,
                                          gt_A_in_cam=np.eye(4), gt_B_in_cam=np.eye(4), debug=False, samples=1)
        self.A_in_cam = ob_in_cam.copy()
        trans = ob_in_cam[:3, 3]
        q_wxyz = quaternion_from_matrix(ob_in_cam)
        q_xyzw = [q_wxyz[1], q_wxyz[2], q_wxyz[3], q_wxyz[0]]
        return trans, q_xyzw, self.cur_time
