// Device-side pose update shared by the tail kernels (kernels_misc.hip, wino_mfma.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace se3tn {

// TrackDataset.processPredict (datasets.py:159-175) for one pair: outv = tanh outputs (trans 0-2, rot 3-5)
__device__ __forceinline__ void pose_compose(const float* outv, const double* __restrict__ A, double* __restrict__ B,
                                             double tn, double rn) {
    // f32 array * python-float normaliser stays f32 in NumPy (datasets.py:169,172)
    const float tf = (float)tn, rf = (float)rn;
    const float tx = outv[0] * tf, ty = outv[1] * tf, tz = outv[2] * tf;
    const double rx = (double)(outv[3] * rf), ry = (double)(outv[4] * rf), rz = (double)(outv[5] * rf);
    // cv2.Rodrigues(vec3 float32): double math, result cast back to float32
    double R[9];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < 2.220446049250313e-16) {
      R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    } else {
      const double c = cos(theta), sn = sin(theta), c1 = 1.0 - c, it = 1.0 / theta;
      const double x = rx * it, y = ry * it, z = rz * it;
      R[0] = c + c1 * x * x;      R[1] = c1 * x * y - sn * z;  R[2] = c1 * x * z + sn * y;
      R[3] = c1 * x * y + sn * z; R[4] = c + c1 * y * y;       R[5] = c1 * y * z - sn * x;
      R[6] = c1 * x * z - sn * y; R[7] = c1 * y * z + sn * x;  R[8] = c + c1 * z * z;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = (double)(float)R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        B[i * 4 + j] = R[i * 3 + 0] * A[0 * 4 + j] + R[i * 3 + 1] * A[1 * 4 + j] + R[i * 3 + 2] * A[2 * 4 + j];
    B[3] = (double)tx + A[3];
    B[7] = (double)ty + A[7];
    B[11] = (double)tz + A[11];
    B[12] = 0.0; B[13] = 0.0; B[14] = 0.0; B[15] = 1.0;
}

}  // namespace se3tn
