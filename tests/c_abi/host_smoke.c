/* Pure-C host of the hot path: proves the drop-in boundary is the C ABI of include/se3tracknet.h (no
 * Python, no torch).  Reads a checkpoint dump + one batch of NCHW inputs + poses written by
 * tests/test_c_host.py, runs se3tn_infer, writes trans | rot | poseB.
 *
 *   host_smoke <weights.bin> <inputs.bin> <outputs.bin>
 *   weights.bin: int32 n_tensors, then per tensor: int32 key_len, key bytes, int32 ndim, int64 shape[ndim],
 *                float32 data
 *   inputs.bin : int32 n, float32 A[n,4,176,176], float32 B[n,4,176,176], float64 poseA[n,16]
 *   outputs.bin: float32 trans[n,3], float32 rot[n,3], float64 poseB[n,16]
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "se3tracknet.h"

#define CHECK(x)                                                                      \
  do {                                                                                \
    int rc_ = (x);                                                                    \
    if (rc_ != 0) {                                                                   \
      fprintf(stderr, "%s failed (rc=%d): %s\n", #x, rc_, se3tn_last_error());        \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

static int rd(void* dst, size_t bytes, FILE* f) { return fread(dst, 1, bytes, f) == bytes ? 0 : 1; }

int main(int argc, char** argv) {
  if (argc != 4) { fprintf(stderr, "usage: %s weights.bin inputs.bin outputs.bin\n", argv[0]); return 2; }
  FILE* fw = fopen(argv[1], "rb");
  FILE* fi = fopen(argv[2], "rb");
  if (!fw || !fi) { perror("open"); return 2; }
  int32_t n;
  if (rd(&n, 4, fi)) return 2;

  se3tn_ctx* ctx = NULL;
  CHECK(se3tn_create(0, n, &ctx));
  int32_t nt;
  if (rd(&nt, 4, fw)) return 2;
  for (int t = 0; t < nt; ++t) {
    int32_t klen, ndim;
    char key[256];
    int64_t shape[4];
    if (rd(&klen, 4, fw) || klen > 255 || rd(key, (size_t)klen, fw)) return 2;
    key[klen] = 0;
    if (rd(&ndim, 4, fw) || ndim > 4 || rd(shape, 8 * (size_t)ndim, fw)) return 2;
    size_t count = 1;
    for (int d = 0; d < ndim; ++d) count *= (size_t)shape[d];
    float* data = (float*)malloc(count * 4);
    if (rd(data, count * 4, fw)) return 2;
    CHECK(se3tn_set_tensor(ctx, key, data, shape, ndim));
    free(data);
  }
  CHECK(se3tn_pack_weights(ctx));
  CHECK(se3tn_upload_weights(ctx, NULL));
  CHECK(se3tn_set_normalizers(ctx, 0.03, 5 * 3.14159265358979323846 / 180));
  /* the round-4 switches from plain C: the reference's pinned NumPy rounding is the default; AUTO tile selection */
  if (se3tn_get_offset_rule(ctx) != SE3TN_OFFSET_RULE_NUMPY1) { fprintf(stderr, "offset rule default\n"); return 1; }
  CHECK(se3tn_set_offset_rule(ctx, SE3TN_OFFSET_RULE_NUMPY2));
  CHECK(se3tn_set_offset_rule(ctx, SE3TN_OFFSET_RULE_NUMPY1));
  CHECK(se3tn_set_winograd(ctx, SE3TN_WINOGRAD_DEFAULT_MIN_BATCH, SE3TN_WINOGRAD_TILE_6_4));
  CHECK(se3tn_set_winograd(ctx, SE3TN_WINOGRAD_DEFAULT_MIN_BATCH, SE3TN_WINOGRAD_TILE_AUTO));

  const size_t img = (size_t)n * 4 * 176 * 176 * 4;
  float *hA = (float*)malloc(img), *hB = (float*)malloc(img);
  double* hP = (double*)malloc((size_t)n * 16 * 8);
  if (rd(hA, img, fi) || rd(hB, img, fi) || rd(hP, (size_t)n * 16 * 8, fi)) return 2;
  float *dA, *dB, *dT, *dR;
  double *dPA, *dPB;
  if (hipMalloc((void**)&dA, img) || hipMalloc((void**)&dB, img) || hipMalloc((void**)&dT, n * 12) ||
      hipMalloc((void**)&dR, n * 12) || hipMalloc((void**)&dPA, n * 128) || hipMalloc((void**)&dPB, n * 128)) return 3;
  hipMemcpy(dA, hA, img, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, img, hipMemcpyHostToDevice);
  hipMemcpy(dPA, hP, (size_t)n * 128, hipMemcpyHostToDevice);
  CHECK(se3tn_infer(ctx, dA, dB, n, SE3TN_NCHW, dT, dR, dPA, dPB, NULL));
  if (hipDeviceSynchronize() != hipSuccess) return 3;

  float* oT = (float*)malloc((size_t)n * 12);
  float* oR = (float*)malloc((size_t)n * 12);
  double* oP = (double*)malloc((size_t)n * 128);
  hipMemcpy(oT, dT, (size_t)n * 12, hipMemcpyDeviceToHost);
  hipMemcpy(oR, dR, (size_t)n * 12, hipMemcpyDeviceToHost);
  hipMemcpy(oP, dPB, (size_t)n * 128, hipMemcpyDeviceToHost);
  FILE* fo = fopen(argv[3], "wb");
  fwrite(oT, 1, (size_t)n * 12, fo); fwrite(oR, 1, (size_t)n * 12, fo); fwrite(oP, 1, (size_t)n * 128, fo);
  fclose(fo);
  /* host-side entry points from C as well */
  int32_t vu[8];
  const double K[9] = {1066.778, 0, 312.9869, 0, 1067.487, 241.3109, 0, 0, 1};
  CHECK(se3tn_compute_bbox(hP, K, 250.0, vu));
  /* live-camera front end and crop_bbox from C: a 64 x 80 depth ramp with holes through se3tn_fill_depth (no blur:
   * selections only, exact), then its centre window through se3tn_crop_raw */
  enum { FH = 64, FW = 80 };
  uint16_t hd[FH * FW], hf[FH * FW];
  uint8_t hrgb[FH * FW * 3];
  for (int y = 0; y < FH; ++y)
    for (int x = 0; x < FW; ++x) {
      hd[y * FW + x] = (uint16_t)(((x * 7 + y * 3) % 11 == 0) ? 0 : 600 + 3 * x + 2 * y);
      for (int c = 0; c < 3; ++c) hrgb[(y * FW + x) * 3 + c] = (uint8_t)((x + 2 * y + 5 * c) & 255);
    }
  uint16_t *dD, *dF, *dCd;
  uint8_t *dRgb, *dCr;
  if (hipMalloc((void**)&dD, sizeof(hd)) || hipMalloc((void**)&dF, sizeof(hd)) || hipMalloc((void**)&dRgb, sizeof(hrgb)) ||
      hipMalloc((void**)&dCr, 176 * 176 * 3) || hipMalloc((void**)&dCd, 176 * 176 * 2)) return 3;
  hipMemcpy(dD, hd, sizeof(hd), hipMemcpyHostToDevice);
  hipMemcpy(dRgb, hrgb, sizeof(hrgb), hipMemcpyHostToDevice);
  CHECK(se3tn_fill_depth(ctx, dD, FH, FW, 2.0, 0, SE3TN_BLUR_NONE, dF, NULL, NULL));
  se3tn_crop crop;
  memset(&crop, 0, sizeof(crop));
  crop.rgb = dRgb; crop.depth = dF; crop.H = FH; crop.W = FW;
  crop.left = 10; crop.top = -6; crop.right = 70; crop.bottom = 54;      /* leaves the frame at the top */
  uint16_t* hc = (uint16_t*)malloc(176 * 176 * 2);
  CHECK(se3tn_crop_raw(ctx, &crop, dCr, dCd, NULL));
  if (hipDeviceSynchronize() != hipSuccess) return 3;
  hipMemcpy(hf, dF, sizeof(hf), hipMemcpyDeviceToHost);
  hipMemcpy(hc, dCd, 176 * 176 * 2, hipMemcpyDeviceToHost);
  unsigned long long sum_f = 0, sum_c = 0;
  int holes = 0;
  for (int i = 0; i < FH * FW; ++i) { sum_f += hf[i]; holes += hf[i] == 0; }
  for (int i = 0; i < 176 * 176; ++i) sum_c += hc[i];
  printf("ok n=%d version=\"%s\" bbox0=(%d,%d) fill_sum=%llu fill_holes=%d crop_sum=%llu\n", n, se3tn_version(), vu[0], vu[1],
         sum_f, holes, sum_c);
  /* one frame of Tracker.on_track in ONE call from C (round 5): an octahedron as the model, the 64 x 80 synthetic frame above as the
   * camera image (HOST pointers), a small pin-hole camera.  Printed: the 4x4 estimate and checksums of image A. */
  {
    const float ov[18] = {0.05f, 0, 0, -0.05f, 0, 0, 0, 0.05f, 0, 0, -0.05f, 0, 0, 0, 0.05f, 0, 0, -0.05f};
    const float on[18] = {1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1};
    const float oc[18] = {1.0f, 0.2f, 0.2f, 0.2f, 1.0f, 0.2f, 0.2f, 0.2f, 1.0f, 0.9f, 0.9f, 0.1f, 0.1f, 0.9f, 0.9f, 0.9f, 0.1f, 0.9f};
    const int32_t of[24] = {0, 2, 4, 2, 1, 4, 1, 3, 4, 3, 0, 4, 2, 0, 5, 1, 2, 5, 3, 1, 5, 0, 3, 5};
    const double Kc[9] = {100.0, 0, 40.0, 0, 100.0, 32.0, 0, 0, 1};
    const double mean[8] = {100, 110, 120, 900, 90, 95, 105, 850}, stdv[8] = {40, 45, 50, 300, 35, 42, 48, 280};
    const double P0[16] = {0.8, -0.6, 0, 0.01, 0.6, 0.8, 0, -0.005, 0, 0, 1, 0.6, 0, 0, 0, 1};
    double pose[16];
    float tr[3], ro[3];
    int32_t bb[8];
    se3tn_mesh* mesh = NULL;
    uint8_t* dIA; uint16_t* dDA;
    if (hipMalloc((void**)&dIA, 176 * 176 * 3) || hipMalloc((void**)&dDA, 176 * 176 * 2)) return 3;
    CHECK(se3tn_set_normalization(ctx, mean, stdv));
    CHECK(se3tn_mesh_create(ctx, ov, on, oc, 6, of, 8, &mesh));
    CHECK(se3tn_on_track(ctx, mesh, P0, Kc, 150.0, hrgb, hd, FH, FW, dIA, dDA, pose, tr, ro, bb, NULL));
    uint8_t* hIA = (uint8_t*)malloc(176 * 176 * 3);
    uint16_t* hDA = (uint16_t*)malloc(176 * 176 * 2);
    hipMemcpy(hIA, dIA, 176 * 176 * 3, hipMemcpyDeviceToHost);
    hipMemcpy(hDA, dDA, 176 * 176 * 2, hipMemcpyDeviceToHost);
    unsigned long long sa = 0, sd = 0;
    for (int i = 0; i < 176 * 176 * 3; ++i) sa += hIA[i] * (unsigned long long)(1 + i % 7);
    for (int i = 0; i < 176 * 176; ++i) sd += hDA[i] * (unsigned long long)(1 + i % 5);
    printf("track imageA_rgb=%llu imageA_depth=%llu bbox=%d,%d,%d,%d,%d,%d,%d,%d pose=", sa, sd, bb[0], bb[1], bb[2], bb[3], bb[4], bb[5], bb[6], bb[7]);
    for (int i = 0; i < 16; ++i) printf("%.17g%s", pose[i], i == 15 ? "\n" : ",");
    /* n tracks in ONE call from C (round 6): two poses of the same model over the same frame; pair 0 is the pose above */
    {
      const double P2[32] = {0.8, -0.6, 0, 0.01, 0.6, 0.8, 0, -0.005, 0, 0, 1, 0.6, 0, 0, 0, 1,
                             1, 0, 0, -0.02, 0, 0.8, -0.6, 0.015, 0, 0.6, 0.8, 0.55, 0, 0, 0, 1};
      const uint8_t* frames_rgb[2] = {hrgb, hrgb};
      const uint16_t* frames_d[2] = {hd, hd};
      double pose2[32];
      float tr2[6], ro2[6];
      int32_t bb2[16];
      CHECK(se3tn_on_track_batch(ctx, mesh, 2, P2, Kc, 150.0, frames_rgb, frames_d, FH, FW, NULL, NULL, pose2, tr2, ro2, bb2, NULL));
      printf("trackbatch bbox1=%d,%d,%d,%d,%d,%d,%d,%d pose0=", bb2[8], bb2[9], bb2[10], bb2[11], bb2[12], bb2[13], bb2[14], bb2[15]);
      for (int i = 0; i < 16; ++i) printf("%.17g%s", pose2[i], i == 15 ? " pose1=" : ",");
      for (int i = 0; i < 16; ++i) printf("%.17g%s", pose2[16 + i], i == 15 ? "\n" : ",");
    }
    se3tn_mesh_destroy(mesh);
  }
  se3tn_destroy(ctx);
  return 0;
}
