// 3x3 convolution (pad 1, stride 1|2) + folded-BN bias + {ReLU | residual+ReLU | SELU} as an
// im2col-free implicit GEMM on the gfx950 exact-f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces every nn.Conv2d(k=3)+BatchNorm2d(+SELU/ReLU/+identity) of Se3TrackNet
// (se3_tracknet.py:59-76 via network_modules.py:59-66 ConvBNReLU, :86-120 ResnetBasicBlock).
//
// Common to both kernels below
//  * activations are NHWC float32 with a one-pixel ZERO BORDER kept in HBM ([n,H+2,W+2,C]): the
//    conv padding is data, there is no bounds logic anywhere on the operand path;
//  * GEMM rows = output pixels flattened over the batch, cols = Cout, K = (32-channel chunk) x tap,
//    walked chunk-major / tap-minor (weights are packed [chunk][tap][Cout][32] on the host);
//  * MFMA operand roles are SWAPPED (A-operand = weight rows, B-operand = pixel rows): a lane of the
//    accumulator holds ONE pixel x 4 runs of 4 consecutive couts, so bias / residual / activation /
//    store are float4 traffic;
//  * operand tiles are DMA'd global -> LDS with global_load_lds_dwordx4 (no staging VGPRs, no
//    ds_write pass).  A DMA's LDS image is lane-linear, so LDS rows are unpadded 128-byte runs
//    (32 floats) and ds_read_b128 bank conflicts are removed by an XOR swizzle applied on the SOURCE
//    side (which 16-byte slot a lane fetches) and again on the read address:
//        physical slot = logical slot ^ ((row >> 1) & 7);
//  * the DMA is issued from inline asm: for the builtin hipcc (ROCm 7.2) cannot prove that the DMA
//    destination (the other buffer) does not alias the ds_reads of the current buffer and puts
//    `s_waitcnt vmcnt(0)` in front of the first ds_read of every K-step.  The one wait needed
//    (vmcnt(0) before the barrier that ends a K-step) is written by hand;
//  * workgroup id -> (pixel tile, weight panel) with panel = id % panels, panels | 8: id % 8 is the
//    XCD, so an XCD's 4 MB L2 only ever holds its own weight panel(s) (<= 2.4 MB).
//
// conv3x3_slab_kernel  (stride 1: 78 % of the network's FLOPs)
//    256 px x {128|64} cout per 512-thread workgroup (8 waves, 2 per SIMD, one workgroup per CU).
//    Because the zero border is in memory, the 9 taps of a run of consecutive output pixels are 9
//    SHIFTED WINDOWS of one contiguous run of input pixels: per 32-channel chunk that "slab"
//    (tile + (W+3)-pixel halo either side, <= 464 px) is DMA'd ONCE and the taps read it at
//    different offsets -- pixel-side staging traffic drops ~6x vs re-gathering each tap, and
//    the only per-K-step DMA left is the 16 KB weight tile.
// conv3x3_gather_s2_kernel (stride 2: convAB1, trans|rot conv1)
//    128 px x 128 cout per 256-thread workgroup, 2 workgroups per CU; pixel rows are gathered per
//    tap (strided windows have no contiguous slab).
#include "mfma_common.h"

#ifndef SE3TN_PERSISTENT_SLAB
#define SE3TN_PERSISTENT_SLAB 1  // 0: one workgroup per tile (A/B timing only)
#endif
#ifndef SE3TN_SLAB_RELAXED
#define SE3TN_SLAB_RELAXED 0     // 1: the slab piece of the NEXT chunk stays in flight across the K-step barriers (counted vmcnt)
#endif

namespace se3tn {

// shared epilogue: lane holds pixel l31 x couts {8q + 4hh + 0..3} of each 32x32 tile.
// MM_F16X3 accumulators carry the per-cout power-of-two weight scale: acc * wscale[c] first.
// rpre != nullptr: the float32 residual values were loaded ahead of time (same [i][j][q] order)
template <int PT, int CT, int EPI, int MM, int OUTF, int RESF>
__device__ __forceinline__ void store_tiles(const ConvArgs& a, int g, const f32x16 (&acc)[PT][CT],
                                            const int (&opix)[PT], const bool (&ok)[PT], int cbase, int hh,
                                            const float4* rpre = nullptr) {
  const float* __restrict__ bias = a.bias + (size_t)g * a.bias_gs;
  const float* __restrict__ wsc = (MM == MM_F16X3) ? a.wscale + (size_t)g * a.bias_gs : nullptr;
  const float* __restrict__ res = (EPI == 1) ? a.res + (size_t)g * a.res_gs : nullptr;
  float* __restrict__ out = a.out + (size_t)g * a.out_gs;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    if (!ok[i]) continue;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = cbase + j * 32 + q * 8 + hh * 4;
        float4 v = make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2],
                               acc[i][j][4 * q + 3]);
        if (MM == MM_F16X3) {
          const float4 w = *reinterpret_cast<const float4*>(wsc + c);
          v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w;
        }
        const float4 b = *reinterpret_cast<const float4*>(bias + c);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI == 1) {
          if (rpre) r = rpre[(i * CT + j) * 4 + q];
          else
            r = (RESF == FMT_SPLIT) ? load_split4(res, (size_t)opix[i], a.res_ld, c)
                                    : *reinterpret_cast<const float4*>(res + (size_t)opix[i] * a.res_ld + c);
        }
        v = apply_epilogue<EPI>(v, b, r);
        if (OUTF == FMT_SPLIT) bad |= store_split4(out, (size_t)opix[i], a.out_ld, c, v);
        else *reinterpret_cast<float4*>(out + (size_t)opix[i] * a.out_ld + c) = v;
      }
    }
  }
  if (OUTF == FMT_SPLIT && bad) atomicOr(a.overflow, 1);
}

// =================================================================================================
// stride 1: slab kernel.  8 waves = WM x WN; wave tile = PT x CT tiles of 32x32.
// LDS: [slab 0][slab 1][weights 0][weights 1], slab = SLABPX pixel rows of 32 floats.
// =================================================================================================
template <int CIN, int WM, int WN, int PT, int CT, int SLABPX, int EPI, int MM, int OUTF, int RESF>
__global__ __launch_bounds__(512, 2) void conv3x3_slab_kernel(const ConvArgs a) {
  constexpr int BM = WM * PT * 32, BN = WN * CT * 32;
  constexpr int NCH = CIN / 32;
  constexpr int SLAB = SLABPX * 32;  // floats
  constexpr int WT = BN * 32;        // floats per weight tile
  constexpr int WPIECES = BN / 8;    // 1 KB DMA pieces per weight tile (16 or 8)
  static_assert(WM * WN == 8 && (WPIECES == 16 || WPIECES == 8), "8 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, hh = lane >> 5;

  // PERSISTENT tiles: workgroup b handles tiles b, b + gridDim.x, ...; gridDim.x is a multiple of `panels`
  // (or the whole tile count), so a workgroup keeps its weight panel and the panel <-> XCD affinity.
  // The next tile's first slab and weight tile are DMA'd during the last chunk of the current one, and
  // the epilogue's stores drain under the next tile's first K-steps: no per-tile prologue / epilogue bubble.
  const int panels = a.groups * a.tiles_n;
  const int ntiles = ((a.M + BM - 1) / BM) * panels;
  const int p = blockIdx.x % panels;
  const int g = p / a.tiles_n, nt = p % a.tiles_n;
  const int n0 = nt * BN;
  const float* __restrict__ in = a.in + (size_t)g * a.in_gs;
  const float* __restrict__ wgt = a.w + (size_t)g * a.w_gs + (size_t)n0 * 32;
  const int W = a.W, Wp = W + 2, HW = a.H * W;
  static_assert(NCH % 2 == 0, "chunk 0 of the next tile reuses slab buffer 0 during the last chunk");

  // slab of pixel tile mt = padded-flat pixels [lo, lo + 8 * npieces) covering the tile's pixels +- (Wp + 1)
#define TILE_GEOMETRY(MT, LO, NP)                                                                     \
  {                                                                                                  \
    const int m0_ = (MT) * BM, ml_ = min(m0_ + BM, a.M) - 1;                                          \
    LO = padded_index(m0_, HW, W) - (Wp + 1);                                                         \
    NP = min((padded_index(ml_, HW, W) + (Wp + 1) + 1 - LO + 7) >> 3, SLABPX / 8);                    \
  }
  int tile = blockIdx.x;
  int lo, npieces;
  TILE_GEOMETRY(tile / panels, lo, npieces)

  // ---- DMA geometry ---------------------------------------------------------------------------
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  // slab piece j = 8 pixels: lane l -> pixel 8 j + (l >> 3), LDS slot l & 7 holds channel block
  // (l & 7) ^ f(pixel), f(i) = (i >> 1) & 7 = (4 j + (l >> 4)) & 7: two lane patterns (j even / odd)
  const unsigned sv_even = (unsigned)((lane >> 3) * a.in_ld * 4 + (((lane & 7) ^ ((lane >> 4) & 7)) << 4));
  const unsigned sv_odd = (unsigned)((lane >> 3) * a.in_ld * 4 + (((lane & 7) ^ ((4 + (lane >> 4)) & 7)) << 4));
  // weight piece q = 8 cout rows of the contiguous [BN][32] tile; same swizzle with row = 8 q + (l >> 3)
  const unsigned wv_even = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 4) & 7)) << 4));
  const unsigned wv_odd = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ ((4 + (lane >> 4)) & 7)) << 4));

  // slab piece wid + 8 T of chunk CH into slab buffer SB (this wave owns pieces wid + 8 t, t = 0..8)
#define SLAB_PIECE(LO, NP, CH, SB, T)                                                                \
  {                                                                                                  \
    const int j_ = wid + 8 * (T);                                                                    \
    if (j_ < (NP)) {                                                                                 \
      const float* sb_ = in + (size_t)((LO) + 8 * j_) * a.in_ld + (CH) * 32;                          \
      glds16<0>(sb_, (j_ & 1) ? sv_odd : sv_even, lds0 + (unsigned)(((SB) * SLAB + j_ * 256) * 4));  \
    }                                                                                                \
  }
  // weight tile of K-step (CH, TAP) into weight buffer WB
#define WEIGHT_TILE(CH, TAP, WB)                                                                     \
  {                                                                                                  \
    const float* tb_ = wgt + (size_t)((CH) * 9 + (TAP)) * (a.tiles_n * BN) * 32;                     \
    const unsigned ld_ = lds0 + (unsigned)((2 * SLAB + (WB) * WT) * 4);                              \
    if (WPIECES == 16) {                                                                             \
      glds16<0>(tb_ + wid * 512, wv_even, ld_ + wid * 2048);                                         \
      glds16<1024>(tb_ + wid * 512, wv_odd, ld_ + wid * 2048);                                       \
    } else {                                                                                         \
      glds16<0>(tb_ + wid * 256, (wid & 1) ? wv_odd : wv_even, ld_ + wid * 1024);                    \
    }                                                                                                \
  }

  // weight fragment offsets (floats): lane (l31, hh) reads logical slot 2 kg + hh of row base + l31
  const int X = (l31 >> 1) & 7;
  const int wlo = (hh ^ (X & 1)) * 4, xk = X >> 1;
  const int fo0 = ((0 ^ xk) << 3) + wlo, fo1 = ((1 ^ xk) << 3) + wlo, fo2 = ((2 ^ xk) << 3) + wlo,
            fo3 = ((3 ^ xk) << 3) + wlo;

  // prologue of the first tile: whole slab of chunk 0 + first weight tile
#pragma unroll
  for (int t = 0; t < 9; ++t) SLAB_PIECE(lo, npieces, 0, 0, t)
  WEIGHT_TILE(0, 0, 0)
  wait_dma_and_barrier();

  for (;;) {
    const int m0 = (tile / panels) * BM;
    const int mlast = min(m0 + BM, a.M) - 1;
    // per-lane pixel rows of this wave's tiles
    int ibase[PT];  // slab index of the pixel at the centre tap
    int opix[PT];   // padded-flat index in the output / residual tensor (same geometry as the input)
    bool ok[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int m = m0 + (wm * PT + i) * 32 + l31;
      ok[i] = m < a.M;
      opix[i] = padded_index(ok[i] ? m : mlast, HW, W);
      ibase[i] = opix[i] - lo;
    }
    const int tnext = tile + (int)gridDim.x;
    const bool more = tnext < ntiles;
    int nlo = 0, nnp = 0;
    if (more) TILE_GEOMETRY(tnext / panels, nlo, nnp)

    f32x16 acc[PT][CT];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // residual + ReLU epilogue of the short-K (64-channel) tiles: the residual values are fetched while the last
    // chunk's MFMAs run instead of after them (an L2 / HBM round trip per tile otherwise sits in the epilogue)
    constexpr bool PREFETCH_RES = (EPI == 1 && RESF == FMT_F32 && PT * CT <= 2);
    float4 rpre[PREFETCH_RES ? PT * CT * 4 : 1];

    int kt = 0;
    for (int ch = 0; ch < NCH; ++ch) {
      const int sb = ch & 1;
      if (PREFETCH_RES && ch == NCH - 1) {
        const float* __restrict__ res_ = a.res + (size_t)g * a.res_gs;
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
          for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              rpre[(i * CT + j) * 4 + q] = *reinterpret_cast<const float4*>(
                  res_ + (size_t)opix[i] * a.res_ld + n0 + wn * CT * 32 + j * 32 + q * 8 + hh * 4);
      }
      for (int tap = 0; tap < 9; ++tap, ++kt) {
        const int wb = kt & 1;
        // next K-step's weights, and this wave's share of the next slab (next chunk, or chunk 0 of the
        // workgroup's next tile)
        if (tap < 8) {
          WEIGHT_TILE(ch, tap + 1, wb ^ 1)
        } else if (ch + 1 < NCH) {
          WEIGHT_TILE(ch + 1, 0, wb ^ 1)
        } else if (more) {
          WEIGHT_TILE(0, 0, wb ^ 1)
        }
        // (the weight tile is issued BEFORE the slab piece: a counted wait that leaves one operation in flight leaves the
        // slab piece, which nobody reads before the next chunk)
        bool piece_in_flight = false;
        if (ch + 1 < NCH) {
          SLAB_PIECE(lo, npieces, ch + 1, sb ^ 1, tap)
          piece_in_flight = (wid + 8 * tap) < npieces;
        } else if (more) {
          SLAB_PIECE(nlo, nnp, 0, 0, tap)
          piece_in_flight = (wid + 8 * tap) < nnp;
        }

        const int r = tap / 3, s = tap - r * 3;
        const int tapoff = (r - 1) * Wp + (s - 1);
        // pixel fragment addresses: slab row idx = ibase + tapoff, slot (2 kg + hh) ^ ((idx >> 1) & 7),
        // i.e. float offset (8 kg) ^ py with py = (hh ^ ((idx >> 1) & 7)) * 4
        int prow[PT], py[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) {
          const int idx = ibase[i] + tapoff;
          prow[i] = sb * SLAB + idx * 32;
          py[i] = (hh ^ ((idx >> 1) & 7)) << 2;
        }
        const float* pW = smem + 2 * SLAB + wb * WT + (wn * CT * 32 + l31) * 32;
#define PXF(G) *reinterpret_cast<const float4*>(smem + prow[i] + ((8 * (G)) ^ py[i]))
#define WTF(G) *reinterpret_cast<const float4*>(pW + j * 1024 + ((G) == 0 ? fo0 : (G) == 1 ? fo1 : (G) == 2 ? fo2 : fo3))
        if (MM == MM_F16X3) {
          SE3TN_MMA_SPLIT(PT, CT, PXF, WTF)
        } else {
          SE3TN_MMA_GROUP(PT, CT, PXF(0), WTF(0))
          SE3TN_MMA_GROUP(PT, CT, PXF(1), WTF(1))
          SE3TN_MMA_GROUP(PT, CT, PXF(2), WTF(2))
          SE3TN_MMA_GROUP(PT, CT, PXF(3), WTF(3))
        }
#undef PXF
#undef WTF
        if (kt + 1 < NCH * 9 || more) {
          if (SE3TN_SLAB_RELAXED && MM == MM_F32 && tap < 8) {
            // next K-step reads only the weight tile just fetched: wait for it (operations retire in order), let this wave's
            // slab piece fly on; the whole slab is drained (vmcnt(0)) at the chunk's last K-step, before anybody reads it
            if (piece_in_flight) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
          } else {
            wait_dma_and_barrier();
          }
        }
      }
    }
    store_tiles<PT, CT, EPI, MM, OUTF, RESF>(a, g, acc, opix, ok, n0 + wn * CT * 32, hh, PREFETCH_RES ? rpre : nullptr);
    if (!more) break;
    tile = tnext;
    lo = nlo;
    npieces = nnp;
  }
#undef SLAB_PIECE
#undef WEIGHT_TILE
#undef TILE_GEOMETRY
}

// =================================================================================================
// stride 2: gather kernel, 4 waves = 2 x 2, wave tile 2 x 2 tiles (128 px x 128 cout), 2 WGs per CU
// =================================================================================================
// NW = 4: 128 px x 128 cout, 2 workgroups per CU (small batches: more, smaller tiles);  NW = 8: 256 px x 128 cout, one
// workgroup per CU: 48 KB instead of 2 x 32 KB of operands DMA'd per K-step for the same flops (the DMA, not the MFMA
// stream, is what these kernels lose time to: scripts/probes/mfma_stream.hip)
template <int CIN, int EPI, int MM, int OUTF, int NW>
__global__ __launch_bounds__(NW * 64, 2) void conv3x3_gather_s2_kernel(const ConvArgs a) {
  constexpr int BM = NW * 32, BN = 128, PT = 2, CT = 2;
  constexpr int RS = NW * 8;          // rows staged per DMA step of the whole workgroup
  constexpr int PJ = BM / RS, WJ = BN / RS;
  constexpr int NCH = CIN / 32, KT = NCH * 9;
  constexpr int BUF = (BM + BN) * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, hh = lane >> 5;

  const int panels = a.groups * a.tiles_n;
#if defined(SE3TN_GATHER_XCD_PIXELS)
  // variant (EXPERIMENTS items 12, 42): an XCD owns a RANGE OF PIXEL TILES and walks every weight panel over it (its L2 keeps the
  // input rows; the panels stream), instead of owning weight panels
  const int tiles_m = (a.M + BM - 1) / BM, per_x = (tiles_m + 7) / 8;
  const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int p = q % panels, mt = x * per_x + q / panels;
  if (mt >= tiles_m) return;
#else
  const int p = blockIdx.x % panels, mt = blockIdx.x / panels;
#endif
  const int g = p / a.tiles_n, nt = p % a.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;
  const float* __restrict__ in = a.in + (size_t)g * a.in_gs;
  const float* __restrict__ wgt = a.w + (size_t)g * a.w_gs + (size_t)n0 * 32;
  const int Wp = a.W + 2, Hp = a.H + 2, HoWo = a.Ho * a.Wo;
  const int mlast = a.M - 1;

  // staging: thread t fills LDS slot (t & 7) of rows (t >> 3) + 32 j with channel block
  // (t & 7) ^ ((row >> 1) & 7); in padded coordinates tap (r,s) of output (ho,wo) is input (2ho+r, 2wo+s)
  const int r0 = tid >> 3;
  const int c4 = (tid & 7) ^ ((r0 >> 1) & 7);
  unsigned pvoff[PJ];
#pragma unroll
  for (int j = 0; j < PJ; ++j) {
    const int m = min(m0 + r0 + RS * j, mlast);
    const int n = m / HoWo, rem = m - n * HoWo;
    const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
    pvoff[j] = (unsigned)((((n * Hp + 2 * ho) * Wp + 2 * wo) * a.in_ld + c4 * 4) * 4);
  }
  const unsigned wvoff = (unsigned)((r0 * 32 + c4 * 4) * 4);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

#define ISSUE_TILE(CH, TAP, BUFI)                                                                    \
  {                                                                                                  \
    const int r_ = (TAP) / 3, s_ = (TAP) - r_ * 3;                                                   \
    const float* pb_ = in + (size_t)(r_ * Wp + s_) * a.in_ld + (CH) * 32;                            \
    const unsigned lb_ = lds0 + (unsigned)(((BUFI) * BUF + wid * 256) * 4);                          \
    _Pragma("unroll") for (int j_ = 0; j_ < PJ; ++j_) glds16<0>(pb_, pvoff[j_], lb_ + j_ * RS * 128);  \
    const float* tb_ = wgt + (size_t)((CH) * 9 + (TAP)) * (a.tiles_n * BN) * 32;                     \
    _Pragma("unroll") for (int j_ = 0; j_ < WJ; ++j_)                                                \
        glds16<0>(tb_ + j_ * RS * 32, wvoff, lb_ + BM * 128 + j_ * RS * 128);                        \
  }

  const int X = (l31 >> 1) & 7;
  const int lo = (hh ^ (X & 1)) * 4, xk = X >> 1;
  const int fo0 = ((0 ^ xk) << 3) + lo, fo1 = ((1 ^ xk) << 3) + lo, fo2 = ((2 ^ xk) << 3) + lo,
            fo3 = ((3 ^ xk) << 3) + lo;

  f32x16 acc[PT][CT];
#pragma unroll
  for (int i = 0; i < PT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  ISSUE_TILE(0, 0, 0)
  wait_dma_and_barrier();

  int ch = 0, tap = 0;
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (++tap == 9) { tap = 0; ++ch; }
    if (kt + 1 < KT) ISSUE_TILE(ch, tap, buf ^ 1)
    const float* pP = smem + buf * BUF + (wm * PT * 32 + l31) * 32;
    const float* pW = smem + buf * BUF + (BM + wn * CT * 32 + l31) * 32;
#define FOG(G) ((G) == 0 ? fo0 : (G) == 1 ? fo1 : (G) == 2 ? fo2 : fo3)
#define PXF(G) *reinterpret_cast<const float4*>(pP + i * 1024 + FOG(G))
#define WTF(G) *reinterpret_cast<const float4*>(pW + j * 1024 + FOG(G))
    if (MM == MM_F16X3) {
      SE3TN_MMA_SPLIT(PT, CT, PXF, WTF)
    } else {
      SE3TN_MMA_GROUP(PT, CT, PXF(0), WTF(0))
      SE3TN_MMA_GROUP(PT, CT, PXF(1), WTF(1))
      SE3TN_MMA_GROUP(PT, CT, PXF(2), WTF(2))
      SE3TN_MMA_GROUP(PT, CT, PXF(3), WTF(3))
    }
#undef PXF
#undef WTF
#undef FOG
    if (kt + 1 < KT) wait_dma_and_barrier();
  }
#undef ISSUE_TILE

  int opix[PT];
  bool ok[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int m = m0 + (wm * PT + i) * 32 + l31;
    ok[i] = m < a.M;
    opix[i] = padded_index(ok[i] ? m : mlast, HoWo, a.Wo);
  }
  store_tiles<PT, CT, EPI, MM, OUTF, FMT_F32>(a, g, acc, opix, ok, n0 + wn * CT * 32, hh);
}

// 16-byte store with device (agent) scope: written through the non-coherent per-XCD L2, visible to every compute unit once
// s_waitcnt vmcnt(0) has returned
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_agent_scope(float* p, const float4 v) {
  f32x4_t t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
}

#ifndef SE3TN_SPLITK_WT
#define SE3TN_SPLITK_WT 1   // 1 (shipped): partial sums written through the L2 and device-scope counters -- correct under ANY workgroup ->
                            // XCD mapping (CU masks, other partition modes: ADVICE r5); 0 (measurement builds only, EXPERIMENTS item 41):
                            // left in the tile's own XCD's L2, which ASSUMES blockIdx % 8 is the XCD
#endif
// the arrival counters: device scope (memory side) with write-through partial sums, the XCD's own L2 otherwise
#if SE3TN_SPLITK_WT
#define SE3TN_SEM_SCOPE __HIP_MEMORY_SCOPE_AGENT
#else
#define SE3TN_SEM_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
__device__ __forceinline__ void store_partial(float* p, const float4 v) {
#if SE3TN_SPLITK_WT
  store_agent_scope(p, v);
#else
  *reinterpret_cast<float4*>(p) = v;
#endif
}

// one float4 of the output: the slices' partial sums added in a FIXED order, then bias / residual / activation
template <int EPI, int MM, int OUTF, int RESF>
__device__ __forceinline__ void reduce_element(const ConvArgs& a, const float* src, size_t slice_stride, int g, int m, int c) {
  float4 v = *reinterpret_cast<const float4*>(src);
#pragma unroll 8
  for (int s = 1; s < a.slices; ++s) {  // fixed order: deterministic (unrolled: the loads of 8 slices are in flight together)
    const float4 u = *reinterpret_cast<const float4*>(src + s * slice_stride);
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  if (MM == MM_F16X3) {
    const float4 w = *reinterpret_cast<const float4*>(a.wscale + (size_t)g * a.bias_gs + c);
    v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w;
  }
  const size_t opix = (size_t)padded_index(m, a.Ho * a.Wo, a.Wo);
  const float4 b = *reinterpret_cast<const float4*>(a.bias + (size_t)g * a.bias_gs + c);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (EPI == 1) {
    const float* res = a.res + (size_t)g * a.res_gs;
    r = (RESF == FMT_SPLIT) ? load_split4(res, opix, a.res_ld, c) : *reinterpret_cast<const float4*>(res + opix * a.res_ld + c);
  }
  v = apply_epilogue<EPI>(v, b, r);
  float* out = a.out + (size_t)g * a.out_gs;
  if (OUTF == FMT_SPLIT) {
    if (store_split4(out, opix, a.out_ld, c, v)) atomicOr(a.overflow, 1);
  } else {
    *reinterpret_cast<float4*>(out + opix * a.out_ld + c) = v;
  }
}

template <int EPI, int MM, int OUTF, int RESF>
__global__ __launch_bounds__(256) void conv_reduce_kernel(const ConvArgs a, int cout, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // (group, m, c4)
  if (idx >= total) return;
  const int q4 = cout >> 2;
  const int c = (idx % q4) * 4, t = idx / q4;
  const int g = t / a.M, m = t % a.M;
  reduce_element<EPI, MM, OUTF, RESF>(a, a.part + ((size_t)g * a.M + m) * cout + c, (size_t)a.groups * a.M * cout, g, m, c);
}

// The reduction INSIDE the split-K launch (a.sem != nullptr): the grid carries, after the conv workgroups, a.rpt reduce
// workgroups per output tile (a power of two <= 32: each takes 128 / rpt rows).  A conv workgroup publishes its partial tile (device-scope release) and counts itself into sem[tile];
// a reduce workgroup sleeps until the count reaches `slices`, then sums its rows of the tile in the same fixed slice order as
// conv_reduce_kernel -- bitwise the same result, without the second launch.  MEASURED SLOWER (EXPERIMENTS item 41: 363 vs 268 us per
// batch-1 forward): partial sums that another XCD must see inside the launch have to be written through / read around the per-XCD
// L2, which costs more than the 5-6 us launch it saves.  Kept behind SE3TN_SPLITK_FUSED=1 with its bitwise A/B check.
// No deadlock: conv workgroups never wait, and the reduce workgroups (rpt x tiles <= 256) cannot occupy every workgroup
// slot of the device (launch_splitk checks), whatever order the dispatcher takes.
template <int MM, int BM, int BN>
__device__ __forceinline__ void splitk_reduce_part(const ConvArgs& a, int tile, int seg) {
  const int panels = a.groups * a.tiles_n;
  const int p = tile % panels, mt = tile / panels;
  const int g = p / a.tiles_n, nt = p - g * a.tiles_n;
  const int tid = threadIdx.x;
  if (tid == 0) {
    // (an atomic read-modify-write executes in the XCD's L2, where the conv workgroups' arrivals are counted: no stale L1 line)
    while (__hip_atomic_fetch_add(a.sem + tile, 0, __ATOMIC_RELAXED, SE3TN_SEM_SCOPE) < a.slices) __builtin_amdgcn_s_sleep(1);
    // the last reduce workgroup to see the tile complete re-arms both counters for the next launch
    if (__hip_atomic_fetch_add(a.sem + SE3TN_SPLITK_MAX_TILES + tile, 1, __ATOMIC_RELAXED, SE3TN_SEM_SCOPE) == a.rpt - 1) {
      __hip_atomic_exchange(a.sem + tile, 0, __ATOMIC_RELAXED, SE3TN_SEM_SCOPE);
      __hip_atomic_exchange(a.sem + SE3TN_SPLITK_MAX_TILES + tile, 0, __ATOMIC_RELAXED, SE3TN_SEM_SCOPE);
    }
  }
  __syncthreads();
#if SE3TN_SPLITK_WT
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every thread reads the partial sums other compute units stored
#else
  // the partial sums are in THIS XCD's L2 (written by its other compute units): only this compute unit's vector L1 may hold stale
  // lines of the workspace (an agent-scope acquire would also drop the L2's clean lines under the conv workgroups still running)
  asm volatile("buffer_inv sc0" ::: "memory");
#endif
  // the partial tiles are stored in ACCUMULATOR-FRAGMENT order (conv3x3_splitk_kernel): float4 number
  //   f = (((wave * PT + i) * CT + j) * 4 + q) * 64 + lane   of tile `tile` of slice s at part[(s * tiles + tile) * BM * BN + 4 f]
  // -- every store / load instruction of a wave moves one contiguous KB
  constexpr int PT = 2, CT = BN / 64, E = BM * BN / 4;
  const int tiles = ((a.M + BM - 1) / BM) * panels;
  const int per = E / a.rpt;
  const size_t slice_stride = (size_t)tiles * BM * BN;
  for (int e = tid; e < per; e += 256) {
    const int f = seg * per + e;
    const int lane = f & 63, q = (f >> 6) & 3, t8 = f >> 8;
    const int j = t8 % CT, i = (t8 / CT) % PT, wv = t8 / (CT * PT);
    const int m = mt * BM + ((wv >> 1) * PT + i) * 32 + (lane & 31);
    const int c = nt * BN + ((wv & 1) * CT + j) * 32 + q * 8 + (lane >> 5) * 4;
    if (m >= a.M) continue;
    const float* src = a.part + (size_t)tile * BM * BN + (size_t)f * 4;
    if (MM == MM_F16X3) {
      if (a.epi == 0) reduce_element<0, MM_F16X3, FMT_SPLIT, FMT_F32>(a, src, slice_stride, g, m, c);
      else if (a.epi == 2) reduce_element<2, MM_F16X3, FMT_SPLIT, FMT_F32>(a, src, slice_stride, g, m, c);
      else if (a.outf == FMT_SPLIT) reduce_element<1, MM_F16X3, FMT_SPLIT, FMT_SPLIT>(a, src, slice_stride, g, m, c);
      else reduce_element<1, MM_F16X3, FMT_F32, FMT_SPLIT>(a, src, slice_stride, g, m, c);
    } else {
      if (a.epi == 0) reduce_element<0, MM_F32, FMT_F32, FMT_F32>(a, src, slice_stride, g, m, c);
      else if (a.epi == 1) reduce_element<1, MM_F32, FMT_F32, FMT_F32>(a, src, slice_stride, g, m, c);
      else reduce_element<2, MM_F32, FMT_F32, FMT_F32>(a, src, slice_stride, g, m, c);
    }
  }
}

// =================================================================================================
// small-batch (latency) path: split-K.  At batch 1 the big-tile kernels above would occupy 8-64 of
// the 256 CUs (M = 121..1936 rows), so the K dimension -- cin / 32 chunks x 9 taps K-steps -- is cut into `slices` runs of
// consecutive K-steps (round 4: runs of >= 3 (chunk, tap) steps instead of whole chunks: a K-step of these small grids is DMA-latency
// bound, ~1.3-2.6 us, so the kernel's time is its number of SEQUENTIAL steps; 9 per workgroup before)
// handled by different workgroups (128 px x {128|64} cout, 4 waves, per-tap gather on the
// zero-bordered input, stride 1 or 2, either arithmetic mode).  Each workgroup stores its raw partial accumulators to
// part[slice][group][M][cout]; conv_reduce_kernel sums the slices in a FIXED order (deterministic,
// no atomics) and applies the bias / residual / activation epilogue into the padded output.
// =================================================================================================
template <int CIN, int STRIDE, int CT, int MM>
__global__ __launch_bounds__(256, 2) void conv3x3_splitk_kernel(const ConvArgs a) {
  constexpr int BM = 128, BN = 64 * CT, PT = 2;
  constexpr int NCH = CIN / 32;
  constexpr int WR = BN / 32;  // weight rows per staging thread
  constexpr int BUF = (BM + BN) * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, hh = lane >> 5;

  const int panels = a.groups * a.tiles_n;
  int sl = blockIdx.x % a.slices, rest = blockIdx.x / a.slices;
  if (a.sem) {   // (uniform) fused reduction: every workgroup of tile t -- its slices and its reduce workgroups -- has blockIdx = t (mod 8),
    // i.e. runs on ONE XCD, whose L2 then holds the tile's partial sums; the workgroups behind the conv workgroups reduce
    const int tiles = ((a.M + BM - 1) / BM) * panels, tg = (tiles + 7) / 8;
    const int conv_wgs = tg * a.slices * 8;
    const int b = (int)blockIdx.x;
    if (b >= conv_wgs) {
      const int rb = b - conv_wgs, q = rb >> 3;
      const int t = (q / a.rpt) * 8 + (rb & 7);
      if (t < tiles) splitk_reduce_part<MM, BM, BN>(a, t, q % a.rpt);
      return;
    }
    const int q = b >> 3;
    rest = (q / a.slices) * 8 + (b & 7);
    sl = q % a.slices;
    if (rest >= tiles) return;
  }
  const int p = rest % panels, mt = rest / panels;
  const int g = p / a.tiles_n, nt = p % a.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;
  const float* __restrict__ in = a.in + (size_t)g * a.in_gs;
  const float* __restrict__ wgt = a.w + (size_t)g * a.w_gs + (size_t)n0 * 32;
  const int Wp = a.W + 2, Hp = a.H + 2, HoWo = a.Ho * a.Wo;
  const int mlast = a.M - 1;
  // K-steps (32-channel chunk, tap) are numbered ch * 9 + tap; slice sl takes KT consecutive ones (a.slices divides NCH * 9)
  const int KT = NCH * 9 / a.slices, ks0 = sl * KT;

  const int r0 = tid >> 3;
  const int c4 = (tid & 7) ^ ((r0 >> 1) & 7);
  unsigned pvoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = min(m0 + r0 + 32 * j, mlast);
    const int n = m / HoWo, rem = m - n * HoWo;
    const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
    pvoff[j] = (unsigned)((((n * Hp + STRIDE * ho) * Wp + STRIDE * wo) * a.in_ld + c4 * 4) * 4);
  }
  const unsigned wvoff = (unsigned)((r0 * 32 + c4 * 4) * 4);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

#define ISSUE_TILE(CH, TAP, BUFI)                                                                    \
  {                                                                                                  \
    const int r_ = (TAP) / 3, s_ = (TAP) - r_ * 3;                                                   \
    const float* pb_ = in + (size_t)(r_ * Wp + s_) * a.in_ld + (CH) * 32;                            \
    const unsigned lb_ = lds0 + (unsigned)(((BUFI) * BUF + wid * 256) * 4);                          \
    glds16<0>(pb_, pvoff[0], lb_);                                                                   \
    glds16<0>(pb_, pvoff[1], lb_ + 4096);                                                            \
    glds16<0>(pb_, pvoff[2], lb_ + 8192);                                                            \
    glds16<0>(pb_, pvoff[3], lb_ + 12288);                                                           \
    const float* tb_ = wgt + (size_t)((CH) * 9 + (TAP)) * (a.tiles_n * BN) * 32;                     \
    _Pragma("unroll") for (int j = 0; j < WR; ++j)                                                   \
        glds16<0>(tb_ + j * 1024, wvoff, lb_ + BM * 128 + j * 4096);                                 \
  }

  const int X = (l31 >> 1) & 7;
  const int lo = (hh ^ (X & 1)) * 4, xk = X >> 1;
  const int fo0 = ((0 ^ xk) << 3) + lo, fo1 = ((1 ^ xk) << 3) + lo, fo2 = ((2 ^ xk) << 3) + lo,
            fo3 = ((3 ^ xk) << 3) + lo;

  f32x16 acc[PT][CT];
#pragma unroll
  for (int i = 0; i < PT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int ch = ks0 / 9, tap = ks0 - ch * 9;
  ISSUE_TILE(ch, tap, 0)
  wait_dma_and_barrier();

  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (++tap == 9) { tap = 0; ++ch; }
    if (kt + 1 < KT) ISSUE_TILE(ch, tap, buf ^ 1)
    const float* pP = smem + buf * BUF + (wm * PT * 32 + l31) * 32;
    const float* pW = smem + buf * BUF + (BM + wn * CT * 32 + l31) * 32;
#define FOG(G) ((G) == 0 ? fo0 : (G) == 1 ? fo1 : (G) == 2 ? fo2 : fo3)
#define PXF(G) *reinterpret_cast<const float4*>(pP + i * 1024 + FOG(G))
#define WTF(G) *reinterpret_cast<const float4*>(pW + j * 1024 + FOG(G))
    if (MM == MM_F16X3) {
      SE3TN_MMA_SPLIT(PT, CT, PXF, WTF)
    } else {
      SE3TN_MMA_GROUP(PT, CT, PXF(0), WTF(0))
      SE3TN_MMA_GROUP(PT, CT, PXF(1), WTF(1))
      SE3TN_MMA_GROUP(PT, CT, PXF(2), WTF(2))
      SE3TN_MMA_GROUP(PT, CT, PXF(3), WTF(3))
    }
#undef PXF
#undef WTF
#undef FOG
    if (kt + 1 < KT) wait_dma_and_barrier();
  }
#undef ISSUE_TILE

  // raw partial sums: part[slice][group][m][cout] for conv_reduce_kernel; with the fused reduction, accumulator-fragment order
  // (splitk_reduce_part), written THROUGH the XCD's L2: another XCD's reduce workgroup reads it in this launch
  const int cout = a.tiles_n * BN;
  if (a.sem) {
    const int tiles = ((a.M + BM - 1) / BM) * panels;
    float* __restrict__ ptile = a.part + ((size_t)sl * tiles + rest) * (BM * BN);
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          store_partial(ptile + (size_t)(((((wid * PT + i) * CT + j) * 4 + q) * 64 + lane) * 4),
                        make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]));
  } else {
    float* __restrict__ part = a.part + ((size_t)(sl * a.groups + g) * a.M) * cout;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int m = m0 + (wm * PT + i) * 32 + l31;
      if (m >= a.M) continue;
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = n0 + (wn * CT + j) * 32 + q * 8 + hh * 4;
          *reinterpret_cast<float4*>(part + (size_t)m * cout + c) =
              make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        }
    }
  }
  if (a.sem) {   // publish: the stores have been acknowledged by the device-coherent level, then ONE arrival per workgroup.
    // (A release fence per thread -- __threadfence() -- writes the whole L2 back 256 times per workgroup: 864 us per forward
    // instead of 268, measured.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.sem + rest, 1, __ATOMIC_RELAXED, SE3TN_SEM_SCOPE);
  }
}

template <typename K>
static hipError_t set_lds(K kern, size_t lds, PerDeviceOnce& once) {
  bool* done = once.current();
  if (done && *done) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess && done) *done = true;
  return e;
}

template <int CIN, int WM, int WN, int PT, int CT, int SLABPX, int EPI, int MM = MM_F32, int OUTF = FMT_F32,
          int RESF = FMT_F32>
static hipError_t launch_slab(const ConvArgs& a, hipStream_t st) {
  constexpr int BM = WM * PT * 32, BN = WN * CT * 32;
  constexpr size_t lds = (size_t)(2 * SLABPX * 32 + 2 * BN * 32) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = conv3x3_slab_kernel<CIN, WM, WN, PT, CT, SLABPX, EPI, MM, OUTF, RESF>;
  static PerDeviceOnce attr;
  hipError_t e = set_lds(kern, lds, attr);
  if (e != hipSuccess) return e;
  const int tiles_m = (a.M + BM - 1) / BM;
  // one workgroup per CU (LDS-bound); with more tiles than CUs the workgroups are persistent
  const int ntiles = tiles_m * a.tiles_n * a.groups;
  const int grid = (SE3TN_PERSISTENT_SLAB && ntiles > 256) ? 256 : ntiles;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
  return hipGetLastError();
}

template <int CIN, int EPI, int MM, int OUTF, int NW>
static hipError_t launch_gather_nw(const ConvArgs& a, hipStream_t st) {
  constexpr int BM = NW * 32;
  constexpr size_t lds = (size_t)2 * (BM + 128) * 32 * sizeof(float);
  auto kern = conv3x3_gather_s2_kernel<CIN, EPI, MM, OUTF, NW>;
  static PerDeviceOnce attr;
  hipError_t e = set_lds(kern, lds, attr);
  if (e != hipSuccess) return e;
  const int tiles_m = (a.M + BM - 1) / BM;
#if defined(SE3TN_GATHER_XCD_PIXELS)
  hipLaunchKernelGGL(kern, dim3(((tiles_m + 7) / 8) * 8 * a.tiles_n * a.groups), dim3(NW * 64), lds, st, a);
#else
  hipLaunchKernelGGL(kern, dim3(tiles_m * a.tiles_n * a.groups), dim3(NW * 64), lds, st, a);
#endif
  return hipGetLastError();
}

#ifndef SE3TN_GATHER8_MIN_TILES
#define SE3TN_GATHER8_MIN_TILES 200   // 256-px tiles (8 waves, 1 workgroup per CU) once they fill most of the 256 CUs
#endif
template <int CIN, int EPI, int MM = MM_F32, int OUTF = FMT_F32>
static hipError_t launch_gather(const ConvArgs& a, hipStream_t st) {
  const int tiles256 = ((a.M + 255) / 256) * a.tiles_n * a.groups;
  if (MM == MM_F32 && tiles256 >= SE3TN_GATHER8_MIN_TILES && tiles256 <= 256) return launch_gather_nw<CIN, EPI, MM, OUTF, 8>(a, st);
  return launch_gather_nw<CIN, EPI, MM, OUTF, 4>(a, st);
}

template <int EPI, int MM, int OUTF, int RESF>
static void launch_reduce(const ConvArgs& a, int cout, hipStream_t st) {
  const int total = a.groups * a.M * (cout / 4);
  hipLaunchKernelGGL((conv_reduce_kernel<EPI, MM, OUTF, RESF>), dim3((total + 255) / 256), dim3(256), 0, st, a, cout, total);
}

// the fused reduction's reduce workgroups: at most this many per launch (the launch holds 512 workgroup slots: 2 per compute unit)
#ifndef SE3TN_SPLITK_MAX_REDUCE_WGS
#define SE3TN_SPLITK_MAX_REDUCE_WGS 256
#endif
// outf / resf: FMT_* of the output and residual tensors (f16x3 mode: split rows except the last head conv)
template <int CIN, int STRIDE, int CT, int MM>
static hipError_t launch_splitk(const ConvArgs& a0, int epi, int outf, int resf, hipStream_t st) {
  constexpr int BN = 64 * CT;
  constexpr size_t lds = (size_t)2 * (128 + BN) * 32 * sizeof(float);
  auto kern = conv3x3_splitk_kernel<CIN, STRIDE, CT, MM>;
  static PerDeviceOnce attr;
  hipError_t e = set_lds(kern, lds, attr);
  if (e != hipSuccess) return e;
  const int tiles_m = (a0.M + 127) / 128;
  const int tiles = tiles_m * a0.tiles_n * a0.groups;
  if (a0.sem && tiles <= SE3TN_SPLITK_MAX_TILES && tiles * 4 <= SE3TN_SPLITK_MAX_REDUCE_WGS &&
      (size_t)a0.slices * tiles * 128 * BN * sizeof(float) <= a0.part_bytes) {
    // reduction inside this launch: rpt reduce workgroups per tile, ~128 in all (few tiles: more, thinner row bands)
    ConvArgs a = a0;
    int rpt = 4;
    while (rpt < 32 && tiles * rpt * 2 <= SE3TN_SPLITK_MAX_REDUCE_WGS / 2) rpt *= 2;
    a.rpt = rpt; a.epi = epi; a.outf = outf; a.resf = resf;
    const int tg = (tiles + 7) / 8;
    hipLaunchKernelGGL(kern, dim3(tg * 8 * (a.slices + rpt)), dim3(256), lds, st, a);
    return hipGetLastError();
  }
  ConvArgs a = a0;
  a.sem = nullptr;
  hipLaunchKernelGGL(kern, dim3(tiles * a.slices), dim3(256), lds, st, a);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int cout = a.tiles_n * BN;
  if (MM == MM_F32) {
    if (epi == 0) launch_reduce<0, MM_F32, FMT_F32, FMT_F32>(a, cout, st);
    else if (epi == 1) launch_reduce<1, MM_F32, FMT_F32, FMT_F32>(a, cout, st);
    else launch_reduce<2, MM_F32, FMT_F32, FMT_F32>(a, cout, st);
  } else {
    if (epi == 0) launch_reduce<0, MM_F16X3, FMT_SPLIT, FMT_F32>(a, cout, st);
    else if (epi == 2) launch_reduce<2, MM_F16X3, FMT_SPLIT, FMT_F32>(a, cout, st);
    else if (outf == FMT_SPLIT) launch_reduce<1, MM_F16X3, FMT_SPLIT, FMT_SPLIT>(a, cout, st);
    else launch_reduce<1, MM_F16X3, FMT_F32, FMT_SPLIT>(a, cout, st);
  }
  (void)resf;
  return hipGetLastError();
}

// Split-K is used when the big-tile grid would leave most of the 256 CUs idle and the partial-sum workspace is large enough.
// slices = the largest divisor of the K-step count (cin / 32 x 9) that leaves every workgroup >= SE3TN_SPLITK_MIN_STEPS K-steps and
// the launch <= 512 workgroups (two per CU).  Up to SE3TN_SPLITK_FIXED_MAX_N images the workgroup count is taken from the ONE-image
// geometry -- 6 / 12 / 24 / 24 / 48 slices of 3 steps for the 64-ch / convAB1 / convAB2 / trans|rot conv1 / conv2 layers -- so the
// slice count is a function of the layer (the same summation order for one pair alone or two together); larger batches count the
// real grid (fewer, longer slices), so from 3 pairs on a pair's last bits may depend on the batch it travels in -- as they do across
// the algorithm switches at 6 / 14 pairs, and in the reference under cuDNN.
#ifndef SE3TN_SPLITK_MIN_STEPS
#define SE3TN_SPLITK_MIN_STEPS 3   // measured at batch 1: 2 -> 280 us, 3 -> 268 us, 4 -> 276 us per forward (whole chunks before: 321 us)
#endif
#ifndef SE3TN_SPLITK_MAX_WGS
#define SE3TN_SPLITK_MAX_WGS 512
#endif
#ifndef SE3TN_SPLITK_FIXED_MAX_N
#define SE3TN_SPLITK_FIXED_MAX_N 2  // (5 = up to the Winograd threshold was measured: 15 % slower at n = 4 -- 4 x the partial sums of 48 slices)
#endif
static int pick_slices(const ConvArgs& a, int cin, int cout, int big_tile_rows, int bn_big) {
  const int big_blocks = ((a.M + big_tile_rows - 1) / big_tile_rows) * (cout / bn_big) * a.groups;
  if (big_blocks >= 200 || a.part == nullptr) return 0;
  const int bn = cout >= 128 ? 128 : 64;
  const int hw = a.Ho * a.Wo;
  const int rows = a.M <= SE3TN_SPLITK_FIXED_MAX_N * hw ? hw : a.M;          // one image's rows | the real grid
  const int base = ((rows + 127) / 128) * (cout / bn) * a.groups;            // workgroups per slice
  const int ks = cin / 32 * 9;
  const size_t per_slice = (size_t)a.groups * a.M * cout * sizeof(float);
  int best = 0;
  for (int sl = 1; sl <= ks; ++sl) {
    if (ks % sl != 0 || ks / sl < SE3TN_SPLITK_MIN_STEPS) continue;
    if (per_slice * sl > a.part_bytes) break;
    if (best > 0 && base * sl > SE3TN_SPLITK_MAX_WGS) break;
    best = sl;
  }
  return best;
}

// Slab sizes (pixels, multiple of 8) = worst case of
//   255 + 2 per row break + (2 Wp + 2) per image break + 2 (Wp + 1) + 1     (tests/test_slab_geometry.py)
//   W = 44: 454 -> 464     W = 22: 378 -> 384     W = 11: 410 -> 424
#ifndef SE3TN_CONV64_SMALL_MAX_N
#define SE3TN_CONV64_SMALL_MAX_N 5   // the trunk convs of up to this many pairs take conv64_small_kernel (0: never).  Measured 2 vs 5:
                                     // 8.6k -> 9.6k / 10.2k -> 11.0k / 10.8k -> 11.2k pairs/s at 3 / 4 / 5 pairs (one stream)
#endif
// (SE3TN_SLICES_SMALL_MAX_N: se3tn_internal.h)
hipError_t launch_conv3x3(const ConvArgs& a0, int cin, int cout, int stride, int epi, hipStream_t st) {
  ConvArgs a = a0;
  if (!a.fast && cin == 64 && cout == 64 && stride == 1 && a.W == 44 && a.H == 44 && epi != 2 && a.M % (44 * 44) == 0 &&
      a.M / (44 * 44) <= SE3TN_CONV64_SMALL_MAX_N && a.small_ok)
    return launch_conv64_small(a, a.M / (44 * 44), epi, st);
  // batch 1-5, float32: the wide convs as one round of 128-pixel x 32-cout x channel-slice workgroups (conv_slices_small.hip)
  if (!a.fast && a.small_ok && a.part && a.Ho * a.Wo > 0 && a.M % (a.Ho * a.Wo) == 0 && a.M / (a.Ho * a.Wo) <= SE3TN_SLICES_SMALL_MAX_N) {
    const int sl = conv_slices_small_count(cin, stride, a.H);
    if (sl > 0 && (size_t)sl * a.groups * a.M * cout * sizeof(float) <= a.part_bytes) {
      a.slices = sl;
      a.tiles_n = cout / 32;
      a.sem = nullptr;
      hipError_t e = launch_conv_slices_small(a, cin, stride, st);
      if (e != hipSuccess) return e;
      if (a.skip_reduce) return hipSuccess;              // (the tail adds the slices: kernels_misc.hip tail_parts_kernel)
      if (epi == 0) launch_reduce<0, MM_F32, FMT_F32, FMT_F32>(a, cout, st);
      else if (epi == 1) launch_reduce<1, MM_F32, FMT_F32, FMT_F32>(a, cout, st);
      else launch_reduce<2, MM_F32, FMT_F32, FMT_F32>(a, cout, st);
      return hipGetLastError();
    }
  }
  a.slices = pick_slices(a, cin, cout, stride == 1 ? 256 : 128, cout >= 128 ? 128 : 64);
  if (a.slices > 0) {
    a.tiles_n = cout >= 128 ? cout / 128 : 1;
    if (a.fast) {  // f16x3: split rows everywhere, float32 out of the last head conv (cin 512, residual)
      const int outf = (cin == 512 && epi == 1) ? FMT_F32 : FMT_SPLIT;
      if (cin == 64 && stride == 1) return launch_splitk<64, 1, 1, MM_F16X3>(a, epi, outf, FMT_SPLIT, st);
      if (cin == 128 && stride == 2) return launch_splitk<128, 2, 2, MM_F16X3>(a, epi, outf, FMT_SPLIT, st);
      if (cin == 256 && stride == 1) return launch_splitk<256, 1, 2, MM_F16X3>(a, epi, outf, FMT_SPLIT, st);
      if (cin == 256 && stride == 2) return launch_splitk<256, 2, 2, MM_F16X3>(a, epi, outf, FMT_SPLIT, st);
      if (cin == 512 && stride == 1) return launch_splitk<512, 1, 2, MM_F16X3>(a, epi, outf, FMT_SPLIT, st);
      return hipErrorInvalidValue;
    }
    if (cin == 64 && stride == 1) return launch_splitk<64, 1, 1, MM_F32>(a, epi, FMT_F32, FMT_F32, st);
    if (cin == 128 && stride == 2) return launch_splitk<128, 2, 2, MM_F32>(a, epi, FMT_F32, FMT_F32, st);
    if (cin == 256 && stride == 1) return launch_splitk<256, 1, 2, MM_F32>(a, epi, FMT_F32, FMT_F32, st);
    if (cin == 256 && stride == 2) return launch_splitk<256, 2, 2, MM_F32>(a, epi, FMT_F32, FMT_F32, st);
    if (cin == 512 && stride == 1) return launch_splitk<512, 1, 2, MM_F32>(a, epi, FMT_F32, FMT_F32, st);
    return hipErrorInvalidValue;
  }
  if (a.fast) {
    // f16x3 mode (se3tn_set_precision): every 3x3 conv runs on the f16 matrix cores with split-row
    // operands; the max-pool produces the first split-row tensor, the last conv of the heads writes
    // float32 for the tail.
    if (stride == 1 && cin == 64 && cout == 64 && a.W == 44) {
      a.tiles_n = 1;
      if (epi == 0) return launch_slab<64, 8, 1, 1, 2, 464, 0, MM_F16X3, FMT_SPLIT, FMT_F32>(a, st);
      if (epi == 1) return launch_slab<64, 8, 1, 1, 2, 464, 1, MM_F16X3, FMT_SPLIT, FMT_SPLIT>(a, st);
    }
    a.tiles_n = cout / 128;
    if (stride == 2 && cin == 128 && epi == 2) return launch_gather<128, 2, MM_F16X3, FMT_SPLIT>(a, st);
    if (stride == 2 && cin == 256 && epi == 2) return launch_gather<256, 2, MM_F16X3, FMT_SPLIT>(a, st);
    if (stride == 1 && cin == 256 && a.W == 22 && epi == 0)
      return launch_slab<256, 4, 2, 2, 2, 384, 0, MM_F16X3, FMT_SPLIT, FMT_F32>(a, st);
    if (stride == 1 && cin == 256 && a.W == 22 && epi == 1)
      return launch_slab<256, 4, 2, 2, 2, 384, 1, MM_F16X3, FMT_SPLIT, FMT_SPLIT>(a, st);
    if (stride == 1 && cin == 512 && a.W == 11 && epi == 0)
      return launch_slab<512, 4, 2, 2, 2, 424, 0, MM_F16X3, FMT_SPLIT, FMT_F32>(a, st);
    if (stride == 1 && cin == 512 && a.W == 11 && epi == 1)
      return launch_slab<512, 4, 2, 2, 2, 424, 1, MM_F16X3, FMT_F32, FMT_SPLIT>(a, st);
    return hipErrorInvalidValue;
  }
  if (stride == 1 && cin == 64 && cout == 64 && a.W == 44) {
    a.tiles_n = 1;
    if (epi == 0) return launch_slab<64, 8, 1, 1, 2, 464, 0>(a, st);
    if (epi == 1) return launch_slab<64, 8, 1, 1, 2, 464, 1>(a, st);
  }
  a.tiles_n = cout / 128;
  if (stride == 1 && cin == 256 && a.W == 22) {
    if (epi == 0) return launch_slab<256, 4, 2, 2, 2, 384, 0>(a, st);
    if (epi == 1) return launch_slab<256, 4, 2, 2, 2, 384, 1>(a, st);
  }
  if (stride == 1 && cin == 512 && a.W == 11) {
    if (epi == 0) return launch_slab<512, 4, 2, 2, 2, 424, 0>(a, st);
    if (epi == 1) return launch_slab<512, 4, 2, 2, 2, 424, 1>(a, st);
  }
  if (stride == 2 && cin == 128 && epi == 2) return launch_gather<128, 2>(a, st);
  if (stride == 2 && cin == 256 && epi == 2) return launch_gather<256, 2>(a, st);
  return hipErrorInvalidValue;
}

}  // namespace se3tn
