// The 128 .. 512-channel 3x3 convs (convAB1, convAB2, trans|rot conv1 / conv2: se3_tracknet.py:74-97, network_modules.py:86-120) at
// batch 1-5: the regime Tracker.on_track runs in.  At one pair they were 136 of the forward's 230 us (EXPERIMENTS items 45, 50), and what
// bounds them is not bytes but the float32 matrix rate: 2.6 GFLOP / 157 TFLOP/s = 17 us on 256 compute units -- if every one of them
// computes.  conv3x3_splitk_kernel's 128 x 128 tiles leave 8-32 output tiles per layer, so it cuts K into 12-48 runs of THREE K-steps:
// each workgroup starts, waits for its first bytes, computes for a microsecond and stores a 64 KB partial tile (24 MB of partial sums
// per trans|rot conv), and the grid is 1.5 waves of workgroups.
//
// Here a workgroup owns 128 output pixels of ONE image x 32 couts x ONE slice of the input channels with all nine taps:
//   layer (cin, stride, out)       slices x channels   n-tiles   m-tiles/image   workgroups/pair   LDS
//   convAB1      128, s2, 22x22        4 x 32              8          4               128          127 KB
//   convAB2      256, s1, 22x22        8 x 32              8          4               256           66 KB
//   trans|rot 1  256, s2, 11x11        8 x 32             32          1               256          111 KB
//   trans|rot 2  512, s1, 11x11        8 x 64          2 x 16         1               256          123 KB
// one round of workgroups per pair, 9 or 18 K-steps each, 4-8 partial sums per output instead of 12-48 (conv_reduce_kernel of
// conv3x3_mfma.hip adds them in slice order and applies bias / residual / activation).  The slice count is a function of the layer
// alone: a pair's bits do not depend on the batch it travels in.
//   * taps share the input: the rows of the (zero-bordered) input under the tile are fetched ONCE as they lie -- a contiguous run of
//     padded rows, 128 bytes (32 channels) per pixel -- and every tap reads them at its own offset; the weights are the packed
//     panels as they are ([chunk][tap][cout][32]: a K-step's 32 x 32 tile is 4 KB contiguous);
//   * LDS-DMA (patch, then the nine weight tiles in tap order, per 32-channel chunk): the first patch and four tiles are requested
//     before the first K-step, two more instructions from between the MFMAs of every K-step; a K-step starts after `s_waitcnt
//     vmcnt(what was requested after its tile)` + barrier, and its operands are read one K-step ahead of their MFMAs;
//   * v_mfma_f32_32x32x2_f32, A operand = weights, B operand = pixels: wave w = pixels 32 w .. 32 w + 31 of the tile x the 32 couts;
//     a 16-byte LDS read feeds four MFMAs (lane half h holds channels 8 g + 4 h .. + 3 of an 8-channel group, MFMA e takes element e
//     of both operands); two accumulators (even / odd groups) are added at the end;
//   * LDS images XOR-swizzled like the other kernels' (16-byte column ^ ((row >> 1) & 7), applied on the DMA source address).
// Float32 only (f16x3 keeps conv3x3_splitk_kernel).
#include "mfma_common.h"

namespace se3tn {

// -DSE3TN_SMALL_TRACE (developer build, scripts/small_trace.py): thread 0 of every workgroup stamps wall_clock64() (100 MHz) at the
// kernel's phase boundaries; the last launch's stamps are read back with se3tn_debug_trace_slices()
#if defined(SE3TN_SMALL_TRACE)
static __device__ unsigned long long g_cs_trace[2][1024][8];   // [0]: wall_clock64() (100 MHz), [1]: clock64() (s_memtime: core clock)
#define CS_TRACE(P)                                                                                      \
  {                                                                                                      \
    const unsigned lb_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                 \
    if (threadIdx.x == 0 && lb_ < 1024) {                                                                \
      g_cs_trace[0][lb_][P] = wall_clock64();                                                            \
      g_cs_trace[1][lb_][P] = clock64();                                                                 \
    }                                                                                                    \
  }
#else
#define CS_TRACE(P)
#endif

template <int N>
__device__ __forceinline__ void cs_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int CHUNKS, int PPMAX>
struct CsGeom {
  static constexpr int NB = (PPMAX * 8 + 255) / 256;          // patch DMA instructions per thread and chunk
  static constexpr int PER_CHUNK = NB + 9;                    // + one per weight tile
  static constexpr int TOTAL = CHUNKS * PER_CHUNK;
  static constexpr int NSTEPS = CHUNKS * 9;
  // DMA schedule (instruction index d: chunk d / PER_CHUNK; inside a chunk NB patch instructions, then the nine weight tiles):
  // AHEAD = the first chunk's patch + its first four tiles are requested before the first K-step, every K-step requests RATE more
  // from between its MFMAs.  (Issuing an LDS-DMA instruction blocks the wave ~35 cycles while the four waves share the address
  // pipe: with all 30 up front the matrix pipe idled for the first 1.9 us -- scripts/small_trace.py, EXPERIMENTS item 51.)
  static constexpr int AHEAD = NB + 4, RATE = 2;
  static constexpr int issued_before(int ks) { return AHEAD + RATE * ks < TOTAL ? AHEAD + RATE * ks : TOTAL; }   // ... K-step ks's own requests
  static constexpr int tile_index(int ks) { return (ks / 9) * PER_CHUNK + NB + ks % 9; }
  // DMA instructions requested AFTER K-step s's weight tile when s synchronises (in the body of K-step s - 1, before its requests)
  static constexpr int outstanding(int s) { return (s == 0 ? AHEAD : issued_before(s - 1)) - (tile_index(s) + 1); }
  static constexpr bool schedule_ok() {
    for (int s = 0; s < NSTEPS; ++s)
      if (outstanding(s) < 0) return false;
    return issued_before(NSTEPS - 1) == TOTAL;
  }
  static constexpr int PATCH_FLOATS = NB * 256 * 4;
  static constexpr int CHUNK_FLOATS = PATCH_FLOATS + 9 * 1024;
  static constexpr size_t LDS = (size_t)CHUNKS * CHUNK_FLOATS * sizeof(float);
};

struct CsLane {
  const float* smem;
  int row, kh;      // lane & 31: cout row of the A fragment = pixel column of the B fragment; lane >> 5
  int ppb, Wp;      // patch pixel under tap (0, 0) of this lane's output pixel; padded input width
};

// this thread's part of the DMA stream
struct CsDma {
  const float* in;     // patch origin, channel offset of the slice applied
  const float* wgt;    // panel rows of this workgroup, first chunk of the slice
  unsigned lds0, wvoff;
  int tid, wid, PP, in_ld, cout;
};
template <class G, int D>
__device__ __forceinline__ void cs_request(const CsDma& x) {
  if constexpr (D < G::TOTAL) {
    constexpr int c = D / G::PER_CHUNK, r = D % G::PER_CHUNK;
    if constexpr (r < G::NB) {
      // patch: slot = 16-byte column (slot & 7) of patch pixel (slot >> 3); slots past the patch re-read its last pixel
      const int slot = r * 256 + x.tid;
      const int p = min(slot >> 3, x.PP - 1);
      const int col = (slot & 7) ^ ((p >> 1) & 7);
      glds16<0>(x.in + c * 32, (unsigned)((p * x.in_ld + col * 4) * 4), x.lds0 + (unsigned)((c * G::CHUNK_FLOATS + (r * 256 + x.wid * 64) * 4) * 4));
    } else {
      // weight tile: row tid >> 3, column (tid & 7) ^ ((row >> 1) & 7)
      constexpr int tap = r - G::NB;
      glds16<0>(x.wgt + ((size_t)c * 9 + tap) * x.cout * 32, x.wvoff,
                x.lds0 + (unsigned)((c * G::CHUNK_FLOATS + G::PATCH_FLOATS + tap * 1024 + x.wid * 256) * 4));
    }
  }
}
template <class G, int D0, int D1>
struct CsRequestRange {      // requests D0 .. D1 - 1
  static __device__ __forceinline__ void go(const CsDma& x) {
    if constexpr (D0 < D1) {
      cs_request<G, D0>(x);
      CsRequestRange<G, D0 + 1, D1>::go(x);
    }
  }
};

struct CsFrag {
  float4 x[4], w[4];   // B (pixels) and A (weights) fragments of one K-step: four 8-channel groups
};

template <class G, int KS>
__device__ __forceinline__ void cs_sync() {
  cs_wait_vm<G::outstanding(KS)>();     // everything up to this K-step's weight tile has landed (this wave's part) ...
  __syncthreads();                      // ... and the other waves'
}

template <class G, int KS>
__device__ __forceinline__ void cs_load(const CsLane& f, CsFrag& fr) {
  constexpr int c = KS / 9, tap = KS % 9, r = tap / 3, s = tap % 3;
  const int pp = f.ppb + r * f.Wp + s;
  const float* px = f.smem + c * G::CHUNK_FLOATS + pp * 32;
  const float* wt = f.smem + c * G::CHUNK_FLOATS + G::PATCH_FLOATS + tap * 1024 + f.row * 32;
  const int swp = (pp >> 1) & 7, sww = (f.row >> 1) & 7;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    fr.x[g] = *reinterpret_cast<const float4*>(px + (((2 * g + f.kh) ^ swp) << 2));
    fr.w[g] = *reinterpret_cast<const float4*>(wt + (((2 * g + f.kh) ^ sww) << 2));
  }
}

template <int G0>
__device__ __forceinline__ void cs_mma(const CsFrag& fr, f32x16& acc0, f32x16& acc1) {   // groups G0, G0 + 1 of the K-step: 8 MFMAs
  constexpr int g = G0;
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fr.w[g].x, fr.x[g].x, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fr.w[g + 1].x, fr.x[g + 1].x, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fr.w[g].y, fr.x[g].y, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fr.w[g + 1].y, fr.x[g + 1].y, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fr.w[g].z, fr.x[g].z, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fr.w[g + 1].z, fr.x[g + 1].z, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fr.w[g].w, fr.x[g].w, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fr.w[g + 1].w, fr.x[g + 1].w, acc1, 0, 0, 0);
}

// K-step KS: wait + barrier for K-step KS + 1's weight tile, its fragments read, THEN the 16 MFMAs of KS with this K-step's DMA
// requests in their middle -- a K-step of matrix work (1,024 cycles) between an LDS read and its use.  sched_barrier pins that
// order: left to itself the scheduler sinks the reads to one MFMA before their use (64 cycles, less than an LDS round trip)
template <class G, int KS>
struct CsRun {
  static __device__ __forceinline__ void go(const CsLane& f, const CsDma& x, CsFrag& fr0, CsFrag& fr1, f32x16& a0, f32x16& a1) {
    CsRun<G, KS - 1>::go(f, x, fr0, fr1, a0, a1);
    if constexpr (KS + 1 < G::NSTEPS) {
      cs_sync<G, KS + 1>();
      cs_load<G, KS + 1>(f, (KS & 1) ? fr0 : fr1);
    }
    __builtin_amdgcn_sched_barrier(0);
    cs_mma<0>((KS & 1) ? fr1 : fr0, a0, a1);
    CsRequestRange<G, G::issued_before(KS), G::issued_before(KS + 1)>::go(x);
    cs_mma<2>((KS & 1) ? fr1 : fr0, a0, a1);
    __builtin_amdgcn_sched_barrier(0);
    if (KS == 3) { CS_TRACE(3) }
    if (KS == 8) { CS_TRACE(4) }
  }
};
template <class G>
struct CsRun<G, -1> {
  static __device__ __forceinline__ void go(const CsLane& f, const CsDma&, CsFrag& fr0, CsFrag&, f32x16&, f32x16&) {
    cs_sync<G, 0>();
    CS_TRACE(2)
    cs_load<G, 0>(f, fr0);
  }
};

// grid: (slices x n-tiles, m-tiles x images, groups) workgroups of 256 threads; the slice is the fastest index: consecutive
// workgroups go to consecutive XCDs, so the workgroups that read the same input channels share an L2.  Everything about the
// layer's geometry is a template constant: the index arithmetic of a kernel whose whole life is 12 us must not contain a division by
// a run-time value (the first version spent 2.6 us between entry and its last DMA issue: EXPERIMENTS item 51).
// The partial sums go to conv_reduce_kernel (next launch).  Measured alternative (EXPERIMENTS item 51): the reduction in THIS launch
// -- partial tiles written through to memory with sc1 stores, a ticket per output tile, the last arriver reads the SLICES tiles back
// with sc1 loads, adds them in slice order and applies the epilogue: 210 us per forward against 183 us with the separate launch.
// Three dependent round trips to the memory side (store acknowledgements, ticket, read-back) cost more than a launch boundary.
template <int CHUNKS, int STRIDE, int PPMAX, int WO, int SLICES, int NT>
__global__ __launch_bounds__(256, 1) void conv_slices_small_kernel(const ConvArgs a) {
  using G = CsGeom<CHUNKS, PPMAX>;
  constexpr int HW = WO * WO, TPI = (HW + 127) / 128, H = STRIDE * WO, Wp = H + 2, cout = NT * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  CS_TRACE(0)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sl = blockIdx.x % SLICES, nt = blockIdx.x / SLICES;
  const int img = blockIdx.y / TPI, t = blockIdx.y % TPI, g = blockIdx.z;
  const int m0 = t * 128, m1 = min(m0 + 127, HW - 1);
  const int oy0 = m0 / WO, oy1 = m1 / WO;
  const int PP = ((oy1 - oy0) * STRIDE + 3) * Wp;           // pixels of the patch: whole padded rows oy0 * STRIDE .. oy1 * STRIDE + 2
  const size_t pix0 = ((size_t)img * (H + 2) + (size_t)oy0 * STRIDE) * Wp;
  const float* __restrict__ in = a.in + (size_t)g * a.in_gs + pix0 * a.in_ld + (size_t)sl * CHUNKS * 32;
  // panels [chunk][tap][cout][32] of group g; this workgroup's rows nt * 32 .. + 31 of chunks sl * CHUNKS ..
  const float* __restrict__ wgt = a.w + (size_t)g * a.w_gs + ((size_t)sl * CHUNKS * 9 * cout + (size_t)nt * 32) * 32;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

  static_assert(G::schedule_ok(), "a K-step would wait for a weight tile that has not been requested");
  CsDma dma;
  dma.in = in; dma.wgt = wgt; dma.lds0 = lds0;
  dma.tid = tid; dma.wid = wid; dma.PP = PP; dma.in_ld = a.in_ld; dma.cout = cout;
  {
    const int r0 = tid >> 3;
    dma.wvoff = (unsigned)((r0 * 32 + (((tid & 7) ^ ((r0 >> 1) & 7)) << 2)) * 4);
  }
  CsRequestRange<G, 0, G::AHEAD>::go(dma);

  CS_TRACE(1)
  CsLane f;
  f.smem = smem;
  f.row = lane & 31;
  f.kh = lane >> 5;
  f.Wp = Wp;
  const int m = min(m0 + wid * 32 + f.row, HW - 1);         // (rows past the image compute its last pixel again and are not stored)
  const int oy = m / WO, ox = m - oy * WO;
  f.ppb = (oy - oy0) * STRIDE * Wp + ox * STRIDE;
  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
  CsFrag fr0, fr1;
  CsRun<G, CHUNKS * 9 - 1>::go(f, dma, fr0, fr1, acc0, acc1);

#if defined(SE3TN_SMALL_TRACE)
  if (acc0[0] + acc1[0] == 1.2345e-30f) return;   // (the accumulators are final before the stamp)
#endif
  CS_TRACE(5)
  // partial sums: part[slice][group][m][cout] (conv_reduce_kernel's layout); this lane = pixel (lane & 31), couts 8 q + 4 kh .. + 3
  const bool valid = m0 + wid * 32 + f.row < HW;
  const size_t slice_stride = (size_t)a.groups * a.M * cout;
  float* __restrict__ part0 = a.part + (((size_t)g * a.M + (size_t)img * HW + m) * cout + nt * 32 + f.kh * 4);
  float4 mine[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    mine[q] = make_float4(acc0[4 * q + 0] + acc1[4 * q + 0], acc0[4 * q + 1] + acc1[4 * q + 1], acc0[4 * q + 2] + acc1[4 * q + 2],
                          acc0[4 * q + 3] + acc1[4 * q + 3]);
  if (valid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(part0 + sl * slice_stride + q * 8) = mine[q];
  }
#if defined(SE3TN_SMALL_TRACE)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  CS_TRACE(6)
}

#if defined(SE3TN_SMALL_TRACE)
extern "C" int se3tn_debug_trace_slices(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cs_trace), sizeof(unsigned long long) * 2 * 1024 * 8);
}
#endif

template <int CHUNKS, int STRIDE, int PPMAX, int WO, int SLICES, int NT>
static hipError_t launch_cs(const ConvArgs& a, hipStream_t st) {
  using G = CsGeom<CHUNKS, PPMAX>;
  static_assert(G::LDS <= 160 * 1024 && G::TOTAL <= 63, "LDS image / DMA count out of range");
  constexpr int HW = WO * WO, TPI = (HW + 127) / 128;
  if (a.Ho != WO || a.Wo != WO || a.H != STRIDE * WO || a.W != STRIDE * WO || a.tiles_n != NT || a.slices != SLICES || a.M % HW != 0)
    return hipErrorInvalidValue;
  static PerDeviceOnce attr;
  auto kern = conv_slices_small_kernel<CHUNKS, STRIDE, PPMAX, WO, SLICES, NT>;
  bool* done = attr.current();
  if (!(done && *done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    if (e != hipSuccess) return e;
    if (done) *done = true;
  }
  hipLaunchKernelGGL(kern, dim3(SLICES * NT, TPI * (a.M / HW), a.groups), dim3(256), G::LDS, st, a);
  return hipGetLastError();
}

// slices of the layer (0: this geometry has no kernel here)
int conv_slices_small_count(int cin, int stride, int H) {
  if (cin == 128 && stride == 2 && H == S2) return 4;
  if (cin == 256 && stride == 1 && H == S3) return 8;
  if (cin == 256 && stride == 2 && H == S3) return 8;
  if (cin == 512 && stride == 1 && H == S4) return 8;
  return 0;
}

// the conv part only: a.slices / a.tiles_n (= cout / 32) are set by the caller, who also launches the reduction
hipError_t launch_conv_slices_small(const ConvArgs& a, int cin, int stride, hipStream_t st) {
  if (cin == 128 && stride == 2) return launch_cs<1, 2, 690, S3, 4, 8>(a, st);     // convAB1: 128 -> 256
  if (cin == 256 && stride == 1) return launch_cs<1, 1, 216, S3, 8, 8>(a, st);     // convAB2: 256 -> 256
  if (cin == 256 && stride == 2) return launch_cs<1, 2, 552, S4, 8, 32>(a, st);    // trans|rot conv1: 256 -> 1024
  if (cin == 512 && stride == 1) return launch_cs<2, 1, 169, S4, 8, 16>(a, st);    // trans|rot conv2: 2 x (512 -> 512)
  return hipErrorInvalidValue;
}

}  // namespace se3tn
