// Host side of `model.load_state_dict(checkpoint['state_dict'])` (predict.py:151-156): take the
// reference's state_dict tensors as they are, fold every eval-mode BatchNorm2d into the convolution
// in front of it (float64), and emit the packed device blob described in se3tn_internal.h.
//
//   BN(eval): y = (x - mean) / sqrt(var + 1e-5) * gamma + beta   (network_modules.py:64,96;
//   torch default eps)   =>   w' = w * s,  b' = (b - mean) * s + beta,  s = gamma / sqrt(var+eps)
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "se3tn_internal.h"
#include "weights.h"

namespace se3tn {

// The float32 part of Se3TrackNet.state_dict(): 106 tensors (se3_tracknet.py:57-78).
const std::vector<ExpectedTensor>& expected_tensors() {
  static std::vector<ExpectedTensor> v;
  if (!v.empty()) return v;
  auto bn = [&](const std::string& p, int64_t c) {
    for (const char* s : {".weight", ".bias", ".running_mean", ".running_var"}) v.push_back({p + s, {c}});
  };
  struct CB { const char* n; int64_t cin, cout, k; };
  for (const CB& c : {CB{"convA1", 4, 64, 7}, CB{"convB1", 4, 64, 7}, CB{"convAB1", 128, 256, 3},
                      CB{"trans_conv1", 256, 512, 3}, CB{"rot_conv1", 256, 512, 3}}) {
    v.push_back({std::string(c.n) + ".0.weight", {c.cout, c.cin, c.k, c.k}});
    v.push_back({std::string(c.n) + ".0.bias", {c.cout}});
    bn(std::string(c.n) + ".1", c.cout);
  }
  struct BB { const char* n; int64_t c; };
  for (const BB& b : {BB{"convA2", 64}, BB{"convB2", 64}, BB{"convB3", 64}, BB{"convAB2", 256},
                      BB{"trans_conv2", 512}, BB{"rot_conv2", 512}}) {
    for (int i = 1; i <= 2; ++i) {
      v.push_back({std::string(b.n) + ".conv" + std::to_string(i) + ".weight", {b.c, b.c, 3, 3}});
      v.push_back({std::string(b.n) + ".conv" + std::to_string(i) + ".bias", {b.c}});
      bn(std::string(b.n) + ".bn" + std::to_string(i), b.c);
    }
  }
  for (const char* h : {"trans_out", "rot_out"}) {
    v.push_back({std::string(h) + ".0.weight", {3, 512}});
    v.push_back({std::string(h) + ".0.bias", {3}});
  }
  return v;
}

namespace {

struct Folded {
  std::vector<float> w;  // OIHW
  std::vector<float> b;
  int cout, cin, k;
};

Folded fold(const TensorMap& t, const std::string& conv, const std::string& bn) {
  const StoredTensor& W = t.at(conv + ".weight");
  const StoredTensor& B = t.at(conv + ".bias");
  const StoredTensor& g = t.at(bn + ".weight");
  const StoredTensor& be = t.at(bn + ".bias");
  const StoredTensor& mu = t.at(bn + ".running_mean");
  const StoredTensor& var = t.at(bn + ".running_var");
  Folded f;
  f.cout = (int)W.shape[0]; f.cin = (int)W.shape[1]; f.k = (int)W.shape[2];
  const size_t per = (size_t)f.cin * f.k * f.k;
  f.w.resize(W.data.size());
  f.b.resize(f.cout);
  for (int o = 0; o < f.cout; ++o) {
    const double s = (double)g.data[o] / std::sqrt((double)var.data[o] + 1e-5);
    for (size_t i = 0; i < per; ++i) f.w[o * per + i] = (float)((double)W.data[o * per + i] * s);
    f.b[o] = (float)(((double)B.data[o] - (double)mu.data[o]) * s + (double)be.data[o]);
  }
  return f;
}

// OIHW 3x3 -> [chunk][tap][cout_total][32] at column offset o_off
void pack3(const Folded& f, float* dst, int cout_total, int o_off) {
  const int nch = f.cin / 32;
  for (int ch = 0; ch < nch; ++ch)
    for (int tap = 0; tap < 9; ++tap)
      for (int o = 0; o < f.cout; ++o)
        for (int ci = 0; ci < 32; ++ci)
          dst[(((size_t)ch * 9 + tap) * cout_total + (o + o_off)) * 32 + ci] =
              f.w[(((size_t)o * f.cin) + ch * 32 + ci) * 9 + tap];
}

// float32 -> IEEE binary16 bits, round to nearest even (subnormals kept, overflow -> inf)
uint16_t f32_to_f16_bits(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));  // inf / nan
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                     // rounds to inf
  if (x < 0x38800000u) {                                                                        // subnormal / zero
    if (x < 0x33000000u) return (uint16_t)sign;                                                 // < 2^-25
    const int shift = 126 - (int)(x >> 23);  // 14..24: value = mant * 2^-(shift + 10)
    const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
    uint32_t h = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1))) ++h;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((x >> 23) - 112) << 10 | ((x >> 13) & 0x3ffu);
  const uint32_t rem = x & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
  return (uint16_t)(sign | h);
}
float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) { x = sign; }
    else { float v = std::ldexp((float)m, -24); std::memcpy(&x, &v, 4); x |= sign; }
  } else if (e == 31) { x = sign | 0x7f800000u | (m << 13); }
  else { x = sign | ((e + 112) << 23) | (m << 13); }
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

// OIHW 7x7, Cin=4 -> [64][204]: k = pair*8 + half*4 + c, the (tap of half 0 | tap of half 1) pairs in
// the order stem7x7_mfma.hip walks them: 21 in-row pairs (r,2j)|(r,2j+1), 3 column-6 pairs
// (2j,6)|(2j+1,6), then (6,6)|zero.  k 200..203 pad the row to 816 bytes.
void pack_stem(const Folded& f, float* dst) {
  std::memset(dst, 0, sizeof(float) * 64 * 204);
  int taps[25][2][2];  // [pair][half] -> (r, s); r = -1: zero weights
  int np = 0;
  for (int r = 0; r < 7; ++r)
    for (int sp = 0; sp < 3; ++sp, ++np) {
      taps[np][0][0] = r; taps[np][0][1] = 2 * sp;
      taps[np][1][0] = r; taps[np][1][1] = 2 * sp + 1;
    }
  for (int j = 0; j < 3; ++j, ++np) {
    taps[np][0][0] = 2 * j;     taps[np][0][1] = 6;
    taps[np][1][0] = 2 * j + 1; taps[np][1][1] = 6;
  }
  taps[np][0][0] = 6; taps[np][0][1] = 6; taps[np][1][0] = -1; taps[np][1][1] = 0;
  for (int o = 0; o < 64; ++o)
    for (int pr = 0; pr < 25; ++pr)
      for (int h = 0; h < 2; ++h) {
        const int r = taps[pr][h][0], sx = taps[pr][h][1];
        if (r < 0) continue;
        for (int c = 0; c < 4; ++c)
          dst[(size_t)o * 204 + pr * 8 + h * 4 + c] = f.w[(((size_t)o * 4 + c) * 7 + r) * 7 + sx];
      }
}

}  // namespace

// returns "" on success, else the missing key / problem
std::string pack_blob(const TensorMap& t, std::vector<float>& blob) {
  for (const ExpectedTensor& e : expected_tensors())
    if (!t.count(e.key)) return "missing state_dict tensor: " + e.key;
  const BlobLayout L = blob_layout();
  blob.assign(L.total, 0.f);
  uint32_t hdr[4] = {BLOB_MAGIC, BLOB_VERSION, (uint32_t)L.total, 0};
  std::memcpy(blob.data(), hdr, sizeof(hdr));

  const char* stems[2] = {"convA1", "convB1"};
  for (int br = 0; br < 2; ++br) {
    Folded f = fold(t, std::string(stems[br]) + ".0", std::string(stems[br]) + ".1");
    pack_stem(f, blob.data() + L.stem_w + (size_t)br * 64 * 204);
    std::memcpy(blob.data() + L.stem_b + br * 64, f.b.data(), 64 * sizeof(float));
  }
  struct Src { ConvId id; int group; int o_off; int cout_total; const char* conv; const char* bn; };
  const Src srcs[] = {
      {L64_1, 0, 0, 64, "convA2.conv1", "convA2.bn1"},   {L64_1, 1, 0, 64, "convB2.conv1", "convB2.bn1"},
      {L64_2, 0, 0, 64, "convA2.conv2", "convA2.bn2"},   {L64_2, 1, 0, 64, "convB2.conv2", "convB2.bn2"},
      {L64_3, 0, 0, 64, "convB3.conv1", "convB3.bn1"},   {L64_4, 0, 0, 64, "convB3.conv2", "convB3.bn2"},
      {LAB1, 0, 0, 256, "convAB1.0", "convAB1.1"},
      {LAB2_1, 0, 0, 256, "convAB2.conv1", "convAB2.bn1"}, {LAB2_2, 0, 0, 256, "convAB2.conv2", "convAB2.bn2"},
      {LH1, 0, 0, 1024, "trans_conv1.0", "trans_conv1.1"}, {LH1, 0, 512, 1024, "rot_conv1.0", "rot_conv1.1"},
      {LH2_1, 0, 0, 512, "trans_conv2.conv1", "trans_conv2.bn1"}, {LH2_1, 1, 0, 512, "rot_conv2.conv1", "rot_conv2.bn1"},
      {LH2_2, 0, 0, 512, "trans_conv2.conv2", "trans_conv2.bn2"}, {LH2_2, 1, 0, 512, "rot_conv2.conv2", "rot_conv2.bn2"},
  };
  const Conv3* spec = conv_specs();
  for (const Src& s : srcs) {
    Folded f = fold(t, s.conv, s.bn);
    const size_t gw = conv3_words(spec[s.id].cin, spec[s.id].cout);
    pack3(f, blob.data() + L.conv_w[s.id] + gw * s.group, s.cout_total, s.o_off);
    std::memcpy(blob.data() + L.conv_b[s.id] + (size_t)spec[s.id].cout * s.group + s.o_off, f.b.data(),
                f.b.size() * sizeof(float));
  }
  const char* heads[2] = {"trans_out", "rot_out"};
  for (int h = 0; h < 2; ++h) {
    const StoredTensor& W = t.at(std::string(heads[h]) + ".0.weight");
    const StoredTensor& B = t.at(std::string(heads[h]) + ".0.bias");
    std::memcpy(blob.data() + L.fc_w + (size_t)h * 3 * 512, W.data.data(), 3 * 512 * sizeof(float));
    std::memcpy(blob.data() + L.fc_b + h * 4, B.data.data(), 3 * sizeof(float));
  }
  return "";
}

// Host statement of what launch_split_weights (kernels_misc.hip) derives on the device for the f16x3 mode, from the PACKED
// float32 panels: per cout row an exact power-of-two scaling 2^k (split_exponent: the row's largest |w| lands in
// [2^10, 2^11), so the f16 `lo` parts never underflow), then w 2^k = hi + lo with hi = f16(w 2^k), lo = f16(w 2^k - hi),
// stored as split rows [chunk][tap][cout][32 f16 hi | 32 f16 lo] (stems: 16-byte entries 4 hi | 4 lo) + 2^-k per cout.
// Used by the tests to check the device derivation bit for bit (se3tn_split_weights_host); not on the product path.
void split_blob_host(const float* blob, float* split) {
  const BlobLayout L = blob_layout();
  const SplitLayout S = split_layout();
  std::memset(split, 0, S.total * sizeof(float));
  const Conv3* spec = conv_specs();
  for (int id = 0; id < NUM_CONV3; ++id) {
    const int nch = spec[id].cin / 32, cout = spec[id].cout, rows = nch * 9;
    for (int g = 0; g < spec[id].groups; ++g) {
      const float* w = blob + L.conv_w[id] + conv3_words(spec[id].cin, cout) * g;
      uint16_t* d16 = reinterpret_cast<uint16_t*>(split + S.conv_ws[id] + conv3_words(spec[id].cin, cout) * g);
      float* sc = split + S.conv_sc[id] + (size_t)cout * g;
      for (int o = 0; o < cout; ++o) {
        float mx = 0.f;
        for (int r = 0; r < rows; ++r)
          for (int ci = 0; ci < 32; ++ci) mx = std::fmax(mx, std::fabs(w[((size_t)r * cout + o) * 32 + ci]));
        const int k = split_exponent(mx);
        const float scl = std::ldexp(1.0f, k);
        sc[o] = std::ldexp(1.0f, -k);
        for (int r = 0; r < rows; ++r) {
          uint16_t* row = d16 + (((size_t)r * cout + o) * 32) * 2;
          for (int ci = 0; ci < 32; ++ci) {
            const float v = w[((size_t)r * cout + o) * 32 + ci] * scl;  // exact (power of 2)
            const uint16_t hi = f32_to_f16_bits(v);
            row[ci] = hi;
            row[32 + ci] = f32_to_f16_bits(v - f16_bits_to_f32(hi));
          }
        }
      }
    }
  }
  for (int br = 0; br < 2; ++br) {
    const float* w = blob + L.stem_w + (size_t)br * 64 * 204;
    uint16_t* d16 = reinterpret_cast<uint16_t*>(split + S.stem_ws + (size_t)br * 64 * 204);
    for (int o = 0; o < 64; ++o) {
      float mx = 0.f;
      for (int k = 0; k < 200; ++k) mx = std::fmax(mx, std::fabs(w[(size_t)o * 204 + k]));
      const int kx = split_exponent(mx);
      const float scl = std::ldexp(1.0f, kx);
      split[S.stem_sc + br * 64 + o] = std::ldexp(1.0f, -kx);
      for (int e = 0; e < 50; ++e)       // 25 tap pairs x 2 halves
        for (int c = 0; c < 4; ++c) {
          const float v = w[(size_t)o * 204 + e * 4 + c] * scl;
          const uint16_t hi = f32_to_f16_bits(v);
          d16[((size_t)o * 204 + e * 4) * 2 + c] = hi;
          d16[((size_t)o * 204 + e * 4) * 2 + 4 + c] = f32_to_f16_bits(v - f16_bits_to_f32(hi));
        }
    }
  }
}

}  // namespace se3tn
