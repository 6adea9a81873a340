// Host-side container of the reference's state_dict tensors and the packer entry points.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace se3tn {
struct StoredTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};
typedef std::map<std::string, StoredTensor> TensorMap;
struct ExpectedTensor {
  std::string key;
  std::vector<int64_t> shape;
};
// float32 entries of Se3TrackNet.state_dict() (106 tensors), se3_tracknet.py:57-78
const std::vector<ExpectedTensor>& expected_tensors();
// fold BN + pack; returns "" or an error message
std::string pack_blob(const TensorMap& t, std::vector<float>& blob);
// host reference of the device-side derivation of the f16x3 split panels (split_layout() words) from a packed blob
void split_blob_host(const float* blob, float* split);
}  // namespace se3tn
