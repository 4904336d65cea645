// NCNN .param/.bin reader (host).  Replaces ncnn.Net.load_param / load_model for the
// detector graphs the reference ships (e2e.py:213-216; format: SURVEY Appendix C).
#pragma once
#include "common.h"

namespace lp {

struct NcnnLayer {
  std::string type, name;
  std::vector<std::string> inputs, outputs;
  std::map<int, double> params;                 // scalar params (ints stored exactly)
  std::map<int, std::vector<double>> arrays;    // array params (-233xx keys), keyed by id
  std::vector<float> weight, bias, data;        // Convolution weights [out][in][kh][kw] / MemoryData
  int in_ch = 0;                                // Convolution: derived from the weight count
  int ipar(int k, int dflt = 0) const {
    auto it = params.find(k);
    return it == params.end() ? dflt : (int)it->second;
  }
  double fpar(int k, double dflt = 0.0) const {
    auto it = params.find(k);
    return it == params.end() ? dflt : it->second;
  }
};

struct NcnnGraph {
  std::vector<NcnnLayer> layers;
  void load(const std::string& param_path, const std::string& bin_path);  // throws lp::Error(LP_ERR_IO / LP_ERR_GRAPH)
};

}  // namespace lp
