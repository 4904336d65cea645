#include "ncnn_graph.h"

#include <fstream>
#include <sstream>

namespace lp {

static std::vector<uint8_t> read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  LP_CHECK(f.good(), LP_ERR_IO, "cannot open %s", path.c_str());
  f.seekg(0, std::ios::end);
  const std::streamoff n = f.tellg();
  f.seekg(0);
  std::vector<uint8_t> buf((size_t)n);
  if (n) f.read(reinterpret_cast<char*>(buf.data()), n);
  LP_CHECK(f.good() || f.eof(), LP_ERR_IO, "short read on %s", path.c_str());
  return buf;
}

void NcnnGraph::load(const std::string& param_path, const std::string& bin_path) {
  layers.clear();
  std::ifstream f(param_path);
  LP_CHECK(f.good(), LP_ERR_IO, "Failed to load param: %s", param_path.c_str());
  std::string line;
  LP_CHECK((bool)std::getline(f, line), LP_ERR_IO, "empty param file %s", param_path.c_str());
  {
    std::istringstream is(line);
    std::string magic;
    is >> magic;
    LP_CHECK(magic == "7767517", LP_ERR_IO, "%s: bad NCNN magic '%s'", param_path.c_str(), magic.c_str());
  }
  int n_layers = 0, n_blobs = 0;
  {
    LP_CHECK((bool)std::getline(f, line), LP_ERR_IO, "%s: missing layer/blob counts", param_path.c_str());
    std::istringstream is(line);
    is >> n_layers >> n_blobs;
    LP_CHECK(n_layers > 0, LP_ERR_IO, "%s: bad layer count", param_path.c_str());
  }
  while (std::getline(f, line)) {
    std::istringstream is(line);
    NcnnLayer L;
    int n_in = 0, n_out = 0;
    if (!(is >> L.type >> L.name >> n_in >> n_out)) continue;
    for (int i = 0; i < n_in; ++i) { std::string s; is >> s; L.inputs.push_back(s); }
    for (int i = 0; i < n_out; ++i) { std::string s; is >> s; L.outputs.push_back(s); }
    std::string kv;
    while (is >> kv) {
      const size_t eq = kv.find('=');
      LP_CHECK(eq != std::string::npos, LP_ERR_IO, "%s: bad param token '%s' in layer %s", param_path.c_str(), kv.c_str(), L.name.c_str());
      const int key = std::stoi(kv.substr(0, eq));
      const std::string val = kv.substr(eq + 1);
      if (key <= -23300) {
        std::vector<double> arr;
        std::stringstream vs(val);
        std::string tok;
        bool first = true;
        size_t count = 0;
        while (std::getline(vs, tok, ',')) {
          if (first) { count = (size_t)std::stoul(tok); first = false; continue; }
          arr.push_back(std::stod(tok));
        }
        LP_CHECK(arr.size() == count, LP_ERR_IO, "layer %s: array param count mismatch", L.name.c_str());
        L.arrays[-key - 23300] = arr;
      } else {
        L.params[key] = std::stod(val);
      }
    }
    layers.push_back(std::move(L));
  }
  LP_CHECK((int)layers.size() == n_layers, LP_ERR_IO, "%s: %zu layers parsed, header says %d", param_path.c_str(), layers.size(), n_layers);

  // weights: read strictly in layer order
  std::ifstream probe(bin_path, std::ios::binary);
  LP_CHECK(probe.good(), LP_ERR_IO, "Failed to load bin: %s", bin_path.c_str());
  probe.close();
  const std::vector<uint8_t> blob = read_file(bin_path);
  size_t off = 0;
  auto take = [&](std::vector<float>& dst, size_t n, const std::string& who) {
    LP_CHECK(off + 4 * n <= blob.size(), LP_ERR_IO, "%s: weight blob too short at layer %s", bin_path.c_str(), who.c_str());
    dst.resize(n);
    if (n) memcpy(dst.data(), blob.data() + off, 4 * n);
    off += 4 * n;
  };
  for (auto& L : layers) {
    if (L.type == "Convolution" || L.type == "ConvolutionDepthWise") {  // depthwise: weights [C][1][kh][kw], in_ch = 1
      const int out_ch = L.ipar(0), kw = L.ipar(1, 1), kh = L.ipar(11, kw), wcount = L.ipar(6);
      LP_CHECK(out_ch > 0 && kw > 0 && wcount > 0 && wcount % (out_ch * kw * kh) == 0, LP_ERR_GRAPH,
               "layer %s: inconsistent convolution parameters", L.name.c_str());
      LP_CHECK(off + 4 <= blob.size(), LP_ERR_IO, "%s: weight blob too short at layer %s", bin_path.c_str(), L.name.c_str());
      uint32_t flag;
      memcpy(&flag, blob.data() + off, 4);
      off += 4;
      LP_CHECK(flag == 0, LP_ERR_GRAPH, "layer %s: weight storage flag %#x unsupported (only raw fp32)", L.name.c_str(), flag);
      L.in_ch = wcount / (out_ch * kw * kh);
      take(L.weight, (size_t)wcount, L.name);
      if (L.ipar(5, 0)) take(L.bias, (size_t)out_ch, L.name);
    } else if (L.type == "MemoryData") {
      const int w = L.ipar(0), h = L.ipar(1), c = L.ipar(2);
      size_t n = (size_t)(w > 0 ? w : 1);
      if (h > 0) n *= h;
      if (c > 0) n *= c;
      take(L.data, n, L.name);
    }
  }
  LP_CHECK(off == blob.size(), LP_ERR_IO, "%s: %zu bytes consumed of %zu", bin_path.c_str(), off, blob.size());
}

}  // namespace lp
