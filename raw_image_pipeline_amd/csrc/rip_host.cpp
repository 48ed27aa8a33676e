// rip_host.cpp -- host-side parameter handling and constant-table builders (see rip_host.hpp).
// Float expressions here define bit patterns the kernels depend on: compile with
// -ffp-contract=off (build.py does).
#include "rip_host.hpp"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <thread>

namespace rip {

// =============================================================================================
// YAML subset
// =============================================================================================
namespace {
const YamlNode kNullNode;

std::string trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n");
  if (a == std::string::npos) return "";
  size_t b = s.find_last_not_of(" \t\r\n");
  return s.substr(a, b - a + 1);
}

std::string strip_comment(const std::string& line) {
  bool sq = false, dq = false;
  for (size_t i = 0; i < line.size(); i++) {
    char c = line[i];
    if (c == '\'' && !dq) sq = !sq;
    if (c == '"' && !sq) dq = !dq;
    if (c == '#' && !sq && !dq && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) return line.substr(0, i);
  }
  return line;
}

std::string unquote(const std::string& s) {
  if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
    return s.substr(1, s.size() - 2);
  return s;
}

struct Line {
  int indent;
  std::string text;
  int number;
};

std::vector<std::string> split_flow(const std::string& body, int line_no) {
  std::vector<std::string> out;
  std::string cur;
  for (char c : body) {
    if (c == ',') {
      out.push_back(unquote(trim(cur)));
      cur.clear();
    } else if (c == '[' || c == ']' || c == '{' || c == '}') {
      throw YamlError("yaml: nested flow collections are not supported (line " + std::to_string(line_no) + ")");
    } else {
      cur += c;
    }
  }
  if (!trim(cur).empty()) out.push_back(unquote(trim(cur)));
  return out;
}

// Nesting the reader follows (both parsers recurse once per level; a document nested deeper than any configuration file of
// the reference -- three levels -- by an order of magnitude is refused instead of overflowing the stack: tests/test_host_fuzz.py)
constexpr int kMaxYamlDepth = 64;

// flow map starting at text[at] == '{'; leaves `at` behind the closing brace
void parse_flow_map(const std::string& text, size_t& at, int line_no, YamlNode& node, int depth = 0) {
  auto fail = [&](const char* what) { throw YamlError(std::string("yaml: ") + what + " (line " + std::to_string(line_no) + ")"); };
  if (depth > kMaxYamlDepth) fail("flow maps nested too deeply");
  auto skip_ws = [&]() { while (at < text.size() && (text[at] == ' ' || text[at] == '\t')) at++; };
  node.kind = YamlNode::Map;
  at++;  // '{'
  for (;;) {
    skip_ws();
    if (at >= text.size()) fail("unterminated flow map");
    if (text[at] == '}') { at++; return; }
    size_t colon = text.find(':', at);
    if (colon == std::string::npos) fail("expected 'key: value' in flow map");
    const std::string key = unquote(trim(text.substr(at, colon - at)));
    if (key.empty() || key.find_first_of("{}[],") != std::string::npos) fail("bad key in flow map");
    at = colon + 1;
    skip_ws();
    YamlNode child;
    if (at < text.size() && text[at] == '{') {
      parse_flow_map(text, at, line_no, child, depth + 1);
    } else if (at < text.size() && text[at] == '[') {
      size_t close = text.find(']', at);
      if (close == std::string::npos) fail("unterminated '[' in flow map");
      child.kind = YamlNode::Sequence;
      child.seq = split_flow(text.substr(at + 1, close - at - 1), line_no);
      at = close + 1;
    } else {
      size_t end = text.find_first_of(",}", at);
      if (end == std::string::npos) fail("unterminated flow map");
      const std::string v = trim(text.substr(at, end - at));
      if (!v.empty()) {
        child.kind = YamlNode::Scalar;
        child.scalar = unquote(v);
      }
      at = end;
    }
    node.map[key] = child;
    skip_ws();
    if (at < text.size() && text[at] == ',') at++;
    else if (at >= text.size() || text[at] != '}') fail("expected ',' or '}' in flow map");
  }
}

// parses lines[pos..) with indentation == indent into `node` (a map)
void parse_block(const std::vector<Line>& lines, size_t& pos, int indent, YamlNode& node, int depth = 0) {
  node.kind = YamlNode::Map;
  if (depth > kMaxYamlDepth) throw YamlError("yaml: blocks nested too deeply at line " + std::to_string(pos < lines.size() ? lines[pos].number : 0));
  while (pos < lines.size()) {
    const Line& ln = lines[pos];
    if (ln.indent < indent) return;
    if (ln.indent > indent) throw YamlError("yaml: unexpected indentation at line " + std::to_string(ln.number));
    if (ln.text[0] == '-') throw YamlError("yaml: block sequences are not supported (line " + std::to_string(ln.number) + ")");
    size_t colon = std::string::npos;
    {
      bool sq = false, dq = false;
      for (size_t i = 0; i < ln.text.size(); i++) {
        char c = ln.text[i];
        if (c == '\'' && !dq) sq = !sq;
        if (c == '"' && !sq) dq = !dq;
        if (c == ':' && !sq && !dq && (i + 1 == ln.text.size() || ln.text[i + 1] == ' ')) {
          colon = i;
          break;
        }
      }
    }
    if (colon == std::string::npos) throw YamlError("yaml: expected 'key: value' at line " + std::to_string(ln.number));
    std::string key = unquote(trim(ln.text.substr(0, colon)));
    std::string val = trim(ln.text.substr(colon + 1));
    pos++;
    YamlNode child;
    if (val.empty()) {
      if (pos < lines.size() && lines[pos].indent > indent) {
        parse_block(lines, pos, lines[pos].indent, child, depth + 1);
      }  // else: null
    } else if (val[0] == '[') {
      std::string body = val.substr(1);
      int number = ln.number;
      while (body.find(']') == std::string::npos) {
        if (pos >= lines.size()) throw YamlError("yaml: unterminated '[' opened at line " + std::to_string(number));
        body += " " + lines[pos].text;
        pos++;
      }
      size_t close = body.find(']');
      if (!trim(body.substr(close + 1)).empty()) throw YamlError("yaml: trailing characters after ']' (line " + std::to_string(number) + ")");
      child.kind = YamlNode::Sequence;
      child.seq = split_flow(body.substr(0, close), number);
    } else if (val[0] == '{') {
      // flow map {key: value, key: [a, b], key: {...}}; may continue on the following lines
      std::string body = val;
      int number = ln.number;
      auto depth_of = [](const std::string& t) {
        int d = 0;
        for (char c : t) d += (c == '{' || c == '[') ? 1 : ((c == '}' || c == ']') ? -1 : 0);
        return d;
      };
      while (depth_of(body) > 0) {
        if (pos >= lines.size()) throw YamlError("yaml: unterminated '{' opened at line " + std::to_string(number));
        body += " " + lines[pos].text;
        pos++;
      }
      size_t at = 0;
      parse_flow_map(body, at, number, child);
      if (!trim(body.substr(at)).empty()) throw YamlError("yaml: trailing characters after '}' (line " + std::to_string(number) + ")");
    } else {
      child.kind = YamlNode::Scalar;
      child.scalar = unquote(val);
    }
    node.map[key] = child;
  }
}

bool parse_double(const std::string& s, double& out) {
  if (s.empty()) return false;
  char* end = nullptr;
  out = std::strtod(s.c_str(), &end);
  return end && *end == 0;
}
}  // namespace

const YamlNode& YamlNode::operator[](const std::string& key) const {
  if (kind != Map) return kNullNode;
  auto it = map.find(key);
  return it == map.end() ? kNullNode : it->second;
}

bool YamlNode::get(const std::string& key, bool dflt) const {
  const YamlNode& n = (*this)[key];
  if (n.kind != Scalar) return dflt;
  std::string s = n.scalar;
  std::transform(s.begin(), s.end(), s.begin(), ::tolower);
  if (s == "true" || s == "yes" || s == "on" || s == "y") return true;
  if (s == "false" || s == "no" || s == "off" || s == "n") return false;
  return dflt;
}
int YamlNode::get(const std::string& key, int dflt) const {
  const YamlNode& n = (*this)[key];
  double d;
  if (n.kind != Scalar || !parse_double(n.scalar, d)) return dflt;
  // a value no int holds (1e300, inf, nan): yaml-cpp's as<int>() throws BadConversion, which utils::get (utils.hpp:62-73) lets
  // through; the conversion itself would be undefined behaviour here (found by the sanitizer run of tests/test_host_fuzz.py)
  if (!(d >= -2147483648.0 && d <= 2147483647.0)) throw YamlError("yaml: '" + n.scalar + "' for '" + key + "' is not an integer in range");
  return (int)d;
}
double YamlNode::get(const std::string& key, double dflt) const {
  const YamlNode& n = (*this)[key];
  double d;
  if (n.kind != Scalar || !parse_double(n.scalar, d)) return dflt;
  return d;
}
std::string YamlNode::get(const std::string& key, const std::string& dflt) const {
  const YamlNode& n = (*this)[key];
  return n.kind == Scalar ? n.scalar : dflt;
}
std::vector<double> YamlNode::get_vector(const std::string& key) const {
  const YamlNode& n = (*this)[key];
  std::vector<double> out;
  if (n.kind != Sequence) return out;
  for (const std::string& s : n.seq) {
    double d;
    if (!parse_double(s, d)) throw YamlError("yaml: '" + s + "' in sequence '" + key + "' is not a number");
    out.push_back(d);
  }
  return out;
}

YamlNode yaml_parse(const std::string& text) {
  std::vector<Line> lines;
  std::istringstream ss(text);
  std::string raw;
  int number = 0;
  while (std::getline(ss, raw)) {
    number++;
    std::string s = strip_comment(raw);
    if (trim(s).empty()) continue;
    if (trim(s) == "---" || trim(s) == "...") continue;
    if (s.find('\t') != std::string::npos && s.find_first_not_of(" \t") > s.find('\t'))
      throw YamlError("yaml: tab used for indentation at line " + std::to_string(number));
    int indent = (int)s.find_first_not_of(' ');
    lines.push_back({indent, trim(s), number});
  }
  YamlNode root;
  root.kind = YamlNode::Map;
  size_t pos = 0;
  if (!lines.empty()) parse_block(lines, pos, lines[0].indent, root);
  if (pos != lines.size()) throw YamlError("yaml: unexpected dedent at line " + std::to_string(lines[pos].number));
  return root;
}

bool file_exists(const std::string& path) {
  std::ifstream f(path);
  return f.good();
}

YamlNode yaml_load_file(const std::string& path) {
  std::ifstream f(path);
  if (!f.good()) throw YamlError("yaml: cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return yaml_parse(ss.str());
}

// =============================================================================================
// Parameter loaders
// =============================================================================================
namespace {
void set_enhancer_gains_from_config(Modules& m, double hue, double sat, double val) {
  // Through the public setters (color_enhancer.cpp:23-33): setHueGain -> value_gain_,
  // setSaturationGain -> saturation_gain_, setValueGain -> hue_gain_.  The reference's YAML
  // loader calls setHueGain three times (raw_image_pipeline.cpp:143-145) and leaves two
  // members uninitialised; here each key goes through its own setter (documented fix, Q7).
  m.ce_value_gain = hue;
  m.ce_saturation_gain = sat;
  m.ce_hue_gain = val;
}

void params_from_node(Modules& m, const YamlNode& node) {
  // raw_image_pipeline.cpp:54-160, defaults as written there
  m.debayer_enabled = node["debayer"].get("enabled", true);
  m.debayer_encoding = node["debayer"].get("encoding", std::string("auto"));
  m.flip_enabled = node["flip"].get("enabled", false);
  m.flip_angle = node["flip"].get("angle", 0);
  const YamlNode& wb = node["white_balance"];
  m.wb_enabled = wb.get("enabled", false);
  m.wb_method = wb.get("method", std::string("ccc"));
  m.wb_percentile = wb.get("clipping_percentile", 20.0);
  m.wb_bright_thr = wb.get("saturation_bright_thr", 0.8);
  m.wb_dark_thr = wb.get("saturation_dark_thr", 0.1);
  m.wb_temporal = wb.get("temporal_consistency", true);
  m.cc_enabled = node["color_calibration"].get("enabled", false);
  const YamlNode& g = node["gamma_correction"];
  m.gamma_enabled = g.get("enabled", false);
  m.gamma_method = g.get("method", std::string("custom"));
  m.gamma_k = g.get("k", 0.8);
  const YamlNode& v = node["vignetting_correction"];
  m.vig_enabled = v.get("enabled", false);
  m.vig_scale = v.get("scale", 1.5);
  m.vig_a2 = v.get("a2", 1e-3);
  m.vig_a4 = v.get("a4", 1e-6);
  const YamlNode& ce = node["color_enhancer"];
  m.ce_enabled = ce.get("run_color_enhancer", false);  // key name as in the reference (:137)
  set_enhancer_gains_from_config(m, ce.get("hue_gain", 1.0), ce.get("saturation_gain", 1.0), ce.get("value_gain", 1.0));
  const YamlNode& u = node["undistortion"];
  m.und_enabled = u.get("enabled", false);
  m.balance = u.get("balance", 0.0);
  m.fov_scale = u.get("fov_scale", 1.0);
}

void set_camera(Modules& m, int w, int h, const std::vector<double>& K, const std::vector<double>& D,
                const std::string& model, const std::vector<double>& R, const std::vector<double>& P) {
  // the six setters of undistortion.cpp:23-72, each of which writes both the dist_ and rect_ copy
  m.dist_w = m.rect_w = w;
  m.dist_h = m.rect_h = h;
  if (K.size() < 9 || D.size() < 4 || R.size() < 9 || P.size() < 12)
    throw YamlError("camera calibration: camera_matrix needs 9, distortion_coefficients 4, rectification_matrix 9 and projection_matrix 12 values");
  for (int i = 0; i < 9; i++) m.dist_K[i] = m.rect_K[i] = K[i];
  for (int i = 0; i < 4; i++) m.dist_D[i] = m.rect_D[i] = D[i];
  m.dist_model = m.rect_model = model;
  for (int i = 0; i < 9; i++) m.dist_R[i] = m.rect_R[i] = R[i];
  for (int i = 0; i < 12; i++) m.dist_P[i] = m.rect_P[i] = P[i];
}
}  // namespace

void apply_example_params(Modules& m) {
  // values of the reference's config/pipeline_params_example.yaml (facts, not text)
  m.debayer_enabled = true;
  m.debayer_encoding = "auto";
  m.flip_enabled = false;
  m.flip_angle = 0;
  m.wb_enabled = true;
  m.wb_method = "ccc";
  m.wb_percentile = 20;
  m.wb_bright_thr = 0.8;
  m.wb_dark_thr = 0.2;
  m.wb_temporal = false;
  m.cc_enabled = false;
  m.gamma_enabled = false;
  m.gamma_method = "custom";
  m.gamma_k = 0.8;
  m.vig_enabled = false;
  m.vig_scale = 1.5;
  m.vig_a2 = 1e-3;
  m.vig_a4 = 1e-6;
  m.ce_enabled = false;
  set_enhancer_gains_from_config(m, 1.0, 1.5, 1.0);
  m.und_enabled = true;
  m.balance = 0.0;
  m.fov_scale = 0.8;
}

bool load_params_file(Modules& m, const std::string& path) {
  if (!file_exists(path)) return false;  // "Warning: parameters file doesn't exist" (:162-164)
  params_from_node(m, yaml_load_file(path));
  return true;
}

bool load_camera_calibration_file(Modules& m, const std::string& path) {
  if (!file_exists(path)) {
    // undistortion.cpp:175-194
    m.und_available = false;
    set_camera(m, 320, 240, {1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0, 0}, "none", {1, 0, 0, 0, 1, 0, 0, 0, 1},
               {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0});
    return false;
  }
  YamlNode n = yaml_load_file(path);
  set_camera(m, n.get("image_width", 320), n.get("image_height", 240), n["camera_matrix"].get_vector("data"),
             n["distortion_coefficients"].get_vector("data"), n.get("distortion_model", std::string("none")),
             n["rectification_matrix"].get_vector("data"), n["projection_matrix"].get_vector("data"));
  m.und_available = true;
  return true;
}

bool load_color_calibration_file(Modules& m, const std::string& path) {
  if (!file_exists(path)) {
    m.cc_available = false;  // color_calibration.cpp:72-75
    return false;
  }
  YamlNode n = yaml_load_file(path);
  std::vector<double> mat = n["matrix"].get_vector("data");
  std::vector<double> bias = n["bias"].get_vector("data");
  if (mat.size() != 9) mat = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // utils::get default on decode failure
  if (bias.size() != 3) bias = {0, 0, 0};
  for (int i = 0; i < 9; i++) m.cc_matrix[i] = (float)mat[i];  // Matx33d -> Matx33f (:79)
  for (int i = 0; i < 3; i++) m.cc_bias[i] = bias[i];
  m.cc_bias[3] = 0;
  m.cc_available = true;
  return true;
}

void apply_example_camera_calibration(Modules& m) {
  // values of config/alphasense_calib_example.yaml
  set_camera(m, 720, 540,
             {347.548139773951, 0.0, 342.454373227748, 0.0, 347.434712422309, 271.368057185649, 0.0, 0.0, 1.0},
             {-0.0396482888762527, -0.00367688950406141, 0.00391742438164282, -0.00178738156007817}, "equidistant",
             {1, 0, 0, 0, 1, 0, 0, 0, 1},
             {347.548139773951, 0.0, 342.454373227748, 0.0, 0.0, 347.434712422309, 271.368057185649, 0.0, 0.0, 0.0, 1.0, 0.0});
  m.und_available = true;
}

void apply_example_color_calibration(Modules& m) {
  // values of config/alphasense_color_calib_example.yaml
  const double mat[9] = {2.4276948, 0.21479778, -0.30818, 0.09277014, 1.1962607, -0.09772757, -0.24436986, -0.22239459, 2.099912};
  for (int i = 0; i < 9; i++) m.cc_matrix[i] = (float)mat[i];
  for (int i = 0; i < 4; i++) m.cc_bias[i] = 0;
  m.cc_available = true;
}

// =============================================================================================
// Tables
// =============================================================================================
namespace {
inline int round_half_even(double v) { return (int)std::lrint(v); }
inline int round_half_even(float v) { return (int)std::lrintf(v); }
inline uint8_t clamp_u8(int v) { return (uint8_t)std::min(255, std::max(0, v)); }
}  // namespace

void build_gamma_lut(double k, uint8_t lut[256]) {
  for (int i = 0; i < 256; i++) {
    float f = (float)(i / 255.0);
    f = (float)std::pow((double)f, k);
    lut[i] = clamp_u8(round_half_even((double)f * 255.0));  // saturate_cast<uchar>(double)
  }
}

namespace {
// OpenCV's cubeRoot(): exponent split + quartic rational polynomial (relative error < 2^-24)
float opencv_cbrt(float value) {
  uint32_t bits;
  std::memcpy(&bits, &value, 4);
  if ((bits << 1) == 0) return 0.f;
  uint32_t sign = bits & 0x80000000u;
  int32_t mag = (int32_t)(bits & 0x7fffffffu);
  int e = (mag >> 23) - 127;
  int rem = e % 3;
  if (rem >= 0) rem -= 3;
  int e3 = (e - rem) / 3;
  uint32_t mant_bits = (uint32_t)((mag & 0x7fffff) | ((rem + 127) << 23));
  float mant;
  std::memcpy(&mant, &mant_bits, 4);
  double t = mant;  // 0.125 <= t < 1
  double num = (((45.2548339756803022511987494 * t + 192.2798368355061050458134625) * t + 119.1654824285581628956914143) * t +
                13.43250139086239872172837314) * t + 0.1636161226585754240958355063;
  double den = (((14.80884093219134573786480845 * t + 151.9714051044435648658557668) * t + 168.5254414101568283957668343) * t +
                33.9905941350215598754191872) * t + 1.0;
  float root = (float)(num / den);
  uint32_t rb;
  std::memcpy(&rb, &root, 4);
  rb = rb + ((uint32_t)e3 << 23) + sign;
  float out;
  std::memcpy(&out, &rb, 4);
  return out;
}

float srgb_to_linear(float x) {
  const float threshold = 809.f / 20000.f, low_scale = 323.f / 25.f, power = 12.f / 5.f, shift = 11.f / 200.f;
  if (x <= threshold) return x / low_scale;
  float base = (x + shift) / (1.0f + shift);
  return (float)std::pow((double)base, (double)power);
}

float linear_to_srgb(float x) {
  const float threshold = 7827.f / 2500000.f, low_scale = 323.f / 25.f, power = 12.f / 5.f, shift = 11.f / 200.f;
  if (x <= threshold) return x * low_scale;
  float inv_power = 1.0f / power;
  float p = (float)std::pow((double)x, (double)inv_power);
  return p * (1.0f + shift) - shift;
}

ColorTables make_color_tables() {
  ColorTables t;
  const int gamma_shift = 3, lab_shift = 12, lab_shift2 = lab_shift + gamma_shift, base = 1 << 14;
  const float gamma_scale = 255.f * (1 << gamma_shift);
  for (int i = 0; i < 256; i++) t.srgb_gamma[i] = (uint16_t)round_half_even(gamma_scale * srgb_to_linear((float)i / 255.f));
  for (int i = 0; i < 4096; i++) t.inv_gamma[i] = (uint16_t)round_half_even(255.f * linear_to_srgb((1.0f / 4096) * (float)i));
  {
    const float thresh = 216.f / 24389.f, slope = 841.f / 108.f, offset = 16.f / 116.f;
    const float step = 1.0f / gamma_scale;
    for (int i = 0; i < 3072; i++) {
      float x = step * (float)i;
      float f = x < thresh ? std::fmaf(x, slope, offset) : opencv_cbrt(x);
      t.cbrt[i] = (uint16_t)round_half_even((float)(1 << lab_shift2) * f);
    }
  }
  for (int i = 0; i < 256; i++) {
    int y, ify;
    if (i <= 20) {  // L <= 8: linear segment of f^-1
      y = round_half_even((float)(i * base * 20 * 9) / (float)(17 * 29 * 29 * 29));
      ify = round_half_even((float)base * (16.f / 116.f + (float)(i * 5) / (float)(3 * 17 * 29)));
    } else {
      float fy = (float)(i * 100 * base) / (float)(255 * 116) + (float)(16 * base) / 116.f;
      ify = round_half_even(fy);
      y = round_half_even(fy * fy * fy / (float)(base * base));
    }
    t.lab_to_yf[2 * i] = (uint16_t)y;
    t.lab_to_yf[2 * i + 1] = (uint16_t)ify;
  }
  const double rgb2xyz[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
  const double xyz2rgb[3][3] = {{3.240479, -1.53715, -0.498535}, {-0.969256, 1.875991, 0.041556}, {0.055648, -0.204043, 1.057311}};
  const double white[3] = {0.950456, 1., 1.088754};
  const double q12 = 4096.0;
  for (int row = 0; row < 3; row++)       // X, Y, Z
    for (int ch = 0; ch < 3; ch++)        // memory order B, G, R  <-  matrix column R, G, B
      t.fwd[row * 3 + ch] = round_half_even(q12 * rgb2xyz[row][2 - ch] / white[row]);
  for (int ch = 0; ch < 3; ch++)          // output rows B, G, R  <-  matrix rows R, G, B
    for (int k = 0; k < 3; k++)           // multiplies X, Y, Z
      t.inv[ch * 3 + k] = round_half_even(q12 * xyz2rgb[2 - ch][k] * white[k]);
  t.sdiv[0] = t.hdiv180[0] = 0;
  for (int i = 1; i < 256; i++) {
    t.sdiv[i] = round_half_even((255 << 12) / (1. * i));
    t.hdiv180[i] = round_half_even((180 << 12) / (6. * i));
  }
  return t;
}
}  // namespace

const ColorTables& color_tables() {
  static const ColorTables t = make_color_tables();
  return t;
}

// ---------------------------------------------------------------------------------------------
// Vignetting
// ---------------------------------------------------------------------------------------------
// vignetting_correction.cpp:32-63, operation by operation: r = sqrt(pow(dy, 2) + pow(dx, 2)) in double,
// k = pow(r, 2) * a2 + pow(r, 4) * a4, stored as float; mask = k / max (x float(1 / max)), x float(scale), + 1.0f.
// The same libm the reference would call does the work (the sqrt -> pow detour matters: with a2 = 1e-3, a4 = 1e-6
// k lands on float rounding ties often enough that the algebraically equal s * a2 + s^2 * a4 differs in ~1e-3 of
// the pixels).  k depends on (|row - rows/2|, |col - cols/2|) only, so one quadrant is evaluated and mirrored --
// the per-pixel operands are identical, only the number of pow() calls drops.  Once per (geometry, parameters);
// the reference rebuilds it on every non-square frame (quirk Q6).
// fp_contract = 1: the reference's own translation unit compiled for an FMA target (aarch64): pow(., 2) is a multiply and both
// sums of products contract -- r = sqrt(fma(dx, dx, dy * dy)), k = fma(r^2, a2, r^4 * a4) (oracle/rip_oracle.c mode 1).
void build_vignette_mask(int rows, int cols, double scale, double a2, double a4, std::vector<float>& mask, int fp_contract) {
  mask.resize((size_t)rows * cols);
  const double cy = rows / 2.0, cx = cols / 2.0;
  // distinct |2 * d| values per axis: index = |2 * i - n|
  std::vector<float> quad((size_t)(rows + 1) * (cols + 1), 0.f);
  std::vector<char> have((size_t)(rows + 1) * (cols + 1), 0);
  float mx = -FLT_MAX;
  for (int row = 0; row < rows; row++) {
    const int ay = std::abs(2 * row - rows);
    for (int col = 0; col < cols; col++) {
      const int ax = std::abs(2 * col - cols);
      const size_t qi = (size_t)ay * (cols + 1) + ax;
      if (!have[qi]) {
        const double dy = std::fabs(row - cy), dx = std::fabs(col - cx);
        double r = std::sqrt(std::pow(dx, 2) + std::pow(dy, 2));
        double k = std::pow(r, 2) * a2 + std::pow(r, 4) * a4;
        if (fp_contract) {
          r = std::sqrt(std::fma(dx, dx, dy * dy));
          k = std::fma(std::pow(r, 2), a2, std::pow(r, 4) * a4);
        }
        quad[qi] = (float)k;
        have[qi] = 1;
      }
      const float kf = quad[qi];
      mask[(size_t)row * cols + col] = kf;
      if (kf > mx) mx = kf;
    }
  }
  const double maxv = (double)mx;
  const size_t n = mask.size();
  if (maxv > 0) {
    const float inv = (float)(1.0 / maxv);
    for (size_t i = 0; i < n; i++) mask[i] = mask[i] * inv;
  }
  const float sc = (float)scale;
  for (size_t i = 0; i < n; i++) mask[i] = mask[i] * sc;
  for (size_t i = 0; i < n; i++) mask[i] = mask[i] + 1.0f;
}

// ---------------------------------------------------------------------------------------------
// Fisheye
// ---------------------------------------------------------------------------------------------
namespace {
struct Mat3 {
  double v[9];
};
Mat3 mul3(const double* a, const double* b) {
  Mat3 c;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) c.v[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
  return c;
}
Mat3 inverse3(const Mat3& m) {
  const double* a = m.v;
  double c0 = a[4] * a[8] - a[5] * a[7];
  double c1 = a[5] * a[6] - a[3] * a[8];
  double c2 = a[3] * a[7] - a[4] * a[6];
  double inv_det = 1.0 / (a[0] * c0 + a[1] * c1 + a[2] * c2);
  Mat3 o;
  o.v[0] = c0 * inv_det;
  o.v[1] = (a[2] * a[7] - a[1] * a[8]) * inv_det;
  o.v[2] = (a[1] * a[5] - a[2] * a[4]) * inv_det;
  o.v[3] = c1 * inv_det;
  o.v[4] = (a[0] * a[8] - a[2] * a[6]) * inv_det;
  o.v[5] = (a[2] * a[3] - a[0] * a[5]) * inv_det;
  o.v[6] = c2 * inv_det;
  o.v[7] = (a[1] * a[6] - a[0] * a[7]) * inv_det;
  o.v[8] = (a[0] * a[4] - a[1] * a[3]) * inv_det;
  return o;
}

// cv::fisheye::undistortPoints for one point (Newton iteration on theta, <= 10 steps)
void undistort_point(const double K[9], const double D[4], const double R[9], double u, double v, double& ox, double& oy) {
  const double pi = 3.1415926535897932384626433832795;
  double wx = (u - K[2]) / K[0], wy = (v - K[5]) / K[4];
  double theta_d = std::sqrt(wx * wx + wy * wy);
  theta_d = std::min(std::max(-pi / 2., theta_d), pi / 2.);
  double scale = 1.0;
  if (theta_d > 1e-8) {
    double theta = theta_d;
    for (int it = 0; it < 10; it++) {
      double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
      double a = D[0] * t2, b = D[1] * t4, c = D[2] * t6, d = D[3] * t8;
      double fix = (theta * (1 + a + b + c + d) - theta_d) / (1 + 3 * a + 5 * b + 7 * c + 9 * d);
      theta = theta - fix;
      if (std::fabs(fix) < 1e-8) break;
    }
    scale = std::tan(theta) / theta_d;
  }
  double ux = wx * scale, uy = wy * scale;
  double rx = R[0] * ux + R[1] * uy + R[2];
  double ry = R[3] * ux + R[4] * uy + R[5];
  double rz = R[6] * ux + R[7] * uy + R[8];
  ox = rx / rz;
  oy = ry / rz;
}
}  // namespace

void fisheye_estimate_new_camera_matrix(const double K[9], const double D[4], int w, int h, const double R[9],
                                        double balance, int new_w, int new_h, double fov_scale, double newK[9]) {
  balance = std::min(std::max(balance, 0.0), 1.0);
  // mid-points of the four image edges, integer halves as in OpenCV
  const double px[4] = {(double)(w / 2), (double)w, (double)(w / 2), 0.0};
  const double py[4] = {0.0, (double)(h / 2), (double)h, (double)(h / 2)};
  double ux[4], uy[4];
  for (int i = 0; i < 4; i++) undistort_point(K, D, R, px[i], py[i], ux[i], uy[i]);
  double cx = (ux[0] + ux[1] + ux[2] + ux[3]) / 4, cy = (uy[0] + uy[1] + uy[2] + uy[3]) / 4;
  double aspect = K[0] / K[4];
  cx *= aspect;  // OpenCV 4.2 scales cn[0] here (later releases scale cn[1]); kept as in 4.2
  for (int i = 0; i < 4; i++) uy[i] *= aspect;
  double minx = DBL_MAX, miny = DBL_MAX, maxx = -DBL_MAX, maxy = -DBL_MAX;
  for (int i = 0; i < 4; i++) {
    miny = std::min(miny, uy[i]);
    maxy = std::max(maxy, uy[i]);
    minx = std::min(minx, ux[i]);
    maxx = std::max(maxx, ux[i]);
  }
  double f1 = w * 0.5 / (cx - minx), f2 = w * 0.5 / (maxx - cx);
  double f3 = h * 0.5 * aspect / (cy - miny), f4 = h * 0.5 * aspect / (maxy - cy);
  double fmin = std::min(f1, std::min(f2, std::min(f3, f4)));
  double fmax = std::max(f1, std::max(f2, std::max(f3, f4)));
  double f = balance * fmin + (1.0 - balance) * fmax;
  f *= fov_scale > 0 ? 1.0 / fov_scale : 1.0;
  double fx = f, fy = f;
  double ncx = -cx * f + w * 0.5, ncy = -cy * f + (h * aspect) * 0.5;
  fy /= aspect;
  ncy /= aspect;
  if (new_w > 0 && new_h > 0) {
    double rx = new_w / (double)w, ry = new_h / (double)h;
    fx *= rx;
    fy *= ry;
    ncx *= rx;
    ncy *= ry;
  }
  const double out[9] = {fx, 0, ncx, 0, fy, ncy, 0, 0, 1};
  std::memcpy(newK, out, sizeof(out));
}

void fisheye_inverse_PR(const double P[9], const double R[9], double iR_out[9]) {
  const Mat3 iR = inverse3(mul3(P, R));
  for (int i = 0; i < 9; i++) iR_out[i] = iR.v[i];
}

void fisheye_init_undistort_rectify_map(const double K[9], const double D[4], const double R[9],
                                        const double P[9], int w, int h, float* map_xy) {
  Mat3 iR = inverse3(mul3(P, R));
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  auto rows_fn = [&](int r0, int r1) {
    for (int i = r0; i < r1; i++) {
      double X = i * iR.v[1] + iR.v[2], Y = i * iR.v[4] + iR.v[5], W = i * iR.v[7] + iR.v[8];
      float* out = map_xy + (size_t)i * w * 2;
      for (int j = 0; j < w; j++) {
        double x = X / W, y = Y / W;
        double r = std::sqrt(x * x + y * y);
        double theta = std::atan(r);
        double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        double theta_d = theta * (1 + D[0] * t2 + D[1] * t4 + D[2] * t6 + D[3] * t8);
        double s = (r == 0) ? 1.0 : theta_d / r;
        out[2 * j] = (float)(fx * x * s + cx);
        out[2 * j + 1] = (float)(fy * y * s + cy);
        X += iR.v[0];  // accumulated along the row, as OpenCV does
        Y += iR.v[3];
        W += iR.v[6];
      }
    }
  };
  int nthreads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  if ((size_t)w * h < (1u << 18) || nthreads == 1) {
    rows_fn(0, h);
    return;
  }
  std::vector<std::thread> pool;
  int chunk = (h + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; t++) {
    int r0 = t * chunk, r1 = std::min(h, r0 + chunk);
    if (r0 < r1) pool.emplace_back(rows_fn, r0, r1);
  }
  for (auto& th : pool) th.join();
}

// ---------------------------------------------------------------------------------------------
// Compiled remap plan
// ---------------------------------------------------------------------------------------------
namespace {
inline int quantise_map(float v) {
  // cvRound(v * INTER_TAB_SIZE) as cv::remap does (imgwarp.cpp); INT_MIN for NaN / inf / overflow
  float s = v * 32.f;
  if (!(s > -2147483648.f && s < 2147483648.f)) return INT32_MIN;
  return (int)std::lrintf(s);
}
inline int sat_s16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
}  // namespace

size_t remap_tile_lds_bytes(int x0, int w, int h) {
  if (w <= 0 || h <= 0) return 0;
  // rows start at the 16-byte chunk that holds byte 3 * x0 and end with the chunk that holds the last byte
  size_t pitch = ((((size_t)x0 * 3) & 15) + (size_t)w * 3 + 15) & ~(size_t)15;
  return pitch * (size_t)h;
}

void compile_remap_plan(RemapPlan& plan, const float* map_xy, int drows, int dcols, int src_rows, int src_cols) {
  plan = RemapPlan();
  plan.drows = drows;
  plan.dcols = dcols;
  plan.src_rows = src_rows;
  plan.src_cols = src_cols;
  plan.tiles_x = (dcols + kRemapTileW - 1) / kRemapTileW;
  plan.tiles_y = (drows + kRemapTileH - 1) / kRemapTileH;
  const size_t ntiles = (size_t)plan.tiles_x * plan.tiles_y;
  const int tile_px = kRemapTileW * kRemapTileH;
  plan.words.assign(ntiles * tile_px, kRemapOutside);
  plan.tiles.assign(ntiles, RemapTile{0, 0, 0, 0});
  auto do_tiles = [&](size_t t0, size_t t1) {
    std::vector<int> ix(tile_px), iy(tile_px);
    for (size_t t = t0; t < t1; t++) {
      const int ty = (int)(t / plan.tiles_x), tx = (int)(t % plan.tiles_x);
      int minx = INT32_MAX, miny = INT32_MAX, maxx = INT32_MIN, maxy = INT32_MIN;
      uint32_t* words = plan.words.data() + t * tile_px;
      for (int r = 0; r < kRemapTileH; r++)
        for (int c = 0; c < kRemapTileW; c++) {
          const int y = ty * kRemapTileH + r, x = tx * kRemapTileW + c, k = r * kRemapTileW + c;
          if (y >= drows || x >= dcols) {
            words[k] = kRemapOutside;  // never stored
            ix[k] = INT32_MIN;
            continue;
          }
          const float* m = map_xy + ((size_t)y * dcols + x) * 2;
          const int sxq = quantise_map(m[0]), syq = quantise_map(m[1]);
          const int sx = sat_s16(sxq >> 5), sy = sat_s16(syq >> 5);
          if (sx >= src_cols || sx + 1 < 0 || sy >= src_rows || sy + 1 < 0) {
            words[k] = kRemapOutside;
            ix[k] = INT32_MIN;
          } else if (sx >= 0 && sx < src_cols - 1 && sy >= 0 && sy < src_rows - 1) {
            words[k] = ((uint32_t)(sxq & 31) << 22) | ((uint32_t)(syq & 31) << 27);  // rel coords filled below
            ix[k] = sx;
            iy[k] = sy;
            minx = std::min(minx, sx);
            maxx = std::max(maxx, sx + 1);
            miny = std::min(miny, sy);
            maxy = std::max(maxy, sy + 1);
          } else {
            words[k] = kRemapBorder;
            ix[k] = INT32_MIN;
          }
        }
      RemapTile& tile = plan.tiles[t];
      if (minx == INT32_MAX) continue;
      tile.x0 = minx;
      tile.y0 = miny;
      tile.w = maxx - minx + 1;
      tile.h = maxy - miny + 1;
      if (tile.w > 2040 || tile.h > 2040) {
        // footprint too large for the packed form: everything in this tile takes the per-tap path
        for (int k = 0; k < tile_px; k++)
          if (ix[k] != INT32_MIN) words[k] = kRemapBorder;
        tile = RemapTile{0, 0, 0, 0};
        continue;
      }
      for (int k = 0; k < tile_px; k++)
        if (ix[k] != INT32_MIN) words[k] |= (uint32_t)(ix[k] - minx) | ((uint32_t)(iy[k] - miny) << 11);
    }
  };
  int nthreads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  if (ntiles < 64 || nthreads == 1) {
    do_tiles(0, ntiles);
  } else {
    std::vector<std::thread> pool;
    size_t chunk = (ntiles + nthreads - 1) / nthreads;
    for (int i = 0; i < nthreads; i++) {
      size_t a = i * chunk, b = std::min(ntiles, a + chunk);
      if (a < b) pool.emplace_back(do_tiles, a, b);
    }
    for (auto& th : pool) th.join();
  }
  if (drows <= 65535 && dcols <= 65535) {
    for (size_t t = 0; t < ntiles; t++) {
      const int ty = (int)(t / plan.tiles_x), tx = (int)(t % plan.tiles_x);
      const uint32_t* words = plan.words.data() + t * tile_px;
      for (int k = 0; k < tile_px; k++)
        if (words[k] == kRemapBorder) {
          const int y = ty * kRemapTileH + k / kRemapTileW, x = tx * kRemapTileW + k % kRemapTileW;
          if (y < drows && x < dcols) plan.border.push_back(((uint32_t)y << 16) | (uint32_t)x);
        }
    }
  }
  for (const RemapTile& tile : plan.tiles) {
    plan.max_rect_w = std::max(plan.max_rect_w, tile.w);
    plan.max_rect_h = std::max(plan.max_rect_h, tile.h);
    plan.max_lds_bytes = std::max(plan.max_lds_bytes, remap_tile_lds_bytes(tile.x0, tile.w, tile.h));
  }
  plan.valid = true;
}

// ---------------------------------------------------------------------------------------------
// CCC model
// ---------------------------------------------------------------------------------------------
void fft256_twiddles(float re[128], float im[128]) {
  for (int k = 0; k < 128; k++) {
    double ang = -2.0 * 3.14159265358979323846 * k / 256;
    re[k] = (float)std::cos(ang);
    im[k] = (float)std::sin(ang);
  }
}

namespace {
// Reference 256-point radix-2 DIT FFT on interleaved complex data with element stride (in
// complex elements).  Butterfly order and arithmetic are the contract the kernels follow.
void host_fft256(float* data, int stride, const float* twr, const float* twi, bool inverse) {
  float buf[512];
  for (int i = 0; i < 256; i++) {
    unsigned r = (unsigned)i;
    r = ((r & 0xF0u) >> 4) | ((r & 0x0Fu) << 4);
    r = ((r & 0xCCu) >> 2) | ((r & 0x33u) << 2);
    r = ((r & 0xAAu) >> 1) | ((r & 0x55u) << 1);
    buf[2 * r] = data[2 * (size_t)i * stride];
    buf[2 * r + 1] = data[2 * (size_t)i * stride + 1];
  }
  for (int len = 2; len <= 256; len <<= 1) {
    int half = len / 2, tstep = 256 / len;
    for (int b = 0; b < 128; b++) {
      int k = b % half, lo = (b / half) * len + k, hi = lo + half;
      float wr = twr[k * tstep], wi = inverse ? -twi[k * tstep] : twi[k * tstep];
      float xr = buf[2 * hi], xi = buf[2 * hi + 1];
      float tr = wr * xr - wi * xi;
      float ti = wr * xi + wi * xr;
      float ur = buf[2 * lo], ui = buf[2 * lo + 1];
      buf[2 * lo] = ur + tr;
      buf[2 * lo + 1] = ui + ti;
      buf[2 * hi] = ur - tr;
      buf[2 * hi + 1] = ui - ti;
    }
  }
  for (int i = 0; i < 256; i++) {
    data[2 * (size_t)i * stride] = buf[2 * i];
    data[2 * (size_t)i * stride + 1] = buf[2 * i + 1];
  }
}

void host_fft2d(std::vector<float>& c, const float* twr, const float* twi) {
  for (int r = 0; r < 256; r++) host_fft256(c.data() + 2 * (size_t)r * 256, 1, twr, twi, false);
  for (int col = 0; col < 256; col++) host_fft256(c.data() + 2 * (size_t)col, 256, twr, twi, false);
}
}  // namespace

void ccc_build_model(CccModel& m, int w, int h, const float* filter, const float* bias) {
  if (w != 256 || h != 256) throw std::invalid_argument("CCC model must be 256x256 (got " + std::to_string(w) + "x" + std::to_string(h) + ")");
  float twr[128], twi[128];
  fft256_twiddles(twr, twi);
  m.filter_t.assign(65536, 0.f);
  m.bias_t.assign(65536, 0.f);
  m.filter_fft.assign(131072, 0.f);
  m.bias_fft.assign(131072, 0.f);
  for (int y = 0; y < 256; y++)
    for (int x = 0; x < 256; x++) {
      // loadModel transposes both planes (convolutional_color_constancy.cpp:131-132)
      float f = filter[(size_t)x * 256 + y], b = bias[(size_t)x * 256 + y];
      m.filter_t[(size_t)y * 256 + x] = f;
      m.bias_t[(size_t)y * 256 + x] = b;
      m.filter_fft[2 * ((size_t)y * 256 + x)] = f;
      m.bias_fft[2 * ((size_t)y * 256 + x)] = b;
    }
  host_fft2d(m.filter_fft, twr, twi);
  host_fft2d(m.bias_fft, twr, twi);
  m.loaded = true;
}

bool ccc_load_model_file(CccModel& m, const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f.good()) return false;
  int32_t w = 0, h = 0;
  f.read((char*)&w, 4);
  f.read((char*)&h, 4);
  if (!f.good() || w <= 0 || h <= 0 || (int64_t)w * h > (1 << 20)) throw std::invalid_argument("CCC model file " + path + ": bad header");
  std::vector<float> filt((size_t)w * h), bias((size_t)w * h);
  f.read((char*)filt.data(), (std::streamsize)filt.size() * 4);
  f.read((char*)bias.data(), (std::streamsize)bias.size() * 4);
  if (!f.good()) throw std::invalid_argument("CCC model file " + path + ": truncated");
  ccc_build_model(m, w, h, filt.data(), bias.data());
  return true;
}

void ccc_build_scalar_tables(float log_tab[256], std::vector<float>& accum_tab, float exp_neg_tab[256]) {
  log_tab[0] = -INFINITY;
  for (int i = 1; i < 256; i++) log_tab[i] = std::log((float)i);
  const int n = 360 * 270;
  float num_pixels = (float)n;
  float weight = 1.0f / num_pixels;
  accum_tab.assign(n + 1, 0.f);
  float acc = 0.f;
  for (int i = 1; i <= n; i++) {
    acc += weight;
    accum_tab[i] = acc;
  }
  const float bin = 1.0f / 64.0f, uv0 = -1.421875f;
  for (int k = 0; k < 256; k++) {
    float L = k * bin + uv0;
    exp_neg_tab[k] = std::exp(-L);
  }
}

// ---------------------------------------------------------------------------------------------
// Debug stage dumps
// ---------------------------------------------------------------------------------------------
void normalize_minmax_u8(uint8_t* data, size_t n) {
  if (n == 0) return;
  int lo = 255, hi = 0;
  for (size_t i = 0; i < n; i++) {
    lo = std::min(lo, (int)data[i]);
    hi = std::max(hi, (int)data[i]);
  }
  const double smin = lo, smax = hi;
  const double scale = 255.0 * (smax - smin > 2.220446049250313e-16 ? 1.0 / (smax - smin) : 0.0);
  const double shift = 0.0 - smin * scale;
  const float a = (float)scale, b = (float)shift;
  uint8_t lut[256];
  for (int v = 0; v < 256; v++) {
    const float r = (float)v * a + b;  // multiply, then add (the build runs with -ffp-contract=off)
    long q = std::lrintf(r);
    lut[v] = (uint8_t)(q < 0 ? 0 : q > 255 ? 255 : q);
  }
  for (size_t i = 0; i < n; i++) data[i] = lut[data[i]];
}

namespace {
struct Crc32 {
  uint32_t table[256];
  Crc32() {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
  }
  uint32_t update(uint32_t crc, const uint8_t* p, size_t n) const {
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
  }
};
void put_be32(std::vector<uint8_t>& v, uint32_t x) {
  for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s));
}
bool write_chunk(std::FILE* f, const Crc32& crc, const char type[4], const std::vector<uint8_t>& body) {
  std::vector<uint8_t> head;
  put_be32(head, (uint32_t)body.size());
  head.insert(head.end(), type, type + 4);
  uint32_t c = crc.update(0xFFFFFFFFu, head.data() + 4, 4);
  c = crc.update(c, body.data(), body.size()) ^ 0xFFFFFFFFu;
  std::vector<uint8_t> tail;
  put_be32(tail, c);
  return std::fwrite(head.data(), 1, head.size(), f) == head.size() &&
         (body.empty() || std::fwrite(body.data(), 1, body.size(), f) == body.size()) && std::fwrite(tail.data(), 1, 4, f) == 4;
}
}  // namespace

bool write_png(const std::string& path, const uint8_t* data, int rows, int cols, int channels) {
  if (!data || rows < 1 || cols < 1 || (channels != 1 && channels != 3)) return false;
  static const Crc32 crc;
  // raw scanlines: filter byte 0 + pixels (RGB order)
  const size_t row_bytes = (size_t)cols * channels + 1;
  std::vector<uint8_t> raw(row_bytes * rows);
  for (int y = 0; y < rows; y++) {
    uint8_t* d = raw.data() + row_bytes * y;
    const uint8_t* s = data + (size_t)y * cols * channels;
    *d++ = 0;
    if (channels == 1) {
      std::memcpy(d, s, (size_t)cols);
    } else {
      for (int x = 0; x < cols; x++) {
        d[3 * x] = s[3 * x + 2];
        d[3 * x + 1] = s[3 * x + 1];
        d[3 * x + 2] = s[3 * x];
      }
    }
  }
  // zlib stream of stored deflate blocks
  std::vector<uint8_t> z;
  z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
  z.push_back(0x78);
  z.push_back(0x01);
  uint32_t s1 = 1, s2 = 0;
  size_t pos = 0;
  while (pos < raw.size()) {
    const size_t len = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + len == raw.size() ? 1 : 0);
    z.push_back((uint8_t)(len & 0xFF));
    z.push_back((uint8_t)(len >> 8));
    z.push_back((uint8_t)(~len & 0xFF));
    z.push_back((uint8_t)((~len >> 8) & 0xFF));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + len);
    for (size_t i = 0; i < len;) {  // adler32, 5552 bytes between reductions
      const size_t run = std::min<size_t>(5552, len - i);
      for (size_t k = 0; k < run; k++) {
        s1 += raw[pos + i + k];
        s2 += s1;
      }
      s1 %= 65521u;
      s2 %= 65521u;
      i += run;
    }
    pos += len;
  }
  put_be32(z, (s2 << 16) | s1);
  std::FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, (uint32_t)cols);
  put_be32(ihdr, (uint32_t)rows);
  ihdr.push_back(8);                              // bit depth
  ihdr.push_back(channels == 3 ? 2 : 0);          // colour type: truecolour / greyscale
  ihdr.push_back(0);
  ihdr.push_back(0);
  ihdr.push_back(0);
  bool ok = std::fwrite(sig, 1, 8, f) == 8 && write_chunk(f, crc, "IHDR", ihdr) && write_chunk(f, crc, "IDAT", z) &&
            write_chunk(f, crc, "IEND", {});
  ok = (std::fclose(f) == 0) && ok;
  return ok;
}


}  // namespace rip
