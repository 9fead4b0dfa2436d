// ONNX weight importer -- see onnx_import.h.  Host-only code (no HIP calls).
#include "onnx_import.h"

#include <cmath>
#include <cstring>
#include <map>

#include "common.h"

namespace rt {
namespace {

// ---- protobuf wire format ------------------------------------------------------------------
struct Span { const uint8_t* p = nullptr; size_t n = 0; };

struct Reader {
  const uint8_t* p; const uint8_t* end;
  explicit Reader(Span s) : p(s.p), end(s.p + s.n) {}
  bool done() const { return p >= end; }
  uint64_t varint() {
    uint64_t v = 0; int shift = 0;
    while (true) {
      if (p >= end || shift > 63) throw RtError(4, "ONNX: truncated varint");
      uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
      shift += 7;
    }
  }
  // returns field number, sets wire type; for length-delimited fields `s` is the payload
  int next(int* wt, uint64_t* val, Span* s) {
    uint64_t key = varint();
    *wt = (int)(key & 7);
    switch (*wt) {
      case 0: *val = varint(); break;
      case 1: if (end - p < 8) throw RtError(4, "ONNX: truncated fixed64"); memcpy(val, p, 8); p += 8; break;
      case 2: { uint64_t len = varint(); if ((uint64_t)(end - p) < len) throw RtError(4, "ONNX: truncated field"); s->p = p; s->n = (size_t)len; p += len; break; }
      case 5: { if (end - p < 4) throw RtError(4, "ONNX: truncated fixed32"); uint32_t v; memcpy(&v, p, 4); *val = v; p += 4; break; }
      default: throw RtError(4, "ONNX: unsupported protobuf wire type");
    }
    return (int)(key >> 3);
  }
};

static std::string str(Span s) { return std::string((const char*)s.p, s.n); }

// repeated int64, packed or not
static void read_ints(int wt, uint64_t val, Span s, std::vector<long long>* out) {
  if (wt == 0) { out->push_back((long long)val); return; }
  Reader r(s);
  while (!r.done()) out->push_back((long long)r.varint());
}

struct Tensor {
  std::string name;
  std::vector<long long> dims;
  int dtype = 0;  // 1 = float
  std::vector<float> f;
  size_t numel() const { size_t n = 1; for (long long d : dims) n *= (size_t)d; return n; }
};

static Tensor parse_tensor(Span sp) {
  Tensor t; Span raw; bool has_raw = false;
  Reader r(sp);
  while (!r.done()) {
    int wt; uint64_t v = 0; Span s;
    int f = r.next(&wt, &v, &s);
    if (f == 1) read_ints(wt, v, s, &t.dims);
    else if (f == 2) t.dtype = (int)v;
    else if (f == 4) {  // float_data
      if (wt == 5) { float x; uint32_t u = (uint32_t)v; memcpy(&x, &u, 4); t.f.push_back(x); }
      else { size_t n = s.n / 4; size_t o = t.f.size(); t.f.resize(o + n); if (n) memcpy(t.f.data() + o, s.p, n * 4); }
    } else if (f == 8) t.name = str(s);
    else if (f == 9) { raw = s; has_raw = true; }
  }
  if (t.dtype == 1 && has_raw) { t.f.resize(raw.n / 4); if (!t.f.empty()) memcpy(t.f.data(), raw.p, t.f.size() * 4); }
  if (t.dtype == 1 && t.f.size() != t.numel()) throw RtError(4, "ONNX: tensor " + t.name + " has " + std::to_string(t.f.size()) + " values for its dims");
  return t;
}

struct Node {
  std::string op, name;
  std::vector<std::string> in, out;
  std::map<std::string, std::vector<long long>> ints;
  std::map<std::string, float> floats;
  bool has_value = false; Tensor value;  // Constant
};

static Node parse_node(Span sp) {
  Node n;
  Reader r(sp);
  while (!r.done()) {
    int wt; uint64_t v = 0; Span s;
    int f = r.next(&wt, &v, &s);
    if (f == 1) n.in.push_back(str(s));
    else if (f == 2) n.out.push_back(str(s));
    else if (f == 3) n.name = str(s);
    else if (f == 4) n.op = str(s);
    else if (f == 5) {  // AttributeProto
      Reader a(s); std::string an; std::vector<long long> iv; bool has_f = false; float fv = 0.f;
      while (!a.done()) {
        int awt; uint64_t av = 0; Span as;
        int af = a.next(&awt, &av, &as);
        if (af == 1) an = str(as);
        else if (af == 2) { uint32_t u = (uint32_t)av; memcpy(&fv, &u, 4); has_f = true; }
        else if (af == 3) iv.push_back((long long)av);
        else if (af == 8) read_ints(awt, av, as, &iv);
        else if (af == 5) { n.value = parse_tensor(as); n.has_value = true; }
      }
      if (!iv.empty()) n.ints[an] = iv;
      if (has_f) n.floats[an] = fv;
    }
  }
  return n;
}

// ---- parameter events ----------------------------------------------------------------------
enum EvKind { EV_CONV, EV_DECONV, EV_MATMUL, EV_BN, EV_LAB, EV_LN, EV_VADD };
struct Event {
  EvKind kind;
  std::string in, out, where;      // data-flow names, node name for messages
  std::vector<long long> dims;     // weight dims (conv/deconv/matmul as [in,out]), [C] for the others
  std::vector<float> w, b;         // weights / bias (conv...), gamma / beta (LN), a / c (LAB), scale... (BN folded form)
  bool has_b = false;
  bool used = false;
};

struct Graph {
  std::map<std::string, Tensor> init;
  std::vector<Node> nodes;
};

static Graph parse_model(const uint8_t* data, size_t len) {
  Graph g; Span graph; bool found = false;
  Reader r(Span{data, len});
  while (!r.done()) {
    int wt; uint64_t v = 0; Span s;
    int f = r.next(&wt, &v, &s);
    if (f == 7 && wt == 2) { graph = s; found = true; }
  }
  if (!found) throw RtError(4, "ONNX: no graph in the model");
  Reader gr(graph);
  while (!gr.done()) {
    int wt; uint64_t v = 0; Span s;
    int f = gr.next(&wt, &v, &s);
    if (f == 1 && wt == 2) g.nodes.push_back(parse_node(s));
    else if (f == 5 && wt == 2) { Tensor t = parse_tensor(s); g.init[t.name] = std::move(t); }
  }
  for (Node& n : g.nodes)  // Paddle2ONNX emits many parameters as Constant nodes
    if (n.op == "Constant" && n.has_value && !n.out.empty()) { n.value.name = n.out[0]; g.init[n.out[0]] = n.value; }
  return g;
}

static const Tensor* fparam(const Graph& g, const std::string& name) {
  auto it = g.init.find(name);
  return (it != g.init.end() && it->second.dtype == 1) ? &it->second : nullptr;
}

static std::vector<Event> extract_events(const Graph& g) {
  std::vector<Event> ev;
  // Mul(x, param) whose result feeds Add(., param) of the same length = scalar affine (LAB) or LayerNorm tail
  std::map<std::string, size_t> pending_mul;  // output name -> index in `muls`
  struct Mul { std::string in, out, where; const Tensor* t; };
  std::vector<Mul> muls;
  for (const Node& n : g.nodes) {
    if (n.in.empty() || n.out.empty()) continue;  // malformed or parameter-free node: nothing to take
    if (n.op == "Conv" || n.op == "ConvTranspose") {
      const Tensor* w = n.in.size() > 1 ? fparam(g, n.in[1]) : nullptr;
      if (!w || w->dims.size() != 4) continue;
      Event e; e.kind = n.op == "Conv" ? EV_CONV : EV_DECONV; e.in = n.in[0]; e.out = n.out[0]; e.where = n.op + " " + n.name;
      e.dims = w->dims; e.w = w->f;
      if (n.in.size() > 2 && !n.in[2].empty()) { const Tensor* b = fparam(g, n.in[2]); if (b) { e.b = b->f; e.has_b = true; } }
      ev.push_back(std::move(e));
    } else if (n.op == "BatchNormalization" && n.in.size() >= 5) {
      const Tensor *sc = fparam(g, n.in[1]), *bi = fparam(g, n.in[2]), *mu = fparam(g, n.in[3]), *va = fparam(g, n.in[4]);
      if (!sc || !bi || !mu || !va) continue;
      if (bi->f.size() != sc->f.size() || mu->f.size() != sc->f.size() || va->f.size() != sc->f.size()) throw RtError(4, "ONNX: BatchNormalization " + n.name + " has parameters of different lengths");
      const float eps = n.floats.count("epsilon") ? n.floats.at("epsilon") : 1e-5f;
      Event e; e.kind = EV_BN; e.in = n.in[0]; e.out = n.out[0]; e.where = "BatchNormalization " + n.name;
      const size_t C = sc->f.size(); e.dims = {(long long)C}; e.w.resize(C); e.b.resize(C); e.has_b = true;
      for (size_t c = 0; c < C; c++) {  // y = w * x + b
        const double s = (double)sc->f[c] / std::sqrt((double)va->f[c] + (double)eps);
        e.w[c] = (float)s; e.b[c] = (float)((double)bi->f[c] - (double)mu->f[c] * s);
      }
      ev.push_back(std::move(e));
    } else if (n.op == "MatMul" || n.op == "Gemm") {
      const Tensor* w = n.in.size() > 1 ? fparam(g, n.in[1]) : nullptr;
      if (!w || w->dims.size() != 2) continue;
      Event e; e.kind = EV_MATMUL; e.in = n.in[0]; e.out = n.out[0]; e.where = n.op + " " + n.name;
      const bool tb = n.op == "Gemm" && n.ints.count("transB") && !n.ints.at("transB").empty() && n.ints.at("transB")[0] != 0;
      const long long K = tb ? w->dims[1] : w->dims[0], N = tb ? w->dims[0] : w->dims[1];
      if (K < 0 || N < 0) throw RtError(4, "ONNX: " + n.op + " " + n.name + " has a negative weight dimension");
      e.dims = {K, N}; e.w.resize((size_t)(K * N));
      for (long long k = 0; k < K; k++)
        for (long long j = 0; j < N; j++) e.w[(size_t)(k * N + j)] = tb ? w->f[(size_t)(j * K + k)] : w->f[(size_t)(k * N + j)];
      if (n.op == "Gemm" && n.in.size() > 2) { const Tensor* b = fparam(g, n.in[2]); if (b) { e.b = b->f; e.has_b = true; } }
      ev.push_back(std::move(e));
    } else if (n.op == "LayerNormalization" && n.in.size() >= 3) {
      const Tensor *gm = fparam(g, n.in[1]), *bt = fparam(g, n.in[2]);
      if (!gm || !bt) continue;
      Event e; e.kind = EV_LN; e.in = n.in[0]; e.out = n.out[0]; e.where = "LayerNormalization " + n.name;
      e.dims = {(long long)gm->f.size()}; e.w = gm->f; e.b = bt->f; e.has_b = true;
      ev.push_back(std::move(e));
    } else if ((n.op == "Mul" || n.op == "Add") && n.in.size() == 2) {
      const Tensor *p0 = fparam(g, n.in[0]), *p1 = fparam(g, n.in[1]);
      if ((p0 != nullptr) == (p1 != nullptr)) continue;  // no parameter, or constant folding leftovers
      const Tensor* p = p0 ? p0 : p1;
      const std::string& x = p0 ? n.in[1] : n.in[0];
      if (n.op == "Mul") {
        pending_mul[n.out[0]] = muls.size();
        muls.push_back({x, n.out[0], "Mul " + n.name, p});
      } else {
        auto it = pending_mul.find(x);
        if (it != pending_mul.end() && muls[it->second].t->f.size() == p->f.size()) {
          const Mul& m = muls[it->second];
          Event e; e.kind = p->f.size() == 1 ? EV_LAB : EV_LN; e.in = m.in; e.out = n.out[0]; e.where = m.where + " + Add " + n.name;
          e.dims = {(long long)p->f.size()}; e.w = m.t->f; e.b = p->f; e.has_b = true;
          ev.push_back(std::move(e));
        } else {  // bias of a MatMul / un-fused conv bias ([C], [1,C,1,1], ...); only used when chained to one
          Event e; e.kind = EV_VADD; e.in = x; e.out = n.out[0]; e.where = "Add " + n.name;
          e.dims = {(long long)p->f.size()}; e.b = p->f; e.has_b = true;
          ev.push_back(std::move(e));
        }
      }
    }
  }
  return ev;
}

// ---- RTWB writer (format: retto_amd/synth.py::pack_blob) ---------------------------------------
struct OutTensor { std::string name; std::vector<int> dims; std::vector<float> data; };

static std::vector<uint8_t> write_rtwb(const std::vector<OutTensor>& ts) {
  std::vector<uint8_t> table;
  std::vector<uint64_t> offs;
  uint64_t off = 0;
  for (const OutTensor& t : ts) { offs.push_back(off); off = (off + t.data.size() * 4 + 63) / 64 * 64; }
  auto put = [&](const void* p, size_t n) { table.insert(table.end(), (const uint8_t*)p, (const uint8_t*)p + n); };
  for (size_t i = 0; i < ts.size(); i++) {
    const OutTensor& t = ts[i];
    uint16_t ln = (uint16_t)t.name.size(); put(&ln, 2); put(t.name.data(), ln);
    uint8_t hdr[4] = {(uint8_t)t.dims.size(), 0, 0, 0}; put(hdr, 4);
    for (int d : t.dims) { uint32_t v = (uint32_t)d; put(&v, 4); }
    uint64_t nb = t.data.size() * 4; put(&offs[i], 8); put(&nb, 8);
  }
  std::vector<uint8_t> out;
  out.insert(out.end(), {'R', 'T', 'W', 'B'});
  uint32_t head[3] = {1, (uint32_t)ts.size(), 0};
  out.insert(out.end(), (const uint8_t*)head, (const uint8_t*)head + 12);
  out.insert(out.end(), table.begin(), table.end());
  out.resize((out.size() + 63) / 64 * 64, 0);
  const size_t base = out.size();
  out.resize(base + (size_t)off, 0);
  for (size_t i = 0; i < ts.size(); i++) memcpy(out.data() + base + offs[i], ts[i].data.data(), ts[i].data.size() * 4);
  return out;
}

static std::string dims_str(const std::vector<long long>& d) {
  std::string s = "[";
  for (size_t i = 0; i < d.size(); i++) s += (i ? "," : "") + std::to_string(d[i]);
  return s + "]";
}

}  // namespace

bool looks_like_rtwb(const std::vector<uint8_t>& bytes) { return bytes.size() >= 4 && memcmp(bytes.data(), "RTWB", 4) == 0; }

std::vector<uint8_t> onnx_to_rtwb(int which, const uint8_t* data, size_t len) {
  const Graph g = parse_model(data, len);
  std::vector<Event> ev = extract_events(g);
  // the server graphs come through the same det / rec sources: told apart by the stem (PP-LCNetV3: 3 -> 16, PPHGNet_small: 3 -> 64)
  if (which == MODEL_DET || which == MODEL_REC)
    for (const Event& e : ev)
      if (e.kind == EV_CONV) {
        if (e.dims.size() == 4 && e.dims[0] == 64 && e.dims[1] == 3) which = which == MODEL_DET ? MODEL_SDET : MODEL_SREC;
        break;
      }
  const std::vector<ManifestEntry> mf = model_manifest(which);
  std::vector<OutTensor> out;

  // first unused event of `kind` whose weight dims match `want` (-1 = any)
  auto find = [&](EvKind kind, const std::vector<int>& want, const std::string& what) -> Event& {
    for (Event& e : ev) {
      if (e.used || e.kind != kind || e.dims.size() != want.size()) continue;
      bool ok = true;
      for (size_t i = 0; i < want.size(); i++) ok = ok && (want[i] < 0 || e.dims[i] == want[i]);
      if (ok) { e.used = true; return e; }
    }
    std::string have;
    for (const Event& e : ev) if (!e.used && e.kind == kind) { have += " " + dims_str(e.dims); if (have.size() > 200) break; }
    throw RtError(4, "ONNX import: no parameter node left for " + what + "; unmatched of that kind:" + (have.empty() ? " none" : have));
  };
  // y = s[c] * y + t[c] folded into (w, b); per-output-channel stride `inner` elements, `cout` channels
  auto fold = [&](std::vector<float>& w, std::vector<float>& b, int cout, bool deconv, int cin_dim, const std::vector<float>& s,
                  const std::vector<float>& t) {
    const size_t per = w.size() / (size_t)cout;  // conv: [cout][...]; deconv: [cin][cout][kh*kw]
    for (int c = 0; c < cout; c++) {
      const float sc = s.size() == 1 ? s[0] : s[(size_t)c], sh = t.empty() ? 0.f : (t.size() == 1 ? t[0] : t[(size_t)c]);
      if (!deconv) {
        for (size_t i = 0; i < per; i++) w[(size_t)c * per + i] *= sc;
      } else {
        const size_t khw = w.size() / ((size_t)cin_dim * cout);
        for (int ci = 0; ci < cin_dim; ci++)
          for (size_t i = 0; i < khw; i++) w[((size_t)ci * cout + c) * khw + i] *= sc;
      }
      b[(size_t)c] = b[(size_t)c] * sc + sh;
    }
  };
  // absorb everything chained to `e`'s output: BatchNorm, pre-activation LAB, separate bias Add
  auto absorb = [&](Event& e, std::vector<float>& w, std::vector<float>& b, int cout, bool deconv, int cin_dim, bool bias_only) {
    std::string cur = e.out;
    bool again = true;
    while (again) {
      again = false;
      for (Event& f : ev) {
        if (f.used || f.in != cur) continue;
        if (!bias_only && f.kind == EV_BN && (int)f.dims[0] == cout) fold(w, b, cout, deconv, cin_dim, f.w, f.b);
        else if (!bias_only && f.kind == EV_LAB) fold(w, b, cout, deconv, cin_dim, f.w, f.b);
        else if (f.kind == EV_VADD && (int)f.dims[0] == cout) { for (int c = 0; c < cout; c++) b[(size_t)c] += f.b[(size_t)c]; }
        else continue;
        f.used = true; cur = f.out; again = true;
        break;
      }
    }
  };

  for (size_t i = 0; i < mf.size(); i++) {
    const ManifestEntry& m = mf[i];
    const std::string base = m.name.substr(0, m.name.rfind('.')), leaf = m.name.substr(m.name.rfind('.') + 1);
    if (leaf == "w") {
      const bool want_b = i + 1 < mf.size() && mf[i + 1].name == base + ".b";
      const bool deconv = base.find("deconv") != std::string::npos;
      OutTensor tw, tb;
      if (m.dims.size() == 4) {
        Event& e = find(deconv ? EV_DECONV : EV_CONV, m.dims, m.name);
        const int cout = deconv ? (int)e.dims[1] : (int)e.dims[0];
        std::vector<float> w = e.w, b = e.has_b ? e.b : std::vector<float>((size_t)cout, 0.f);
        if ((int)b.size() != cout) throw RtError(4, "ONNX import: bias length of " + e.where + " does not match " + m.name);
        absorb(e, w, b, cout, deconv, (int)e.dims[0], false);
        tw = {m.name, m.dims, w};
        bool nz = false; for (float v : b) nz = nz || v != 0.f;
        if (!want_b && nz) throw RtError(4, "ONNX import: " + e.where + " carries a bias but " + base + " has none in this architecture");
        if (want_b) tb = {base + ".b", {cout}, b};
      } else {
        Event& e = find(EV_MATMUL, m.dims, m.name);
        const int cout = (int)e.dims[1];
        std::vector<float> w = e.w, b = e.has_b ? e.b : std::vector<float>((size_t)cout, 0.f);
        absorb(e, w, b, cout, false, 0, true);  // MatMul + Add bias (w is [in][out]; BN / LAB never follow a linear here)
        tw = {m.name, {(int)e.dims[0], cout}, e.w};
        tb = {base + ".b", {cout}, b};
      }
      out.push_back(tw);
      if (want_b) { out.push_back(tb); i++; }
    } else if (leaf == "a") {  // LearnableAffineBlock after the activation: .a, .c
      Event& e = find(EV_LAB, {1}, m.name);
      out.push_back({base + ".a", {1}, e.w});
      out.push_back({base + ".c", {1}, e.b});
      i++;
    } else if (leaf == "g") {  // LayerNorm: .g, .beta
      Event& e = find(EV_LN, m.dims, m.name);
      out.push_back({base + ".g", m.dims, e.w});
      out.push_back({base + ".beta", m.dims, e.b});
      i++;
    } else {
      throw RtError(4, "ONNX import: manifest entry out of place: " + m.name);
    }
  }
  return write_rtwb(out);
}

}  // namespace rt
