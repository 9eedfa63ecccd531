// AHIP model-file reader: bare blob or member `*/extra/allegro_hip.bin` of a TorchScript zip.
// Replaces torch::jit::load(path, device, metadata) of the reference
// (/root/reference/pair_nequip_allegro.cpp:214-222): same file, same five metadata keys,
// but the weights come from the blob instead of a pickled module.
#include "model_io.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace ahip {

static const char MAGIC[] = "AHIPMDL1";

const HostTensor &HostModel::get(const std::string &name) const {
  auto it = tensors.find(name);
  if (it == tensors.end()) throw std::runtime_error("model file: missing tensor '" + name + "'");
  return it->second;
}

HostModel pad_host_model(const HostModel &h, int SF, int UF, int WF, int RF) {
  const int S = h.S, U = h.U, W = h.mlp_width, R = h.readout_width, L = h.l_max, T = h.num_types, B = h.num_bessels, dep = h.mlp_depth;
  if (S > SF || U > UF || W > WF || R > RF) throw std::runtime_error("pad_host_model: the model is wider than the target shape");
  HostModel p = h;
  p.S = SF; p.U = UF; p.mlp_width = WF; p.readout_width = RF;
  // out[rmap(r)][cmap(c)] = in[r][c] for every leading block of a [..][R0][C0] tensor
  auto pad = [&](const std::string &name, int R0, int C0, int R1, int C1, auto rmap, auto cmap) {
    const HostTensor &t = h.get(name);
    const long long blk = (long long)R0 * C0, nb = blk > 0 ? t.numel() / blk : 0;
    HostTensor o;
    o.shape = t.shape;
    o.shape[o.shape.size() - 2] = R1; o.shape[o.shape.size() - 1] = C1;
    o.data.assign((size_t)nb * R1 * C1, 0.0);
    for (long long b = 0; b < nb; ++b)
      for (int r = 0; r < R0; ++r)
        for (int c = 0; c < C0; ++c) o.data[(size_t)(b * R1 + rmap(r)) * C1 + cmap(c)] = t.data[(size_t)(b * R0 + r) * C0 + c];
    p.tensors[name] = o;
  };
  auto id = [](int i) { return i; };
  auto lu = [&](int c) { return (c / U) * UF + (c % U); };          // (l, u) column of an embedding / environment weight vector
  auto cat = [&](int r) { return r < S ? r : SF + (r - S); };        // row of the latent MLP's input [x, scalars]
  auto mlp = [&](const std::string &pre, int din, int din1, int dout, int dout1, auto rmap0) {
    if (dep == 0) { pad(pre + ".w0", din, dout, din1, dout1, rmap0, id); return; }
    pad(pre + ".w0", din, W, din1, WF, rmap0, id);
    for (int k = 1; k < dep; ++k) pad(pre + ".w" + std::to_string(k), W, W, WF, WF, id, id);
    pad(pre + ".w" + std::to_string(dep), W, dout, WF, dout1, id, id);
  };
  mlp("tb", 2 * T + B, 2 * T + B, S, SF, id);
  pad("emb.w", S, U * (L + 1), SF, UF * (L + 1), id, lu);
  for (int k = 1; k <= h.num_layers; ++k) {
    const std::string lk = "l" + std::to_string(k);
    pad(lk + ".env", S, U * (L + 1), SF, UF * (L + 1), id, lu);
    { const HostTensor &tp = h.get(lk + ".tp"); pad(lk + ".tp", tp.shape[0], U, tp.shape[0], UF, id, id); }
    mlp(lk + ".lat", S + U, SF + UF, S, SF, cat);
    if (k < h.num_layers) pad(lk + ".mix", U, U, UF, UF, id, id);
  }
  // read-out MLP S -> R (x readout_depth) -> 1
  if (h.readout_depth == 0) pad("out.w0", S, 1, SF, 1, id, id);
  else {
    pad("out.w0", S, R, SF, RF, id, id);
    for (int k = 1; k < h.readout_depth; ++k) pad("out.w" + std::to_string(k), R, R, RF, RF, id, id);
    pad("out.w" + std::to_string(h.readout_depth), R, 1, RF, 1, id, id);
  }
  return p;
}

static std::vector<unsigned char> read_all(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open model file " + path);
  f.seekg(0, std::ios::end);
  std::streamoff n = f.tellg();
  f.seekg(0);
  std::vector<unsigned char> buf((size_t)n);
  if (n > 0) f.read((char *)buf.data(), n);
  if (!f) throw std::runtime_error("short read on model file " + path);
  return buf;
}

static uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t rd32(const unsigned char *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t rd64(const unsigned char *p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

// Minimal ZIP central-directory walk (STORED members only, ZIP64 aware).
static bool zip_find_member(const std::vector<unsigned char> &z, const std::string &suffix,
                            size_t &off, size_t &len, std::string &err) {
  const size_t n = z.size();
  if (n < 22) { err = "file too small to be a zip archive"; return false; }
  size_t eocd = std::string::npos;
  const size_t lo = n > 22 + 65535 ? n - 22 - 65535 : 0;
  for (size_t i = n - 22 + 1; i-- > lo;) {
    if (rd32(&z[i]) == 0x06054b50u) { eocd = i; break; }
  }
  if (eocd == std::string::npos) { err = "not a zip archive (no end-of-central-directory)"; return false; }
  uint64_t cd_off = rd32(&z[eocd + 16]), cd_entries = rd16(&z[eocd + 10]);
  if (cd_off == 0xFFFFFFFFu || cd_entries == 0xFFFF) {          // ZIP64
    if (eocd < 20 || rd32(&z[eocd - 20]) != 0x07064b50u) { err = "broken zip64 locator"; return false; }
    uint64_t e64 = rd64(&z[eocd - 20 + 8]);
    if (e64 > n || n - e64 < 56 || rd32(&z[e64]) != 0x06064b50u) { err = "broken zip64 record"; return false; }
    cd_entries = rd64(&z[e64 + 32]);
    cd_off = rd64(&z[e64 + 48]);
  }
  if (cd_off > n) { err = "broken zip central directory offset"; return false; }
  size_t p = (size_t)cd_off;
  for (uint64_t k = 0; k < cd_entries; ++k) {
    if (n - p < 46 || rd32(&z[p]) != 0x02014b50u) { err = "broken zip central directory"; return false; }
    uint16_t method = rd16(&z[p + 10]);
    uint64_t csize = rd32(&z[p + 20]), usize = rd32(&z[p + 24]);
    uint16_t nlen = rd16(&z[p + 28]), xlen = rd16(&z[p + 30]), clen = rd16(&z[p + 32]);
    uint64_t lho = rd32(&z[p + 42]);
    if (n - p - 46 < (size_t)nlen + xlen + clen) { err = "zip central directory entry runs past end of file"; return false; }
    std::string name((const char *)&z[p + 46], nlen);
    // zip64 extra field
    size_t xp = p + 46 + nlen, xe = xp + xlen;
    while (xp + 4 <= xe) {
      uint16_t id = rd16(&z[xp]), sz = rd16(&z[xp + 2]);
      if (xp + 4 + sz > xe) { err = "broken zip extra field"; return false; }
      if (id == 0x0001) {
        size_t q = xp + 4;
        const size_t qe = xp + 4 + sz;
        auto take = [&](uint64_t &v) { if (q + 8 > qe) return false; v = rd64(&z[q]); q += 8; return true; };
        if ((usize == 0xFFFFFFFFu && !take(usize)) || (csize == 0xFFFFFFFFu && !take(csize)) || (lho == 0xFFFFFFFFu && !take(lho))) {
          err = "broken zip64 extra field"; return false;
        }
      }
      xp += 4 + sz;
    }
    if (name.size() >= suffix.size() && name.compare(name.size() - suffix.size(), suffix.size(), suffix) == 0) {
      if (method != 0) { err = "zip member " + name + " is compressed; expected STORED"; return false; }
      if (lho > n || n - lho < 30 || rd32(&z[lho]) != 0x04034b50u) { err = "broken zip local header"; return false; }
      uint16_t lnlen = rd16(&z[lho + 26]), lxlen = rd16(&z[lho + 28]);
      if (n - lho - 30 < (size_t)lnlen + lxlen) { err = "zip local header runs past end of file"; return false; }
      off = (size_t)lho + 30 + lnlen + lxlen;
      if (usize > n - off) { err = "zip member runs past end of file"; return false; }
      len = (size_t)usize;
      return true;
    }
    p += 46 + nlen + xlen + clen;
  }
  err = "archive has no */extra/allegro_hip.bin member";
  return false;
}

HostModel parse_blob(const unsigned char *p, size_t n, const std::string &origin) {
  if (n < 18 || std::memcmp(p, MAGIC, 8) != 0) throw std::runtime_error(origin + ": bad AHIP magic");
  size_t header_bytes = (size_t)std::strtoull(std::string((const char *)p + 9, 8).c_str(), nullptr, 10);
  if (header_bytes < 18 || header_bytes > n) throw std::runtime_error(origin + ": bad AHIP header size");
  std::string text((const char *)p + 18, header_bytes - 18);
  text = text.substr(0, text.find('\0'));
  std::istringstream in(text);
  std::string line;
  HostModel m;
  struct Dir { std::string name; std::vector<int> shape; size_t off; };
  std::vector<Dir> dir;
  bool ended = false;
  while (std::getline(in, line)) {
    std::istringstream ls(line);
    std::string key;
    if (!(ls >> key)) continue;
    if (key == "end") { ended = true; break; }
    if (key == "tensor") {
      Dir d; int nd = 0;
      ls >> d.name >> nd;
      if (nd < 0 || nd > 8) throw std::runtime_error(origin + ": bad tensor rank: " + line);
      for (int k = 0; k < nd; ++k) {
        int s = -1; ls >> s;
        if (!ls || s < 0) throw std::runtime_error(origin + ": bad tensor dimension: " + line);
        d.shape.push_back(s);
      }
      ls >> d.off;
      if (!ls) throw std::runtime_error(origin + ": bad tensor line: " + line);
      dir.push_back(d);
    } else if (key == "type_names") {
      std::string t;
      while (ls >> t) {
        if (!m.type_names_joined.empty()) m.type_names_joined += " ";
        m.type_names_joined += t;
        m.type_names.push_back(t);
      }
    } else if (key == "per_edge_type_cutoff") {
      double v;
      while (ls >> v) m.per_edge_type_cutoff.push_back(v);
    } else if (key == "model_dtype") ls >> m.model_dtype;
    else if (key == "r_max") ls >> m.r_max;
    else if (key == "avg_num_neighbors") ls >> m.avg_num_neighbors;
    else if (key == "num_types") ls >> m.num_types;
    else if (key == "num_bessels") ls >> m.num_bessels;
    else if (key == "poly_p") ls >> m.poly_p;
    else if (key == "l_max") ls >> m.l_max;
    else if (key == "num_layers") ls >> m.num_layers;
    else if (key == "num_scalar_features") ls >> m.S;
    else if (key == "num_tensor_features") ls >> m.U;
    else if (key == "mlp_depth") ls >> m.mlp_depth;
    else if (key == "mlp_width") ls >> m.mlp_width;
    else if (key == "readout_depth") ls >> m.readout_depth;
    else if (key == "readout_width") ls >> m.readout_width;
    else if (key == "seed") ls >> m.seed;
    else if (key == "allow_tf32") ls >> m.allow_tf32;
    else if (key == "version") { int v; ls >> v; if (v != 1) throw std::runtime_error(origin + ": unsupported AHIP version"); }
    // unknown keys are ignored (forward compatibility)
  }
  if (!ended) throw std::runtime_error(origin + ": AHIP header not terminated");
  if ((int)m.type_names.size() != m.num_types || m.num_types <= 0)
    throw std::runtime_error(origin + ": num_types does not match type_names");
  if (!m.per_edge_type_cutoff.empty() && (int)m.per_edge_type_cutoff.size() != m.num_types * m.num_types)
    throw std::runtime_error(origin + ": per_edge_type_cutoff must have num_types^2 entries");
  if (m.model_dtype != "float32" && m.model_dtype != "float64")
    throw std::runtime_error(origin + ": model_dtype must be float32 or float64");
  for (const Dir &d : dir) {
    HostTensor t;
    t.shape = d.shape;
    const size_t avail = n - header_bytes;                     // header_bytes <= n checked above
    size_t cnt = 1;
    for (int s : t.shape) {                                    // overflow-safe element count
      if (s != 0 && cnt > (avail / 8) / (size_t)s + 1) throw std::runtime_error(origin + ": tensor " + d.name + " out of bounds");
      cnt *= (size_t)s;
    }
    if (d.off > avail || cnt > (avail - d.off) / 8) throw std::runtime_error(origin + ": tensor " + d.name + " out of bounds");
    t.data.resize(cnt);
    std::memcpy(t.data.data(), p + header_bytes + d.off, cnt * 8);   // little-endian f64 host assumed
    m.tensors[d.name] = std::move(t);
  }
  return m;
}

static bool ends_with(const std::string &s, const std::string &suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

HostModel load_model_file(const std::string &path) {
  // Same extension gate as the reference (pair_nequip_allegro.cpp:197-206), plus the bare blob.
  const bool is_pth = ends_with(path, ".nequip.pth");
  const bool is_pt2 = ends_with(path, ".nequip.pt2");
  const bool is_ahip = ends_with(path, ".ahip");
  if (!is_pth && !is_pt2 && !is_ahip)
    throw std::runtime_error("Only accepts model paths with extension `.nequip.pth` or `.nequip.pt2` (or a bare `.ahip` blob), but found " + path);
  std::vector<unsigned char> buf = read_all(path);
  if (buf.size() >= 8 && std::memcmp(buf.data(), MAGIC, 8) == 0) return parse_blob(buf.data(), buf.size(), path);
  size_t off = 0, len = 0;
  std::string err;
  if (!zip_find_member(buf, "extra/allegro_hip.bin", off, len, err))
    throw std::runtime_error(path + ": " + err +
                             " -- this model file carries no allegro-hip weight section; add one with "
                             "`python -m pair_allegro_amd.tools.convert_nequip <in>.nequip.pth <out>.nequip.pth` (INTEGRATION.md section 2)");
  HostModel m = parse_blob(buf.data() + off, len, path);
  // the archive's own `extra/allow_tf32` member ("0" / "1"), the key the reference reads (pair_nequip_allegro.cpp:214-220, 267-270)
  size_t toff = 0, tlen = 0;
  std::string terr;
  if (zip_find_member(buf, "extra/allow_tf32", toff, tlen, terr) && tlen >= 1) m.allow_tf32 = buf[toff] == '1' ? 1 : 0;
  return m;
}

}  // namespace ahip
