// iem_model.hpp — parsed form of a transcribed model (include/iem_blob.h) plus the
// per-template slot analysis the kernels are generated from.
//
// A template corresponds to one ExaModels SIMDFunction created by an add_con/add_obj
// call of /root/reference/src/transform.jl (:458,559,597,614,700,741).  The analysis
// reproduces how ExaModels lays a template's derivatives out:
//   * reals (constants, item data, θ) fold into "fixed" unary nodes,
//   * Jacobian slots  = first-occurrence-unique variable index expressions met by a
//     left-to-right depth-first walk                                    (o1step),
//   * Hessian slots   = first-occurrence-unique ordered index pairs met by the
//     second-order walk (top-level +,-,real-scaling pass through; below that every
//     variable visit owns a diagonal slot, every binary node crosses its subtrees)
//                                                                         (o2step),
//   * offsets o0/o1/o2 are running counters in add_con/add_obj call order.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <deque>
#include <vector>

#include "../../include/iem_blob.h"

namespace iem {

enum NodeKind { K_REAL = 0, K_VAR = 1, K_N1 = 2, K_N2 = 3 };
enum Fixed { FX_NONE = 0, FX_FIRST = 1, FX_SECOND = 2 };

struct Node {
  int op = 0, a = 0, b = 0;
  double imm = 0.0;
  int kind = K_REAL, fixed = FX_NONE, inner = -1;
};

struct FieldDesc {
  int mode = 0;
  int64_t base = 0, step[3] = {0, 0, 0};
  int arr = -1;
};

struct IdxExpr {
  int64_t c0 = 0;
  int nterms = 0;
  int field[IEM_MAX_IDX_TERMS] = {0, 0, 0};
  int64_t coef[IEM_MAX_IDX_TERMS] = {0, 0, 0};
};

struct ArrayDesc {
  int kind = 0;
  int64_t n = 0;
  const void *data = nullptr;  // into the owned blob copy
  double fill = 0.0;
  int64_t r0 = 0, rstep = 0;
  double f(int64_t j) const {
    switch (kind) {
      case IEM_A_F64_DATA: return static_cast<const double *>(data)[j];
      case IEM_A_I64_DATA: return (double)static_cast<const int64_t *>(data)[j];
      case IEM_A_F64_FILL: return fill;
      default: return (double)(r0 + rstep * j);
    }
  }
  int64_t i(int64_t j) const {
    switch (kind) {
      case IEM_A_I64_DATA: return static_cast<const int64_t *>(data)[j];
      case IEM_A_I64_RANGE: return r0 + rstep * j;
      case IEM_A_F64_DATA: return (int64_t) static_cast<const double *>(data)[j];
      default: return (int64_t)fill;
    }
  }
};

struct Template {
  int kind = 0, nd = 1;
  int64_t n_items = 0, dims[3] = {1, 1, 1};
  int64_t grid_id = -1, origin[3] = {0, 0, 0};
  bool lattice_recovered = false;   // box and grid hint synthesised by recover_lattice (no real group ids)
  int root = 0;
  std::vector<FieldDesc> ifields, ffields;
  std::vector<IdxExpr> idx;
  std::vector<Node> nodes;
  int lmode = 0, umode = 0, larr = -1, uarr = -1;
  double lval = 0, uval = 0;
  // layout
  int64_t o0 = 0, o1 = 0, o2 = 0;
  int o1step = 0, o2step = 0;
  std::vector<int> comp1, comp2;          // visit -> slot
  std::vector<int> slot1_idx;             // slot -> idx id
  std::vector<int> slot2_i, slot2_j;      // slot -> idx ids (unswapped)
};

// one add_var slab of x: `dims` (first index fastest) starting at 0-based `off`; `group[a]` = the
// infinite-parameter group axis a runs over (0: none) — the blob's optional slab table (header word 9)
struct Slab {
  int64_t off = 0, dims[3] = {1, 1, 1};
  int nd = 1, group[3] = {0, 0, 0};
  int64_t length() const { return dims[0] * dims[1] * dims[2]; }
};

struct Model {
  std::vector<int64_t> blob;  // owned copy
  std::vector<Slab> slabs;    // empty when the producer wrote no slab table
  int64_t nvar = 0, npar = 0, ncon = 0, nnzj = 0, nnzh = 0;
  int minimize = 1;
  int arr_x0 = 0, arr_lvar = 0, arr_uvar = 0, arr_theta = 0;
  std::vector<ArrayDesc> arrs;
  std::vector<Template> tpl;
  std::deque<std::vector<double>> synth;  // arrays created by recover_lattice / the shard cut (ArrayDesc::data points here)
  std::deque<std::vector<int64_t>> synth_i;   // ... integer columns (shard cut of explicit item lists)
};

inline double w2d(int64_t w) {
  double d;
  std::memcpy(&d, &w, 8);
  return d;
}

inline bool is_linear_n1(const Node &nd) {
  if (nd.kind != K_N1) return false;
  if (nd.fixed != FX_NONE) return nd.op == IEM_OP_MUL || nd.op == IEM_OP_ADD || nd.op == IEM_OP_SUB;
  return nd.op == IEM_OP_NEG || nd.op == IEM_OP_POS;
}

// ---- symbolic walks (visit sequences) ---------------------------------------
struct SymWalk {
  const Template &t;
  std::vector<int> v1, vi, vj;
  explicit SymWalk(const Template &tt) : t(tt) {}
  void gr(int n) {
    const Node &nd = t.nodes[n];
    if (nd.kind == K_VAR) v1.push_back(nd.a);
    else if (nd.kind == K_N1) gr(nd.inner);
    else if (nd.kind == K_N2) { gr(nd.a); gr(nd.b); }
  }
  void hd(int n1, int n2) {
    const Node &a = t.nodes[n1], &b = t.nodes[n2];
    if (a.kind == K_REAL || b.kind == K_REAL) return;
    if (a.kind == K_VAR && b.kind == K_VAR) { vi.push_back(a.a); vj.push_back(b.a); }
    else if (a.kind == K_N1 && b.kind == K_N1) hd(a.inner, b.inner);
    else if (a.kind == K_VAR && b.kind == K_N1) hd(n1, b.inner);
    else if (a.kind == K_N1 && b.kind == K_VAR) hd(a.inner, n2);
    else if (a.kind == K_N2 && b.kind == K_N2) { hd(a.a, b.a); hd(a.a, b.b); hd(a.b, b.a); hd(a.b, b.b); }
    else if (a.kind == K_N1 && b.kind == K_N2) { hd(a.inner, b.a); hd(a.inner, b.b); }
    else if (a.kind == K_N2 && b.kind == K_N1) { hd(a.a, b.inner); hd(a.b, b.inner); }
    else if (a.kind == K_VAR && b.kind == K_N2) { hd(n1, b.a); hd(n1, b.b); }
    else { hd(a.a, n2); hd(a.b, n2); }
  }
  void hr(int n) {
    const Node &nd = t.nodes[n];
    if (nd.kind == K_VAR) { vi.push_back(nd.a); vj.push_back(nd.a); }
    else if (nd.kind == K_N1) hr(nd.inner);
    else if (nd.kind == K_N2) { hr(nd.a); hr(nd.b); hd(nd.a, nd.b); }
  }
  void hr0(int n) {
    const Node &nd = t.nodes[n];
    if (nd.kind == K_VAR || nd.kind == K_REAL) return;
    if (is_linear_n1(nd)) hr0(nd.inner);
    else if (nd.kind == K_N2 && (nd.op == IEM_OP_ADD || nd.op == IEM_OP_SUB)) { hr0(nd.a); hr0(nd.b); }
    else hr(n);
  }
};

// upper bound on any count in a blob (items, array lengths, nvar …): 2^40 elements = 8 TiB of doubles
constexpr int64_t IEM_MAX_COUNT = (int64_t)1 << 40;

inline void analyse_template(Template &t) {
  for (size_t n = 0; n < t.nodes.size(); ++n) {
    Node &nd = t.nodes[n];
    nd.fixed = FX_NONE;
    nd.inner = -1;
    if (nd.op == IEM_OP_VAR) nd.kind = K_VAR;
    else if (nd.op <= IEM_OP_PAR) nd.kind = K_REAL;
    else if (IEM_OP_IS_UNARY(nd.op)) {
      nd.kind = t.nodes[nd.a].kind == K_REAL ? K_REAL : K_N1;
      nd.inner = nd.a;
    } else if (IEM_OP_IS_BINARY(nd.op)) {
      int ka = t.nodes[nd.a].kind, kb = t.nodes[nd.b].kind;
      if (ka == K_REAL && kb == K_REAL) nd.kind = K_REAL;
      else if (ka == K_REAL) { nd.kind = K_N1; nd.fixed = FX_FIRST; nd.inner = nd.b; }
      else if (kb == K_REAL) { nd.kind = K_N1; nd.fixed = FX_SECOND; nd.inner = nd.a; }
      else nd.kind = K_N2;
    } else {
      throw std::runtime_error("unknown opcode " + std::to_string(nd.op));
    }
  }
  SymWalk w(t);
  w.gr(t.root);
  w.hr0(t.root);
  t.comp1.clear(); t.slot1_idx.clear();
  for (int id : w.v1) {
    int s = -1;
    for (size_t j = 0; j < t.slot1_idx.size(); ++j)
      if (t.slot1_idx[j] == id) { s = (int)j; break; }
    if (s < 0) { s = (int)t.slot1_idx.size(); t.slot1_idx.push_back(id); }
    t.comp1.push_back(s);
  }
  t.o1step = (int)t.slot1_idx.size();
  t.comp2.clear(); t.slot2_i.clear(); t.slot2_j.clear();
  for (size_t v = 0; v < w.vi.size(); ++v) {
    int s = -1;
    for (size_t j = 0; j < t.slot2_i.size(); ++j)
      if (t.slot2_i[j] == w.vi[v] && t.slot2_j[j] == w.vj[v]) { s = (int)j; break; }
    if (s < 0) { s = (int)t.slot2_i.size(); t.slot2_i.push_back(w.vi[v]); t.slot2_j.push_back(w.vj[v]); }
    t.comp2.push_back(s);
  }
  t.o2step = (int)t.slot2_i.size();
}

// A foreign producer (the Julia writer) sees a template's iterator as a flat list of records and
// writes one explicit column per item field.  When the iterator was an Iterators.product of two
// support sets (transform.jl:445: pandemic's t x xi), every integer column is affine on the
// n0 x n1 lattice  k = k0 + n0*k1  and most float columns depend on one coordinate only.  This
// pass recovers that structure: the template becomes a 2-D box with affine integer fields (no index
// column is read at run time), float columns shrink to n0 or n1 values, and the template joins the
// lane-fused kernel of its grid.  Item order — hence every COO position — is unchanged.
inline void recover_lattice(Model &m, Template &t) {
  if (t.nd != 1 || t.n_items < 4) return;
  const int64_t n = t.n_items;
  auto ival = [&](const FieldDesc &f, int64_t k) {
    return f.mode == IEM_F_AFFINE ? f.base + f.step[0] * k : m.arrs[f.arr].i(f.base + f.step[0] * k);
  };
  auto fval = [&](const FieldDesc &f, int64_t k) { return m.arrs[f.arr].f(f.base + f.step[0] * k); };
  int64_t n0 = 0;
  for (const FieldDesc &f : t.ifields) {
    if (f.mode != IEM_F_GATHER) continue;
    const int64_t v0 = ival(f, 0), a = ival(f, 1) - v0;
    for (int64_t k = 2; k < n; ++k)
      if (ival(f, k) != v0 + a * k) { n0 = k; break; }
    if (n0) break;
  }
  if (n0 < 2 || n % n0 != 0 || n / n0 < 2) return;
  const int64_t n1 = n / n0;
  std::vector<FieldDesc> nif(t.ifields.size());
  int coord[2] = {-1, -1};
  for (size_t i = 0; i < t.ifields.size(); ++i) {
    const FieldDesc &f = t.ifields[i];
    const int64_t v0 = ival(f, 0), a = ival(f, 1) - v0, b = ival(f, n0) - v0;
    for (int64_t k = 0; k < n; ++k)
      if (ival(f, k) != v0 + a * (k % n0) + b * (k / n0)) return;   // not a lattice: leave the template alone
    nif[i].mode = IEM_F_AFFINE; nif[i].base = v0; nif[i].step[0] = a; nif[i].step[1] = b; nif[i].step[2] = 0; nif[i].arr = -1;
    if (a == 1 && b == 0 && coord[0] < 0) coord[0] = (int)i;
    if (a == 0 && b == 1 && coord[1] < 0) coord[1] = (int)i;
  }
  std::vector<FieldDesc> nff(t.ffields.size());
  std::vector<std::pair<size_t, std::vector<double>>> fresh;   // (ffield, new array) — committed only on success
  for (size_t i = 0; i < t.ffields.size(); ++i) {
    const FieldDesc &f = t.ffields[i];
    bool only0 = true, only1 = true;
    for (int64_t k = 0; k < n && (only0 || only1); ++k) {
      const double v = fval(f, k);
      uint64_t bv, b0, b1;
      const double v0 = fval(f, k % n0), v1 = fval(f, (k / n0) * n0);
      std::memcpy(&bv, &v, 8); std::memcpy(&b0, &v0, 8); std::memcpy(&b1, &v1, 8);
      if (bv != b0) only0 = false;
      if (bv != b1) only1 = false;
    }
    nff[i] = f;
    if (only0 || only1) {
      std::vector<double> a(only0 ? n0 : n1);
      for (int64_t j = 0; j < (int64_t)a.size(); ++j) a[j] = fval(f, only0 ? j : j * n0);
      nff[i].base = 0; nff[i].step[0] = only0 ? 1 : 0; nff[i].step[1] = only0 ? 0 : 1; nff[i].step[2] = 0;
      fresh.emplace_back(i, std::move(a));
    } else {
      nff[i].step[1] = f.step[0] * n0;
    }
  }
  for (auto &pr : fresh) {
    m.synth.push_back(std::move(pr.second));
    ArrayDesc a;
    a.kind = IEM_A_F64_DATA; a.n = (int64_t)m.synth.back().size(); a.data = m.synth.back().data();
    nff[pr.first].arr = (int)m.arrs.size();
    m.arrs.push_back(a);
  }
  t.ifields = nif; t.ffields = nff;
  t.lattice_recovered = true;
  t.nd = 2; t.dims[0] = n0; t.dims[1] = n1; t.dims[2] = 1;
  if (coord[0] >= 0 && coord[1] >= 0) {   // both grid coordinates are item fields: fusable on the recovered grid
    t.grid_id = 2 * 4096 + 1;
    t.origin[0] = nif[coord[0]].base - 1; t.origin[1] = nif[coord[1]].base - 1; t.origin[2] = 0;
  } else {
    // no coordinate fields (collocation rows: node/element boxes): templates over the same n0 x n1
    // box share lanes from its corner — a scheduling hint only, like every grid id
    t.grid_id = 3 * 4096 + 1 + (n0 & 1023);
    t.origin[0] = t.origin[1] = t.origin[2] = 0;
  }
}

inline void parse_blob(const void *blob, size_t nbytes, Model &m) {
  if (nbytes < 8 * IEM_HDR_WORDS || nbytes % 8) throw std::runtime_error("blob too small / not word aligned");
  const int64_t *w0 = static_cast<const int64_t *>(blob);
  if (w0[0] != IEM_BLOB_MAGIC) throw std::runtime_error("bad blob magic");
  if (w0[1] != IEM_BLOB_VERSION) throw std::runtime_error("unsupported blob version");
  if ((size_t)w0[8] * 8 != nbytes) throw std::runtime_error("blob length mismatch");
  m.blob.assign(w0, w0 + nbytes / 8);
  const int64_t *w = m.blob.data();
  const int64_t total = w[8];
  m.nvar = w[2]; m.npar = w[3]; m.ncon = w[4];
  int64_t n_tpl = w[5], n_arr = w[6];
  if (m.nvar < 0 || m.npar < 0 || m.ncon < 0 || m.nvar > IEM_MAX_COUNT || m.npar > IEM_MAX_COUNT || m.ncon > IEM_MAX_COUNT)
    throw std::runtime_error("nvar / npar / ncon out of range");
  if (n_tpl < 0 || n_arr < 0 || n_tpl > total || n_arr > total) throw std::runtime_error("blob tables overrun");
  m.minimize = (int)w[7];
  m.arr_x0 = (int)w[10]; m.arr_lvar = (int)w[11]; m.arr_uvar = (int)w[12]; m.arr_theta = (int)w[13];
  if (IEM_HDR_WORDS + IEM_ARR_WORDS * n_arr + n_tpl > total) throw std::runtime_error("blob tables overrun");
  m.arrs.resize(n_arr);
  const int64_t *aw = w + IEM_HDR_WORDS;
  for (int64_t i = 0; i < n_arr; ++i, aw += IEM_ARR_WORDS) {
    ArrayDesc &a = m.arrs[i];
    a.kind = (int)aw[0]; a.n = aw[1];
    if (aw[0] < IEM_A_F64_DATA || aw[0] > IEM_A_I64_RANGE) throw std::runtime_error("unknown array kind");
    if (a.n < 0 || a.n > IEM_MAX_COUNT) throw std::runtime_error("array length out of range");
    if (a.kind == IEM_A_F64_DATA || a.kind == IEM_A_I64_DATA) {
      if (aw[2] < 0 || aw[2] > total || a.n > total - aw[2]) throw std::runtime_error("array payload out of range");
      a.data = w + aw[2];
    }
    a.fill = w2d(aw[3]); a.r0 = aw[3]; a.rstep = aw[4];
  }
  auto chk_arr = [&](int id, int64_t need, const char *what) {
    if (id < 0 || id >= (int)n_arr || m.arrs[id].n < need) throw std::runtime_error(std::string("bad array for ") + what);
  };
  if (w[10] != m.arr_x0 || w[11] != m.arr_lvar || w[12] != m.arr_uvar || w[13] != m.arr_theta) throw std::runtime_error("bad core array id");
  chk_arr(m.arr_x0, m.nvar, "x0"); chk_arr(m.arr_lvar, m.nvar, "lvar");
  chk_arr(m.arr_uvar, m.nvar, "uvar"); chk_arr(m.arr_theta, m.npar, "theta");
  const int64_t *tw = w + IEM_HDR_WORDS + IEM_ARR_WORDS * n_arr;
  m.tpl.resize(n_tpl);
  int64_t o0 = 0, o1 = 0, o2 = 0;
  for (int64_t i = 0; i < n_tpl; ++i) {
    if (tw[i] < 0 || tw[i] + IEM_TPL_FIXED_WORDS > total) throw std::runtime_error("template offset out of range");
    const int64_t *p = w + tw[i];
    Template &t = m.tpl[i];
    t.kind = (int)*p++; t.n_items = *p++; t.nd = (int)*p++;
    for (int d = 0; d < 3; ++d) t.dims[d] = *p++;
    t.grid_id = *p++;
    for (int d = 0; d < 3; ++d) t.origin[d] = *p++;
    int n_if = (int)*p++, n_ff = (int)*p++, n_idx = (int)*p++, n_nodes = (int)*p++;
    t.root = (int)*p++;
    t.lmode = (int)*p++; t.lval = w2d(*p++); t.larr = (int)*p++;
    t.umode = (int)*p++; t.uval = w2d(*p++); t.uarr = (int)*p++;
    bool box_ok = t.nd >= 1 && t.nd <= 3 && t.n_items >= 0 && t.n_items <= IEM_MAX_COUNT;
    for (int d = 0; d < 3 && box_ok; ++d) box_ok = t.dims[d] >= 0 && t.dims[d] <= IEM_MAX_COUNT && (d < t.nd || t.dims[d] == 1);
    if (!box_ok || (__int128)t.dims[0] * t.dims[1] * t.dims[2] != (__int128)t.n_items)
      throw std::runtime_error("bad template item box");
    for (int d = 0; d < 3; ++d)
      if (t.origin[d] < -IEM_MAX_COUNT || t.origin[d] > IEM_MAX_COUNT) throw std::runtime_error("bad template grid origin");
    auto chk_bound = [&](int mode, int arr) {
      if (mode != 0 && mode != 1) throw std::runtime_error("bad constraint bound mode");
      if (mode == 1 && (arr < 0 || arr >= (int)n_arr || m.arrs[arr].n < t.n_items)) throw std::runtime_error("bad constraint bound array");
    };
    chk_bound(t.lmode, t.larr); chk_bound(t.umode, t.uarr);
    int64_t need = (int64_t)IEM_FIELD_WORDS * (n_if + n_ff) + (int64_t)IEM_IDX_WORDS * n_idx + (int64_t)IEM_NODE_WORDS * n_nodes;
    if (n_if < 0 || n_ff < 0 || n_idx < 0 || n_nodes <= 0 || (p - w) + need > total)
      throw std::runtime_error("template record overruns blob");
    if (n_if >= 65536 || n_ff >= 65536 || n_idx >= 65536 || n_nodes >= 65536)
      throw std::runtime_error("template too large (more than 65535 fields / index expressions / nodes)");
    auto rd_field = [&](FieldDesc &f, bool is_int) {
      const int64_t mode = *p++, arr = p[4];
      f.mode = (int)mode; f.base = *p++;
      for (int d = 0; d < 3; ++d) f.step[d] = *p++;
      f.arr = (int)*p++;
      if (mode != IEM_F_AFFINE && mode != IEM_F_GATHER) throw std::runtime_error("unknown field mode");
      if (!is_int && mode != IEM_F_GATHER) throw std::runtime_error("float item field must be a gather");
      __int128 lo = f.base, hi = f.base;
      for (int d = 0; d < 3; ++d) {
        if (t.dims[d] == 0) continue;
        __int128 e = (__int128)f.step[d] * (t.dims[d] - 1);
        if (e < 0) lo += e; else hi += e;
      }
      if (lo < -(__int128)IEM_MAX_COUNT || hi > (__int128)IEM_MAX_COUNT) throw std::runtime_error("item field value range too large");
      if (mode == IEM_F_GATHER) {
        if (arr < 0 || arr >= n_arr) throw std::runtime_error("field array id out of range");
        const ArrayDesc &a = m.arrs[arr];
        if (is_int && a.kind != IEM_A_I64_DATA && a.kind != IEM_A_I64_RANGE) throw std::runtime_error("integer item field gathers from a float array");
        if (t.n_items > 0 && (lo < 0 || hi >= a.n)) throw std::runtime_error("field gather out of array bounds");
      }
    };
    t.ifields.resize(n_if); t.ffields.resize(n_ff);
    for (auto &f : t.ifields) rd_field(f, true);
    for (auto &f : t.ffields) rd_field(f, false);
    t.idx.resize(n_idx);
    for (auto &ix : t.idx) {
      ix.c0 = *p++; ix.nterms = (int)*p++;
      if (ix.nterms < 0 || ix.nterms > IEM_MAX_IDX_TERMS) throw std::runtime_error("bad index expression");
      for (int j = 0; j < IEM_MAX_IDX_TERMS; ++j) {
        ix.field[j] = (int)*p++; ix.coef[j] = *p++;
        if (j < ix.nterms && (ix.field[j] < 0 || ix.field[j] >= n_if)) throw std::runtime_error("index field out of range");
      }
    }
    t.nodes.resize(n_nodes);
    for (int n = 0; n < n_nodes; ++n) {
      Node &nd = t.nodes[n];
      nd.op = (int)*p++; nd.a = (int)*p++; nd.b = (int)*p++; nd.imm = w2d(*p++);
      bool bin = IEM_OP_IS_BINARY(nd.op), un = IEM_OP_IS_UNARY(nd.op);
      if (!bin && !un && (nd.op < IEM_OP_CONST || nd.op > IEM_OP_VAR)) throw std::runtime_error("unknown opcode " + std::to_string(nd.op));
      if ((bin || un) && (nd.a < 0 || nd.a >= n)) throw std::runtime_error("node child out of order");
      if (bin && (nd.b < 0 || nd.b >= n)) throw std::runtime_error("node child out of order");
      if ((nd.op == IEM_OP_VAR || nd.op == IEM_OP_PAR) && (nd.a < 0 || nd.a >= n_idx)) throw std::runtime_error("node index id out of range");
      if (nd.op == IEM_OP_DATA && (nd.a < 0 || nd.a >= n_ff)) throw std::runtime_error("node field id out of range");
    }
    if (t.root < 0 || t.root >= n_nodes) throw std::runtime_error("bad root");
    recover_lattice(m, t);
    analyse_template(t);
    t.o2 = o2; o2 += t.n_items * t.o2step;
    if (o2 > IEM_MAX_COUNT * 64 || o1 > IEM_MAX_COUNT * 64) throw std::runtime_error("nnz out of range");
    if (t.kind == IEM_T_CON) {
      t.o0 = o0; o0 += t.n_items;
      t.o1 = o1; o1 += t.n_items * t.o1step;
    } else if (t.kind != IEM_T_OBJ) {
      throw std::runtime_error("bad template kind");
    }
  }
  if (o0 != m.ncon) throw std::runtime_error("ncon does not match the constraint templates");
  m.nnzj = o1; m.nnzh = o2;
  if (w[9] != 0) {   // optional slab table: n, then n x {off, nd, dims[3], group[3]}; must tile 0..nvar
    if (w[9] < IEM_HDR_WORDS || w[9] >= total) throw std::runtime_error("slab table offset out of range");
    const int64_t *sw = w + w[9];
    const int64_t ns = sw[0];
    if (ns < 0 || ns > total || w[9] + 1 + IEM_SLAB_WORDS * ns > total) throw std::runtime_error("slab table overruns blob");
    int64_t cover = 0;
    m.slabs.resize(ns);
    for (int64_t i = 0; i < ns; ++i) {
      const int64_t *q = sw + 1 + IEM_SLAB_WORDS * i;
      Slab &sl = m.slabs[i];
      sl.off = q[0]; sl.nd = (int)q[1];
      if (q[1] < 1 || q[1] > 3) throw std::runtime_error("bad slab rank");
      for (int d = 0; d < 3; ++d) {
        sl.dims[d] = q[2 + d]; sl.group[d] = (int)q[5 + d];
        if (sl.dims[d] < 0 || sl.dims[d] > IEM_MAX_COUNT || (d >= sl.nd && sl.dims[d] != 1) || q[5 + d] < 0 || q[5 + d] > 4094)
          throw std::runtime_error("bad slab record");
      }
      if ((__int128)sl.dims[0] * sl.dims[1] * sl.dims[2] > (__int128)IEM_MAX_COUNT || sl.off != cover) throw std::runtime_error("slab table does not tile the variables");
      cover += sl.length();
    }
    if (cover != m.nvar) throw std::runtime_error("slab table does not tile the variables");
  }
}

}  // namespace iem
