// iem_codegen.cpp — specialising generator: templates → fused gfx950 lane programs.
//
// Design: templates that iterate over the same support grid are fused
// into ONE kernel per NLPModels call (cons!/jac_coord!/hess_coord!/obj/grad!), lane =
// grid point.  Every template of the group is differentiated symbolically into a
// hash-consed expression DAG following the reverse sweeps ExaModels performs per item
// (first order: depth-first adjoint walk; second order: hrpass0/hrpass/hdrpass — same
// visit order, same accumulation order as the CPU restatement), so
//   * x[off_k + q] is loaded once per lane however many templates read it,
//   * sin/cos/tan of a state are computed once per lane (sincos pairing),
//   * a template's item block goes out through the block store of iem_device.h
//     (iem_stage / iem_flush: LDS staging, whole 128-byte lines, overlapped tiles).
// One launch per call: grids that share a call become __device__ bodies behind one
// workgroup-id dispatcher; a grid of <= split_small workgroups gets one body per template
// (latency-bound regime), larger grids one lane-fused body (HBM-bound regime).
// The emitted code is size-independent within a regime: slab offsets, extents and COO
// offsets are kernel arguments, so one code object serves 10^5 and 10^7 supports.
#include "iem_codegen.hpp"

#include <algorithm>
#include <array>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <tuple>
#include <set>
#include <sstream>
#include <unordered_map>

namespace iem {

uint64_t fnv1a64(const std::string &s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

namespace {

constexpr double kD2R = 0.017453292519943295;
constexpr double kR2D = 57.29577951308232;
constexpr double kLN2 = 0.6931471805599453;
constexpr double kLN10 = 2.302585092994046;

// `space` names what the value indexes (0 grid/box constants, 1 x, 2 theta, 3 an output position,
// 100 + slot an item-data array, 200 + slot an integer column): values of different spaces are never
// merged, so that a NUMERIC coincidence between, say, a local slab offset and a global data offset
// (which depends on the rank and the world size of a shard) cannot change the shape of the source.
struct AffQ {
  int64_t c = 0, k[3] = {0, 0, 0};
  int space = 0;
  bool operator<(const AffQ &o) const {
    if (space != o.space) return space < o.space;
    if (c != o.c) return c < o.c;
    for (int d = 0; d < 3; ++d) if (k[d] != o.k[d]) return k[d] < o.k[d];
    return false;
  }
  bool operator==(const AffQ &o) const { return space == o.space && c == o.c && k[0] == o.k[0] && k[1] == o.k[1] && k[2] == o.k[2]; }
};

struct Group {
  int64_t grid_id = -1;
  int nd = 1;
  int64_t lo[3] = {0, 0, 0}, ext[3] = {1, 1, 1};
  std::vector<int> tpls, scalars;
  // flat: the box is walked by ONE linear lane index (q = q0 + ext0*(q1 + ext1*q2)) instead of
  // lanes along dim 0 only — used when dim 0 is too short to fill a wave (e.g. the 2 x (S-1)
  // item box of an OrthogonalCollocation(3) derivative)
  bool flat = false;
  // FOLDED templates (orthogonal collocation): the 1-D grid also carries templates whose items form node x element boxes
  // (w x ne, w <= fold_n) or element lists — on a 2-D grid (t x xi) boxes with the second grid dimension behind them
  // (w x ne x n_xi).  Every lane derives  qe = floor((q0 - fold_off) / fold_n)  (the element) and
  // q2 = (q0 - fold_off) mod fold_n  (the node within it); a folded template's item (j, e) is evaluated by lane
  // fold_off + j + fold_n * e ("natural", w = fold_n) or by the lane that owns the entry it adds into (a "pinned" clone:
  // lanes with q2 == K evaluate item (J, q1 + de)).  A scheduling decision only, like every grid id.
  int64_t fold_n = 0, fold_off = 0;
  std::set<int> folded;
};

std::string hexf(double v) {
  if (std::isnan(v)) return "__builtin_nan(\"\")";
  if (std::isinf(v)) return v > 0 ? "__builtin_inf()" : "(-__builtin_inf())";
  char buf[64];
  std::snprintf(buf, sizeof buf, "%a", v);
  std::string s(buf);
  if (s[0] == '-') return "(" + s + ")";
  return s;
}

// pseudo unary ops beyond the blob vocabulary
enum { U_SGN = 1000 };

// ---------------------------------------------------------------------------
// expression DAG
// ---------------------------------------------------------------------------
enum VOp { VC = 0, VDP = 1, VLD = 2, VUN = 3, VBIN = 4, VW = 5, VSEL = 6, VGUARD = 7 };   // VGUARD: guard[sub] ? a : 0

struct VNode {
  int op, sub, a, b, c;
  double imm;
};

struct VKey {
  int op, sub, a, b, c;
  uint64_t bits;
  bool operator<(const VKey &o) const {
    if (op != o.op) return op < o.op;
    if (sub != o.sub) return sub < o.sub;
    if (a != o.a) return a < o.a;
    if (b != o.b) return b < o.b;
    if (c != o.c) return c < o.c;
    return bits < o.bits;
  }
};

struct IdxVal {  // 1-based variable/parameter index or 0-based array position
  AffQ aff;
  std::vector<std::pair<int64_t, int>> ind;  // coef × iload id
  bool operator<(const IdxVal &o) const {
    if (!(aff == o.aff)) return aff < o.aff;
    return ind < o.ind;
  }
};

struct ILoad {
  int ia_slot;
  AffQ pos;
  std::set<int> guards;
};

struct Load {
  int arr;       // 0 x, 1 theta, 2 y, 3 fa[slot]
  int slot;
  int idxval;    // position (0-based) as IdxVal id
  std::set<int> guards;
};

struct Output {
  int kind;          // KernelKind
  int tpl;
  int guard;
  int pos_idx;       // IdxVal id: position of slot 0 / the row / unused
  int64_t pos_off = 0;   // ... = pos_off + (values per item) * (item ordinal)
  std::vector<int> vals;      // DAG ids
  std::vector<int> grad_idx;  // KK_GRAD: IdxVal ids (0-based) per value
  std::vector<int> grad_mode; // 0 exclusive store, 1 wave-uniform, 2 atomic, 3 shared entry (parked), 4 axis sum (parked rows), -1 folded
  std::map<int, int64_t> axis_off;   // slot -> offset (doubles) of its rows in the kind's aux buffer (mode 4)
  std::vector<int> slot_ia, slot_ib;  // KK_HESS: 1-based IdxVal ids of each slot's pair
  std::vector<int> slot_ti, slot_tj;  // ... and the template-local index-expression ids
  bool scalar = false;
  int64_t qlo[3] = {0, 0, 0}, qhi[3] = {1, 1, 1};  // valid q range of the template in the launch domain
  // folded templates (Group::fold_n): 0 not folded, 1 natural, 2 pinned — then only the lanes of [qlo0, qhi0) with
  // q2 == pin_K are items (J, q1 + de)
  int fold = 0;
  int64_t pin_J = 0, pin_K = -1, pin_de = 0;
  std::vector<int> grad_tidx;   // scatter kinds: template index-expression id behind each destination
};

class KernelBuilder {
 public:
  KernelBuilder(const Model &m, const Group &g, int kind, const Options &opt, const std::string &name)
      : m_(m), g_(g), kind_(kind), opt_(opt), name_(name) {}

  // ---- parameters ---------------------------------------------------------
  std::string ip(int64_t v, int space = 0) {
    auto key = std::make_pair(space, v);
    auto it = ip_ids_.find(key);
    if (it == ip_ids_.end()) {
      it = ip_ids_.emplace(key, (int)ipv_.size()).first;
      ipv_.push_back(v);
    }
    return "A.ip[" + std::to_string(it->second) + "]";
  }
  std::string coefstr(int64_t v) {
    if (v >= -16 && v <= 16) return std::to_string(v) + "LL";
    return ip(v);
  }
  // appends raw entries (no de-duplication) and returns the index of the first: tables the device
  // code walks with a pointer (A.ip + start)
  size_t ip_block(const std::vector<int64_t> &vals) {
    const size_t start = ipv_.size();
    ipv_.insert(ipv_.end(), vals.begin(), vals.end());
    return start;
  }
  int dp(double v) {
    uint64_t b; std::memcpy(&b, &v, 8);
    auto it = dp_ids_.find(b);
    if (it == dp_ids_.end()) { it = dp_ids_.emplace(b, (int)dpv_.size()).first; dpv_.push_back(v); }
    return it->second;
  }
  int fa_slot(int arr) {
    auto it = fa_ids_.find(arr);
    if (it == fa_ids_.end()) { it = fa_ids_.emplace(arr, (int)fav_.size()).first; fav_.push_back(arr); }
    return it->second;
  }
  int ia_slot(int arr) {
    auto it = ia_ids_.find(arr);
    if (it == ia_ids_.end()) { it = ia_ids_.emplace(arr, (int)iav_.size()).first; iav_.push_back(arr); }
    return it->second;
  }

  // ---- DAG ------------------------------------------------------------------
  int mk(int op, int sub, int a, int b, int c, double imm) {
    uint64_t bits; std::memcpy(&bits, &imm, 8);
    VKey k{op, sub, a, b, c, bits};
    auto it = memo_.find(k);
    if (it != memo_.end()) return it->second;
    int id = (int)v_.size();
    v_.push_back(VNode{op, sub, a, b, c, imm});
    memo_.emplace(k, id);
    return id;
  }
  int C(double x) { return mk(VC, 0, -1, -1, -1, x); }
  bool isC(int id, double *val = nullptr) const {
    if (v_[id].op != VC) return false;
    if (val) *val = v_[id].imm;
    return true;
  }
  bool isCv(int id, double x) const { return v_[id].op == VC && v_[id].imm == x; }
  int neg(int a) {
    double x;
    if (isC(a, &x)) return C(-x);
    if (v_[a].op == VUN && v_[a].sub == IEM_OP_NEG) return v_[a].a;
    return mk(VUN, IEM_OP_NEG, a, -1, -1, 0);
  }
  int un(int op, int a) {
    if (op == IEM_OP_NEG) return neg(a);
    if (op == IEM_OP_POS) return a;
    return mk(VUN, op, a, -1, -1, 0);
  }
  int add(int a, int b) {
    double x, y;
    if (isC(a, &x) && isC(b, &y)) return C(x + y);
    if (isCv(a, 0.0)) return b;
    if (isCv(b, 0.0)) return a;
    return mk(VBIN, IEM_OP_ADD, a, b, -1, 0);
  }
  int sub(int a, int b) {
    double x, y;
    if (isC(a, &x) && isC(b, &y)) return C(x - y);
    if (isCv(b, 0.0)) return a;
    if (isCv(a, 0.0)) return neg(b);
    return mk(VBIN, IEM_OP_SUB, a, b, -1, 0);
  }
  int mul(int a, int b) {
    double x, y;
    if (isC(a, &x) && isC(b, &y)) return C(x * y);
    if (isCv(a, 0.0) || isCv(b, 0.0)) return C(0.0);
    if (isCv(a, 1.0)) return b;
    if (isCv(b, 1.0)) return a;
    if (isCv(a, -1.0)) return neg(b);
    if (isCv(b, -1.0)) return neg(a);
    return mk(VBIN, IEM_OP_MUL, a, b, -1, 0);
  }
  int div(int a, int b) {
    double x, y;
    if (isC(a, &x) && isC(b, &y)) return C(x / y);
    if (isCv(b, 1.0)) return a;
    return mk(VBIN, IEM_OP_DIV, a, b, -1, 0);
  }
  int powv(int a, int b) {
    if (isCv(b, 1.0)) return a;
    return mk(VBIN, IEM_OP_POW, a, b, -1, 0);
  }

  // ---- index values -----------------------------------------------------------
  int idxval(const IdxVal &iv) {
    auto it = idx_ids_.find(iv);
    if (it != idx_ids_.end()) return it->second;
    int id = (int)idx_.size();
    idx_.push_back(iv);
    idx_ids_.emplace(iv, id);
    return id;
  }
  int iload(int ia, const AffQ &pos, int guard) {
    for (size_t i = 0; i < iloads_.size(); ++i)
      if (iloads_[i].ia_slot == ia && iloads_[i].pos == pos) { iloads_[i].guards.insert(guard); return (int)i; }
    iloads_.push_back(ILoad{ia, pos, {guard}});
    return (int)iloads_.size() - 1;
  }
  int load(int arr, int slot, int idxv, int guard) {
    auto key = std::make_tuple(arr, slot, idxv);
    auto it = load_ids_.find(key);
    int id;
    if (it == load_ids_.end()) {
      id = (int)loads_.size();
      loads_.push_back(Load{arr, slot, idxv, {}});
      load_ids_.emplace(key, id);
    } else id = it->second;
    loads_[id].guards.insert(guard);
    return mk(VLD, id, -1, -1, -1, 0);
  }

  // ---- template geometry --------------------------------------------------------
  struct TGeo {
    bool scalar = false;
    int64_t sh[3] = {0, 0, 0};  // k_d = q_d + sh_d
    int guard = 0;
    int64_t qlo[3] = {0, 0, 0}, qhi[3] = {1, 1, 1};
    int fold = 0;                    // 1 natural, 2 pinned (Group::fold_n)
    int64_t J = 0, K = 0, de = 0;    // pinned: lanes with q2 == K evaluate item (J, q1 + de)
  };
  struct Pin { int64_t J, K, de; };
  bool is_folded(int ti) const { return g_.fold_n > 0 && g_.folded.count(ti) != 0; }
  // node x element box of a folded template: w nodes (1 for an element list), ne elements
  // a folded template's box: w nodes (1 for an element list) x ne elements [x the grid's second dimension]
  void fold_box(const Template &t, int64_t &w, int64_t &ne) const {
    const int own = t.nd - (g_.nd - 1);          // dimensions of the node x element part
    if (own >= 2) { w = t.dims[0]; ne = t.dims[1]; } else { w = 1; ne = t.dims[0]; }
  }
  void fold_steps(const Template &t, const FieldDesc &f, int64_t &sj, int64_t &se, int64_t &sx) const {
    const int own = t.nd - (g_.nd - 1);
    if (own >= 2) { sj = f.step[0]; se = f.step[1]; } else { sj = 0; se = f.step[0]; }
    sx = g_.nd == 2 ? f.step[own] : 0;
  }
  // value  base + sj*j + se*e [+ sx*xi]  of a folded template's item as a function of the lane.  An element stride that is
  // a multiple of fold_n is rewritten on q0 (n*qe = q0 - off - q2): node-indexed slabs are then read and written at
  // c + k*q0, the SAME index value the templates of the grid itself use — loads merge, scatter slots meet in registers.
  // (On a 1-D grid an element stride that is not such a multiple — the row ordinal of an element list — stays on the
  // element coordinate, which q1 carries there; a 2-D grid has its own q1 and folds full boxes only.)
  AffQ fold_aff(const TGeo &G, int64_t base, int64_t sj, int64_t se, int64_t sx) const {
    const int64_t n = g_.fold_n, off = g_.fold_off;
    AffQ a;
    if (g_.nd == 2 && se % n != 0) throw std::runtime_error("internal: element stride of a box folded onto a 2-D grid");
    if (G.fold == 1) {
      if (se % n == 0) { const int64_t b = se / n; a.c = base - b * off; a.k[0] = b; a.k[2] = sj - b; }
      else { a.c = base; a.k[1] = se; a.k[2] = sj; }
    } else {
      const int64_t c = base + sj * G.J + se * G.de;
      if (se % n == 0) { const int64_t b = se / n; a.c = c - b * (off + G.K); a.k[0] = b; }
      else { a.c = c; a.k[1] = se; }
    }
    if (g_.nd == 2) { a.c += sx * G.sh[1]; a.k[1] = sx; }
    return a;
  }

  // guard id: canonical text → id
  int guard_id(const std::string &txt) {
    auto it = guard_ids_.find(txt);
    if (it == guard_ids_.end()) { it = guard_ids_.emplace(txt, (int)guards_.size()).first; guards_.push_back(txt); }
    return it->second;
  }

  // shift0: lane q of dim 0 evaluates the item that normally sits on lane q - shift0 (a "pulled" clone of the
  // template, build(): scatter slots that land on a neighbour lane's entry are computed BY that neighbour)
  TGeo geo(int ti, bool scalar, int64_t shift0 = 0, const Pin *pin = nullptr) {
    const Template &t = m_.tpl[ti];
    TGeo G;
    G.scalar = scalar;
    std::ostringstream os;
    if (scalar) {
      os << (g_.fold_n > 0 ? (g_.nd == 2 ? "(q0 == 0 && q1 == 0)" : "(q0 == 0)") : "(q0 == 0 && q1 == 0 && q2 == 0)");
    } else if (is_folded(ti)) {
      const int64_t n = g_.fold_n, off = g_.fold_off;
      int64_t w, ne;
      fold_box(t, w, ne);
      os << "inb";
      if (pin) {
        G.fold = 2; G.J = pin->J; G.K = pin->K; G.de = pin->de;
        const int64_t q1lo = -pin->de, q1hi = ne - pin->de;   // q1 + de in [0, ne)
        os << " && q2 == " << pin->K << "LL && qe >= " << coefstr(q1lo) << " && qe < " << ip(q1hi);
        G.qlo[0] = off + pin->K + n * q1lo; G.qhi[0] = off + pin->K + n * (q1hi - 1) + 1;
      } else {
        if (w != n) throw std::runtime_error("internal: natural geometry of a partial folded template");
        G.fold = 1;
        G.qlo[0] = off; G.qhi[0] = off + n * ne;
        if (G.qlo[0] > 0) os << " && q0 >= " << coefstr(G.qlo[0]);
        if (G.qhi[0] < g_.ext[0]) os << " && q0 < " << ip(G.qhi[0]);
      }
      if (g_.nd == 2) {      // the grid's second dimension is the box's last
        const int dx = t.nd - 1;
        G.sh[1] = g_.lo[1] - t.origin[dx];
        G.qlo[1] = -G.sh[1]; G.qhi[1] = t.dims[dx] - G.sh[1];
        if (G.qlo[1] > 0) os << " && q1 >= " << coefstr(G.qlo[1]);
        if (G.qhi[1] < g_.ext[1]) os << " && q1 < " << ip(G.qhi[1]);
      }
    } else {
      os << "inb";
      for (int d = 0; d < g_.nd; ++d) {
        G.sh[d] = g_.lo[d] - t.origin[d] - (d == 0 ? shift0 : 0);
        int64_t qlo = -G.sh[d], qhi = t.dims[d] - G.sh[d];  // valid q range [qlo, qhi)
        G.qlo[d] = qlo; G.qhi[d] = qhi;
        if (qlo > 0) os << " && q" << d << " >= " << coefstr(qlo);
        if (qhi < g_.ext[d]) os << " && q" << d << " < " << ip(qhi);
      }
    }
    G.guard = guard_id(os.str());
    return G;
  }

  AffQ field_aff(const Template &t, const FieldDesc &f, const TGeo &G) const {
    if (G.fold) {
      int64_t sj, se, sx;
      fold_steps(t, f, sj, se, sx);
      return fold_aff(G, f.base, sj, se, sx);
    }
    AffQ a;
    a.c = f.base;
    for (int d = 0; d < t.nd; ++d) {
      if (G.scalar) continue;
      a.c += f.step[d] * G.sh[d];
      a.k[d] = f.step[d];
    }
    return a;
  }

  int tpl_idx(int ti, int idx_id, const TGeo &G, int64_t bias, int space) {
    const Template &t = m_.tpl[ti];
    const IdxExpr &ix = t.idx[idx_id];
    IdxVal iv;
    iv.aff.space = space;
    iv.aff.c = ix.c0 + bias;
    for (int j = 0; j < ix.nterms; ++j) {
      const FieldDesc &f = t.ifields[ix.field[j]];
      AffQ fa = field_aff(t, f, G);
      if (f.mode == IEM_F_AFFINE) {
        iv.aff.c += ix.coef[j] * fa.c;
        for (int d = 0; d < 3; ++d) iv.aff.k[d] += ix.coef[j] * fa.k[d];
      } else {
        fa.space = 200 + ia_slot(f.arr);
        int il = iload(ia_slot(f.arr), fa, G.guard);
        iv.ind.emplace_back(ix.coef[j], il);
      }
    }
    std::sort(iv.ind.begin(), iv.ind.end());
    return idxval(iv);
  }

  AffQ klin_aff(const Template &t, const TGeo &G, int64_t scale, int64_t off) const {
    if (G.fold) {   // item ordinal j + w*e
      int64_t w, ne;
      fold_box(t, w, ne);
      const int own = t.nd - (g_.nd - 1);
      AffQ a = fold_aff(G, off, own >= 2 ? scale : 0, own >= 2 ? scale * w : scale, g_.nd == 2 ? scale * w * ne : 0);
      a.space = 3;
      return a;
    }
    AffQ a;
    a.space = 3;
    a.c = off;
    int64_t stride = 1;
    for (int d = 0; d < t.nd; ++d) {
      if (!G.scalar) {
        a.c += scale * stride * G.sh[d];
        a.k[d] = scale * stride;
      }
      stride *= t.dims[d];
    }
    return a;
  }

  // ---- per-template symbolic differentiation -----------------------------------
  struct TplGen {
    KernelBuilder &K;
    int ti;
    const Template &t;
    TGeo G;
    std::vector<int> val, y1, y2, h11, h12, h22;
    std::vector<int> vidx;  // idx id -> IdxVal id (1-based value)
    TplGen(KernelBuilder &k, int ti_, const TGeo &g) : K(k), ti(ti_), t(k.m_.tpl[ti_]), G(g) {
      size_t n = t.nodes.size();
      val.assign(n, -1); y1.assign(n, -1); y2.assign(n, -1); h11.assign(n, -1); h12.assign(n, -1); h22.assign(n, -1);
      vidx.assign(t.idx.size(), -1);
    }
    int idx1(int id) {
      if (vidx[id] < 0) vidx[id] = K.tpl_idx(ti, id, G, 0, 1);
      return vidx[id];
    }
    int pos0(int id, int space = 1) {  // 0-based position of a 1-based index (space 1: x / g, 2: theta)
      return K.tpl_idx(ti, id, G, -1, space);
    }
    void forward(int order) {
      for (size_t n = 0; n < t.nodes.size(); ++n) {
        const Node &nd = t.nodes[n];
        switch (nd.op) {
          case IEM_OP_CONST: val[n] = K.C(nd.imm); break;
          case IEM_OP_DATA: {
            const FieldDesc &f = t.ffields[nd.a];
            const ArrayDesc &ad = K.m_.arrs[f.arr];
            if (ad.kind == IEM_A_F64_FILL) {
              val[n] = K.mk(VDP, K.dp(ad.fill), -1, -1, -1, 0);
            } else {
              IdxVal iv;
              iv.aff = K.field_aff(t, f, G);
              iv.aff.space = 100 + K.fa_slot(f.arr);
              val[n] = K.load(3, K.fa_slot(f.arr), K.idxval(iv), G.guard);
            }
            break;
          }
          case IEM_OP_PAR: val[n] = K.load(1, 0, pos0(nd.a, 2), G.guard); break;
          case IEM_OP_VAR: val[n] = K.load(0, 0, pos0(nd.a), G.guard); break;
          default:
            if (IEM_OP_IS_UNARY(nd.op)) unary((int)n, order);
            else binary((int)n, order);
        }
      }
    }
    void unary(int n, int order) {
      const Node &nd = t.nodes[n];
      KernelBuilder &B = K;
      int x = val[nd.a];
      int f = -1, d = -1, h = -1;
      auto Cn = [&](double v) { return B.C(v); };
      bool need = order >= 1 && nd.kind != K_REAL;
      switch (nd.op) {
        case IEM_OP_NEG: f = B.neg(x); d = Cn(-1); h = Cn(0); break;
        case IEM_OP_POS: f = x; d = Cn(1); h = Cn(0); break;
        case IEM_OP_INV: { int u = B.div(Cn(1), x); f = u; if (need) { d = B.neg(B.mul(u, u)); h = B.mul(B.mul(Cn(2), B.mul(u, u)), u); } break; }
        case IEM_OP_SQRT: { int s = B.un(IEM_OP_SQRT, x); f = s; if (need) { d = B.div(Cn(0.5), s); h = B.div(Cn(-0.25), B.mul(x, s)); } break; }
        case IEM_OP_CBRT: { int c = B.un(IEM_OP_CBRT, x); f = c; if (need) { d = B.div(Cn(1), B.mul(B.mul(Cn(3), c), c)); h = B.div(Cn(-2), B.mul(B.mul(B.mul(Cn(9), x), c), c)); } break; }
        case IEM_OP_ABS: f = B.un(IEM_OP_ABS, x); if (need) { d = B.mk(VUN, U_SGN, x, -1, -1, 0); h = Cn(0); } break;
        case IEM_OP_ABS2: f = B.mul(x, x); if (need) { d = B.mul(Cn(2), x); h = Cn(2); } break;
        case IEM_OP_EXP: f = B.un(IEM_OP_EXP, x); d = f; h = f; break;
        case IEM_OP_EXP2: f = B.un(IEM_OP_EXP2, x); if (need) { d = B.mul(f, Cn(kLN2)); h = B.mul(B.mul(f, Cn(kLN2)), Cn(kLN2)); } break;
        case IEM_OP_LOG: f = B.un(IEM_OP_LOG, x); if (need) { int u = B.div(Cn(1), x); d = u; h = B.neg(B.mul(u, u)); } break;
        case IEM_OP_LOG2: f = B.un(IEM_OP_LOG2, x); if (need) { int u = B.div(Cn(1), x); d = B.div(u, Cn(kLN2)); h = B.div(B.neg(B.mul(u, u)), Cn(kLN2)); } break;
        case IEM_OP_LOG10: f = B.un(IEM_OP_LOG10, x); if (need) { int u = B.div(Cn(1), x); d = B.div(u, Cn(kLN10)); h = B.div(B.neg(B.mul(u, u)), Cn(kLN10)); } break;
        case IEM_OP_LOG1P: f = B.un(IEM_OP_LOG1P, x); if (need) { int u = B.div(Cn(1), B.add(Cn(1), x)); d = u; h = B.neg(B.mul(u, u)); } break;
        case IEM_OP_SIN: f = B.un(IEM_OP_SIN, x); if (need) { d = B.un(IEM_OP_COS, x); h = B.neg(f); } break;
        case IEM_OP_COS: f = B.un(IEM_OP_COS, x); if (need) { d = B.neg(B.un(IEM_OP_SIN, x)); h = B.neg(f); } break;
        case IEM_OP_TAN: { f = B.un(IEM_OP_TAN, x); if (need) { int u = B.add(Cn(1), B.mul(f, f)); d = u; h = B.mul(B.mul(Cn(2), f), u); } break; }
        case IEM_OP_ASIN: { f = B.un(IEM_OP_ASIN, x); if (need) { int u = B.sub(Cn(1), B.mul(x, x)); int s = B.un(IEM_OP_SQRT, u); d = B.div(Cn(1), s); h = B.div(x, B.mul(u, s)); } break; }
        case IEM_OP_ACOS: { f = B.un(IEM_OP_ACOS, x); if (need) { int u = B.sub(Cn(1), B.mul(x, x)); int s = B.un(IEM_OP_SQRT, u); d = B.div(Cn(-1), s); h = B.div(B.neg(x), B.mul(u, s)); } break; }
        case IEM_OP_CSC: { int s = B.div(Cn(1), B.un(IEM_OP_SIN, x)); f = s; if (need) { int tt = B.mul(B.un(IEM_OP_COS, x), s); d = B.mul(B.neg(s), tt); h = B.mul(s, B.add(B.mul(tt, tt), B.mul(s, s))); } break; }
        case IEM_OP_SEC: { int c = B.div(Cn(1), B.un(IEM_OP_COS, x)); f = c; if (need) { int tt = B.mul(B.un(IEM_OP_SIN, x), c); d = B.mul(c, tt); h = B.mul(c, B.add(B.mul(tt, tt), B.mul(c, c))); } break; }
        case IEM_OP_COT: { int tt = B.div(Cn(1), B.un(IEM_OP_TAN, x)); f = tt; if (need) { int u = B.add(Cn(1), B.mul(tt, tt)); d = B.neg(u); h = B.mul(B.mul(Cn(2), tt), u); } break; }
        case IEM_OP_ATAN: { f = B.un(IEM_OP_ATAN, x); if (need) { int u = B.div(Cn(1), B.add(Cn(1), B.mul(x, x))); d = u; h = B.mul(B.mul(B.mul(Cn(-2), x), u), u); } break; }
        case IEM_OP_ACOT: { f = B.un(IEM_OP_ATAN, B.div(Cn(1), x)); if (need) { int u = B.div(Cn(1), B.add(Cn(1), B.mul(x, x))); d = B.neg(u); h = B.mul(B.mul(B.mul(Cn(2), x), u), u); } break; }
        case IEM_OP_SIND: { int xr = B.mul(x, Cn(kD2R)); f = B.un(IEM_OP_SIN, xr); if (need) { d = B.mul(Cn(kD2R), B.un(IEM_OP_COS, xr)); h = B.mul(Cn(-(kD2R * kD2R)), f); } break; }
        case IEM_OP_COSD: { int xr = B.mul(x, Cn(kD2R)); f = B.un(IEM_OP_COS, xr); if (need) { d = B.mul(Cn(-kD2R), B.un(IEM_OP_SIN, xr)); h = B.mul(Cn(-(kD2R * kD2R)), f); } break; }
        case IEM_OP_TAND: { int xr = B.mul(x, Cn(kD2R)); f = B.un(IEM_OP_TAN, xr); if (need) { int u = B.add(Cn(1), B.mul(f, f)); d = B.mul(Cn(kD2R), u); h = B.mul(B.mul(Cn((kD2R * kD2R) * 2.0), f), u); } break; }
        case IEM_OP_CSCD: { int xr = B.mul(x, Cn(kD2R)); int s = B.div(Cn(1), B.un(IEM_OP_SIN, xr)); f = s; if (need) { int tt = B.mul(B.un(IEM_OP_COS, xr), s); d = B.mul(B.mul(Cn(-kD2R), s), tt); h = B.mul(B.mul(Cn(kD2R * kD2R), s), B.add(B.mul(tt, tt), B.mul(s, s))); } break; }
        case IEM_OP_SECD: { int xr = B.mul(x, Cn(kD2R)); int c = B.div(Cn(1), B.un(IEM_OP_COS, xr)); f = c; if (need) { int tt = B.mul(B.un(IEM_OP_SIN, xr), c); d = B.mul(B.mul(Cn(kD2R), c), tt); h = B.mul(B.mul(Cn(kD2R * kD2R), c), B.add(B.mul(tt, tt), B.mul(c, c))); } break; }
        case IEM_OP_COTD: { int xr = B.mul(x, Cn(kD2R)); int tt = B.div(Cn(1), B.un(IEM_OP_TAN, xr)); f = tt; if (need) { int u = B.add(Cn(1), B.mul(tt, tt)); d = B.mul(Cn(-kD2R), u); h = B.mul(B.mul(Cn((kD2R * kD2R) * 2.0), tt), u); } break; }
        case IEM_OP_ATAND: { f = B.mul(Cn(kR2D), B.un(IEM_OP_ATAN, x)); if (need) { int u = B.div(Cn(1), B.add(Cn(1), B.mul(x, x))); d = B.mul(Cn(kR2D), u); h = B.mul(B.mul(B.mul(Cn(-kR2D * 2.0), x), u), u); } break; }
        case IEM_OP_ACOTD: { f = B.mul(Cn(kR2D), B.un(IEM_OP_ATAN, B.div(Cn(1), x))); if (need) { int u = B.div(Cn(1), B.add(Cn(1), B.mul(x, x))); d = B.mul(Cn(-kR2D), u); h = B.mul(B.mul(B.mul(Cn(kR2D * 2.0), x), u), u); } break; }
        case IEM_OP_SINH: f = B.un(IEM_OP_SINH, x); if (need) { d = B.un(IEM_OP_COSH, x); h = f; } break;
        case IEM_OP_COSH: f = B.un(IEM_OP_COSH, x); if (need) { d = B.un(IEM_OP_SINH, x); h = f; } break;
        case IEM_OP_TANH: { f = B.un(IEM_OP_TANH, x); if (need) { int u = B.sub(Cn(1), B.mul(f, f)); d = u; h = B.mul(B.mul(Cn(-2), f), u); } break; }
        case IEM_OP_CSCH: { int s = B.div(Cn(1), B.un(IEM_OP_SINH, x)); f = s; if (need) { int tt = B.mul(B.un(IEM_OP_COSH, x), s); d = B.mul(B.neg(s), tt); h = B.mul(s, B.add(B.mul(tt, tt), B.mul(s, s))); } break; }
        case IEM_OP_SECH: { int c = B.div(Cn(1), B.un(IEM_OP_COSH, x)); f = c; if (need) { int tt = B.un(IEM_OP_TANH, x); d = B.mul(B.neg(c), tt); h = B.mul(c, B.sub(B.mul(tt, tt), B.mul(c, c))); } break; }
        case IEM_OP_COTH: { int tt = B.div(Cn(1), B.un(IEM_OP_TANH, x)); f = tt; if (need) { int u = B.sub(Cn(1), B.mul(tt, tt)); d = u; h = B.mul(B.mul(Cn(-2), tt), u); } break; }
        case IEM_OP_ATANH: { f = B.un(IEM_OP_ATANH, x); if (need) { int u = B.div(Cn(1), B.sub(Cn(1), B.mul(x, x))); d = u; h = B.mul(B.mul(B.mul(Cn(2), x), u), u); } break; }
        case IEM_OP_ACOTH: { f = B.un(IEM_OP_ATANH, B.div(Cn(1), x)); if (need) { int u = B.div(Cn(1), B.sub(Cn(1), B.mul(x, x))); d = u; h = B.mul(B.mul(B.mul(Cn(2), x), u), u); } break; }
        default: throw std::runtime_error("codegen: unsupported unary opcode " + std::to_string(nd.op));
      }
      val[n] = f;
      if (need) { y1[n] = d; h11[n] = h; }
    }
    void binary(int n, int order) {
      const Node &nd = t.nodes[n];
      KernelBuilder &B = K;
      int a = val[nd.a], b = val[nd.b];
      auto Cn = [&](double v) { return B.C(v); };
      switch (nd.op) {
        case IEM_OP_ADD: val[n] = B.add(a, b); break;
        case IEM_OP_SUB: val[n] = B.sub(a, b); break;
        case IEM_OP_MUL: val[n] = B.mul(a, b); break;
        case IEM_OP_DIV: val[n] = B.div(a, b); break;
        case IEM_OP_POW: val[n] = B.powv(a, b); break;
        default: throw std::runtime_error("codegen: unsupported binary opcode");
      }
      if (order < 1 || nd.kind == K_REAL) return;
      int p1 = Cn(0), p2 = Cn(0), q11 = Cn(0), q12 = Cn(0), q22 = Cn(0);
      int need = nd.fixed;
      switch (nd.op) {
        case IEM_OP_ADD: p1 = Cn(1); p2 = Cn(1); break;
        case IEM_OP_SUB: p1 = Cn(1); p2 = Cn(-1); break;
        case IEM_OP_MUL: p1 = b; p2 = a; q12 = Cn(1); break;
        case IEM_OP_DIV: {
          int ib = B.div(Cn(1), b);
          p1 = ib;
          if (need != FX_SECOND) {
            p2 = B.mul(B.mul(B.neg(a), ib), ib);
            q22 = B.mul(B.mul(B.mul(B.mul(Cn(2), a), ib), ib), ib);
          }
          if (need == FX_NONE) q12 = B.neg(B.mul(ib, ib));
          break;
        }
        case IEM_OP_POW: {
          if (need != FX_FIRST) {
            p1 = B.mul(b, B.powv(a, B.sub(b, Cn(1))));
            q11 = B.mul(B.mul(b, B.sub(b, Cn(1))), B.powv(a, B.sub(b, Cn(2))));
          }
          if (need != FX_SECOND) {
            int la = B.un(IEM_OP_LOG, a), p = val[n];
            p2 = B.mul(p, la);
            q22 = B.mul(B.mul(p, la), la);
            if (need == FX_NONE) q12 = B.mul(B.powv(a, B.sub(b, Cn(1))), B.add(Cn(1), B.mul(b, la)));
          }
          break;
        }
      }
      if (nd.fixed == FX_FIRST) { y1[n] = p2; h11[n] = q22; }
      else if (nd.fixed == FX_SECOND) { y1[n] = p1; h11[n] = q11; }
      else { y1[n] = p1; y2[n] = p2; h11[n] = q11; h12[n] = q12; h22[n] = q22; }
    }

    // first-order reverse sweep → per-slot sums
    std::vector<int> slots1;
    int gr(int n, int cnt, int adj) {
      const Node &nd = t.nodes[n];
      switch (nd.kind) {
        case K_VAR: {
          int s = t.comp1[cnt];
          slots1[s] = slots1[s] < 0 ? adj : K.add(slots1[s], adj);
          return cnt + 1;
        }
        case K_N1: return gr(nd.inner, cnt, K.mul(adj, y1[n]));
        case K_N2:
          cnt = gr(nd.a, cnt, K.mul(adj, y1[n]));
          return gr(nd.b, cnt, K.mul(adj, y2[n]));
        default: return cnt;
      }
    }
    // second-order
    std::vector<int> slots2;
    void acc2(int cnt, int v) {
      int s = t.comp2[cnt];
      slots2[s] = slots2[s] < 0 ? v : K.add(slots2[s], v);
    }
    int hd(int n1, int n2, int cnt, int adj) {
      const Node &a = t.nodes[n1], &b = t.nodes[n2];
      if (a.kind == K_REAL || b.kind == K_REAL) return cnt;
      if (a.kind == K_VAR && b.kind == K_VAR) {
        int two = K.mul(K.C(2.0), adj);
        int ia = idx1(a.a), ib = idx1(b.a);
        int v;
        if (ia == ib) v = two;
        else {
          const IdxVal &A = K.idx_[ia], &Bv = K.idx_[ib];
          bool never = A.ind.empty() && Bv.ind.empty() && A.aff.k[0] == Bv.aff.k[0] && A.aff.k[1] == Bv.aff.k[1] &&
                       A.aff.k[2] == Bv.aff.k[2] && A.aff.c != Bv.aff.c;
          v = never ? adj : K.mk(VSEL, K.sel_id(ia, ib), two, adj, -1, 0);
        }
        acc2(cnt, v);
        return cnt + 1;
      } else if (a.kind == K_N1 && b.kind == K_N1) {
        return hd(a.inner, b.inner, cnt, K.mul(K.mul(adj, y1[n1]), y1[n2]));
      } else if (a.kind == K_VAR && b.kind == K_N1) {
        return hd(n1, b.inner, cnt, K.mul(adj, y1[n2]));
      } else if (a.kind == K_N1 && b.kind == K_VAR) {
        return hd(a.inner, n2, cnt, K.mul(adj, y1[n1]));
      } else if (a.kind == K_N2 && b.kind == K_N2) {
        cnt = hd(a.a, b.a, cnt, K.mul(K.mul(adj, y1[n1]), y1[n2]));
        cnt = hd(a.a, b.b, cnt, K.mul(K.mul(adj, y1[n1]), y2[n2]));
        cnt = hd(a.b, b.a, cnt, K.mul(K.mul(adj, y2[n1]), y1[n2]));
        return hd(a.b, b.b, cnt, K.mul(K.mul(adj, y2[n1]), y2[n2]));
      } else if (a.kind == K_N1 && b.kind == K_N2) {
        cnt = hd(a.inner, b.a, cnt, K.mul(K.mul(adj, y1[n1]), y1[n2]));
        return hd(a.inner, b.b, cnt, K.mul(K.mul(adj, y1[n1]), y2[n2]));
      } else if (a.kind == K_N2 && b.kind == K_N1) {
        cnt = hd(a.a, b.inner, cnt, K.mul(K.mul(adj, y1[n1]), y1[n2]));
        return hd(a.b, b.inner, cnt, K.mul(K.mul(adj, y2[n1]), y1[n2]));
      } else if (a.kind == K_VAR && b.kind == K_N2) {
        cnt = hd(n1, b.a, cnt, K.mul(adj, y1[n2]));
        return hd(n1, b.b, cnt, K.mul(adj, y2[n2]));
      } else {
        cnt = hd(a.a, n2, cnt, K.mul(adj, y1[n1]));
        return hd(a.b, n2, cnt, K.mul(adj, y2[n1]));
      }
    }
    int hr(int n, int cnt, int adj, int adj2) {
      const Node &nd = t.nodes[n];
      switch (nd.kind) {
        case K_VAR: acc2(cnt, adj2); return cnt + 1;
        case K_N1: {
          int y = y1[n];
          return hr(nd.inner, cnt, K.mul(adj, y), K.add(K.mul(adj2, K.mul(y, y)), K.mul(adj, h11[n])));
        }
        case K_N2: {
          int ya = y1[n], yb = y2[n];
          int adj2y1y2 = K.mul(K.mul(adj2, ya), yb);
          int adjh12 = K.mul(adj, h12[n]);
          cnt = hr(nd.a, cnt, K.mul(adj, ya), K.add(K.mul(adj2, K.mul(ya, ya)), K.mul(adj, h11[n])));
          cnt = hr(nd.b, cnt, K.mul(adj, yb), K.add(K.mul(adj2, K.mul(yb, yb)), K.mul(adj, h22[n])));
          return hd(nd.a, nd.b, cnt, K.add(adj2y1y2, adjh12));
        }
        default: return cnt;
      }
    }
    int hr0(int n, int cnt, int adj, int adj2) {
      const Node &nd = t.nodes[n];
      if (nd.kind == K_VAR || nd.kind == K_REAL) return cnt;
      if (is_linear_n1(nd)) {
        int y = y1[n];
        return hr0(nd.inner, cnt, K.mul(adj, y), K.mul(adj2, K.mul(y, y)));
      }
      if (nd.kind == K_N2 && (nd.op == IEM_OP_ADD || nd.op == IEM_OP_SUB)) {
        cnt = hr0(nd.a, cnt, K.mul(adj, y1[n]), adj2);
        return hr0(nd.b, cnt, K.mul(adj, y2[n]), adj2);
      }
      return hr(n, cnt, adj, adj2);
    }
  };

  int sel_id(int ia, int ib) {
    for (size_t i = 0; i < sels_.size(); ++i)
      if (sels_[i].first == ia && sels_[i].second == ib) return (int)i;
    sels_.emplace_back(ia, ib);
    return (int)sels_.size() - 1;
  }

  // ---- build the kernel's outputs ---------------------------------------------
  bool relevant(const Template &t) const {
    switch (kind_) {
      case KK_CONS: return t.kind == IEM_T_CON;
      case KK_JAC: return t.kind == IEM_T_CON && t.o1step > 0;
      case KK_HESS: return t.o2step > 0;
      case KK_OBJ: return t.kind == IEM_T_OBJ;
      case KK_GRAD: return t.kind == IEM_T_OBJ && t.o1step > 0;
      case KK_JPROD: return t.kind == IEM_T_CON;
      case KK_JTPROD: return t.kind == IEM_T_CON && t.o1step > 0;
      case KK_HPROD: return t.o2step > 0;
    }
    return false;
  }

  bool build(const std::vector<std::pair<AffQ, std::pair<int64_t, int64_t>>> *grad_ranges) {
    (void)grad_ranges;
    std::vector<std::pair<int, bool>> order;
    for (int ti : g_.tpls) order.emplace_back(ti, false);
    for (int ti : g_.scalars) order.emplace_back(ti, true);
    std::sort(order.begin(), order.end());
    auto make_output = [&](int ti, bool scalar, int64_t shift0, const Pin *pin) -> Output {
      const Template &t = m_.tpl[ti];
      TGeo G = geo(ti, scalar, shift0, pin);
      TplGen tg(*this, ti, G);
      Output o;
      o.kind = kind_; o.tpl = ti; o.guard = G.guard; o.pos_idx = -1;
      o.scalar = scalar;
      o.fold = G.fold;
      if (G.fold == 2) { o.pin_J = G.J; o.pin_K = G.K; o.pin_de = G.de; }
      for (int d = 0; d < 3; ++d) { o.qlo[d] = G.qlo[d]; o.qhi[d] = G.qhi[d]; }
      switch (kind_) {
        case KK_CONS: {
          tg.forward(0);
          IdxVal iv; iv.aff = klin_aff(t, G, 1, t.o0);
          o.pos_idx = idxval(iv); o.pos_off = t.o0;
          o.vals = {tg.val[t.root]};
          alg_w_ += t.n_items;
          break;
        }
        case KK_OBJ:
          tg.forward(0);
          o.vals = {tg.val[t.root]};
          break;
        case KK_JAC:
        case KK_GRAD: {
          tg.forward(1);
          tg.slots1.assign(t.o1step, -1);
          tg.gr(t.root, 0, C(1.0));
          o.vals = tg.slots1;
          if (kind_ == KK_JAC) {
            IdxVal iv; iv.aff = klin_aff(t, G, t.o1step, t.o1);
            o.pos_idx = idxval(iv); o.pos_off = t.o1;
            alg_w_ += t.n_items * t.o1step;
          } else {
            for (int s = 0; s < t.o1step; ++s) {
              int id = tg.pos0(t.slot1_idx[s]);
              o.grad_idx.push_back(id);
              o.grad_tidx.push_back(t.slot1_idx[s]);
              o.grad_mode.push_back(2);
              alg_w_ += t.n_items;
            }
          }
          break;
        }
        case KK_JPROD: {   // (J v)[row] = sum_slots dc/dx_slot * v[col(slot)]
          tg.forward(1);
          tg.slots1.assign(t.o1step, -1);
          tg.gr(t.root, 0, C(1.0));
          int acc = C(0.0);
          for (int s = 0; s < t.o1step; ++s) {
            int vv = load(4, 0, tg.pos0(t.slot1_idx[s]), G.guard);
            acc = add(acc, mul(tg.slots1[s] < 0 ? C(0.0) : tg.slots1[s], vv));
          }
          IdxVal iv; iv.aff = klin_aff(t, G, 1, t.o0);
          o.pos_idx = idxval(iv); o.pos_off = t.o0;
          o.vals = {acc};
          alg_w_ += t.n_items;
          break;
        }
        case KK_JTPROD: {  // (J' v)[col] += dc/dx_slot * v[row]  == gradient of  v . c(x)
          tg.forward(1);
          tg.slots1.assign(t.o1step, -1);
          IdxVal rv; rv.aff = klin_aff(t, G, 1, t.o0); rv.aff.space = 4;   // a row index (into v), not an output position
          int seed = load(4, 0, idxval(rv), G.guard);
          tg.gr(t.root, 0, seed);
          o.vals = tg.slots1;
          for (int s = 0; s < t.o1step; ++s) {
            o.grad_idx.push_back(tg.pos0(t.slot1_idx[s]));
            o.grad_tidx.push_back(t.slot1_idx[s]);
            o.grad_mode.push_back(2);
            alg_w_ += t.n_items;
          }
          break;
        }
        case KK_HPROD: {   // (H v): slot (a, b) with value h adds h*v[b] to row a and, if a != b, h*v[a] to row b
          tg.forward(2);
          tg.slots2.assign(t.o2step, -1);
          int adj;
          if (t.kind == IEM_T_OBJ) adj = mk(VW, 0, -1, -1, -1, 0);
          else {
            IdxVal iv; iv.aff = klin_aff(t, G, 1, t.o0); iv.aff.space = 4;   // a row index (into y), not an output position
            adj = load(2, 0, idxval(iv), G.guard);
          }
          tg.hr0(t.root, 0, adj, C(0.0));
          std::map<int, int> dest;   // destination IdxVal id (0-based position) -> accumulated DAG value
          std::map<int, int> dest_tidx;
          std::vector<int> dest_order;
          auto contribute = [&](int pos_id, int val, int tidx) {
            auto it = dest.find(pos_id);
            if (it == dest.end()) { dest.emplace(pos_id, val); dest_tidx.emplace(pos_id, tidx); dest_order.push_back(pos_id); }
            else it->second = add(it->second, val);
          };
          for (int s = 0; s < t.o2step; ++s) {
            int h = tg.slots2[s] < 0 ? C(0.0) : tg.slots2[s];
            int ia = tg.idx1(t.slot2_i[s]), ib = tg.idx1(t.slot2_j[s]);
            int pa = tg.pos0(t.slot2_i[s]), pb = tg.pos0(t.slot2_j[s]);
            contribute(pa, mul(h, load(4, 0, pb, G.guard)), t.slot2_i[s]);
            if (ia != ib) {
              int c2 = mul(h, load(4, 0, pa, G.guard));
              const IdxVal &A = idx_[ia], &Bv = idx_[ib];
              bool never = A.ind.empty() && Bv.ind.empty() && A.aff.k[0] == Bv.aff.k[0] && A.aff.k[1] == Bv.aff.k[1] &&
                           A.aff.k[2] == Bv.aff.k[2] && A.aff.c != Bv.aff.c;
              if (!never) c2 = mk(VSEL, sel_id(ia, ib), C(0.0), c2, -1, 0);
              contribute(pb, c2, t.slot2_j[s]);
            }
          }
          for (int pid : dest_order) {
            o.vals.push_back(dest[pid]);
            o.grad_idx.push_back(pid);
            o.grad_tidx.push_back(dest_tidx[pid]);
            o.grad_mode.push_back(2);
            alg_w_ += t.n_items;
          }
          break;
        }
        case KK_HESS: {
          tg.forward(2);
          tg.slots2.assign(t.o2step, -1);
          int adj;
          if (t.kind == IEM_T_OBJ) adj = mk(VW, 0, -1, -1, -1, 0);
          else {
            IdxVal iv; iv.aff = klin_aff(t, G, 1, t.o0); iv.aff.space = 4;   // a row index (into y), not an output position
            adj = load(2, 0, idxval(iv), G.guard);
          }
          tg.hr0(t.root, 0, adj, C(0.0));
          o.vals = tg.slots2;
          for (int s2 = 0; s2 < t.o2step; ++s2) {
            o.slot_ia.push_back(tg.idx1(t.slot2_i[s2])); o.slot_ib.push_back(tg.idx1(t.slot2_j[s2]));
            o.slot_ti.push_back(t.slot2_i[s2]); o.slot_tj.push_back(t.slot2_j[s2]);
          }
          IdxVal iv; iv.aff = klin_aff(t, G, t.o2step, t.o2);
          o.pos_idx = idxval(iv); o.pos_off = t.o2;
          if (!opt_.hess_merge) alg_w_ += t.n_items * t.o2step;
          break;
        }
      }
      for (int &v : o.vals) if (v < 0) v = C(0.0);
      return o;
    };
    for (auto &pr : order) {
      if (!relevant(m_.tpl[pr.first])) continue;
      const Template &t = m_.tpl[pr.first];
      int64_t w = 0, ne = 0;
      if (!pr.second && is_folded(pr.first)) fold_box(t, w, ne);
      if (w > 0 && w != g_.fold_n) {
        // a partial folded template (fewer nodes per element than the fold) has no natural geometry: node J of every
        // element starts on the lanes with q2 == J, pull_folded moves each slot to the lane that owns its entry
        for (int64_t J = 0; J < w; ++J) { Pin pin{J, J, 0}; outs_.push_back(make_output(pr.first, false, 0, &pin)); }
        continue;
      }
      outs_.push_back(make_output(pr.first, pr.second, 0, nullptr));
    }
    if (kind_ == KK_GRAD || kind_ == KK_JTPROD || kind_ == KK_HPROD) {
      if (opt_.pull_scatter) pull_neighbours(make_output);
      if (g_.fold_n > 0) pull_folded(make_output);
      merge_scatter();
    }
    return !outs_.empty();
  }

  // Scatter kinds, folded templates (orthogonal collocation).  Row (j, e) of a derivative adds into the entries of ALL
  // nodes of its element and of the element's lower boundary — which is the LAST node of the element before, and which
  // the rows of the grid itself (the dynamics at that support) write too.  Every slot of a folded template whose
  // destination is affine in (j, e) with an element stride of fold_n entries is therefore computed by the lane that OWNS
  // the entry: for node J of every element a clone of the template pinned to the lanes with q2 == K evaluates item
  // (J, q1 + de), its destination is the same index value the owner's own slots have, merge_scatter sums them in registers
  // and the entry leaves through one exclusive store.  fold_n (+1) clones per slot, each a handful of multiplies — the
  // derivative approximation is linear, its partials are item data (src/transform.jl:511-562 of the reference).
  template <class MakeOutput>
  void pull_folded(MakeOutput &make_output) {
    const int64_t n = g_.fold_n, off = g_.fold_off;
    auto fdiv = [](int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; };
    // pure lane-affine destinations of the grid's own templates — and of the natural folded ones (a row's own node: a
    // state no row of the grid itself touches still has ONE owner per entry): (space, k0) -> constants
    std::map<std::array<int64_t, 3>, std::vector<int64_t>> canon;     // (space, lane stride, stride of the grid's second dimension)
    const bool two_d = g_.nd == 2;
    for (const Output &o : outs_) {
      if (o.scalar || o.fold == 2) continue;
      for (size_t s = 0; s < o.grad_idx.size(); ++s) {
        const IdxVal &iv = idx_[o.grad_idx[s]];
        if (iv.ind.empty() && iv.aff.k[0] != 0 && (two_d || iv.aff.k[1] == 0) && iv.aff.k[2] == 0)
          canon[{(int64_t)iv.aff.space, iv.aff.k[0], two_d ? iv.aff.k[1] : 0}].push_back(iv.aff.c);
      }
    }
    std::map<std::array<int64_t, 4>, std::vector<int>> moved;   // (template, J, K, de) -> slots
    const size_t n_outs = outs_.size();
    for (size_t oi = 0; oi < n_outs; ++oi) {
      Output &o = outs_[oi];
      if (!o.fold || o.scalar) continue;
      const Template &t = m_.tpl[o.tpl];
      int64_t w, ne;
      fold_box(t, w, ne);
      for (size_t s = 0; s < o.grad_idx.size(); ++s) {
        if (o.grad_mode[s] < 0) continue;
        // raw destination (0-based): c + a*j + b*e [+ x*xi, which every lane of the family shares]
        const IdxExpr &ix = t.idx[o.grad_tidx[s]];
        int64_t c = ix.c0 - 1, a = 0, b = 0, xk = 0;
        bool affine = true;
        for (int j = 0; j < ix.nterms; ++j) {
          const FieldDesc &f = t.ifields[ix.field[j]];
          if (f.mode != IEM_F_AFFINE) { affine = false; break; }
          int64_t sj, se, sx;
          fold_steps(t, f, sj, se, sx);
          c += ix.coef[j] * f.base; a += ix.coef[j] * sj; b += ix.coef[j] * se; xk += ix.coef[j] * sx;
        }
        if (!affine || b == 0 || b % n != 0) continue;
        if (two_d) c += xk * (g_.lo[1] - t.origin[t.nd - 1]);     // (as fold_aff: the constant at q1 = 0)
        const int64_t bp = b / n;
        const int space = idx_[o.grad_idx[s]].aff.space;
        // the owner map  entry = c* + bp*q0: the grid's own slot of that stride nearest to this one, else the slot's own lanes
        int64_t cstar = c - bp * off;
        bool found = false;
        auto it = canon.find({(int64_t)space, bp, two_d ? xk : 0});
        if (it != canon.end())
          for (int64_t cg : it->second) {
            if ((c - cg) % bp != 0) continue;
            const int64_t m0 = (c - cg) / bp - off;
            if (m0 < -n || m0 > 2 * n) continue;
            if (!found || std::llabs(m0) < std::llabs((c - cstar) / bp - off)) { cstar = cg; found = true; }
          }
        if (a % bp != 0 || (c - cstar) % bp != 0) continue;
        const int64_t Jlo = o.fold == 2 ? o.pin_J : 0, Jhi = o.fold == 2 ? o.pin_J + 1 : w;
        std::vector<std::array<int64_t, 3>> plan;   // (J, K, de) per node
        bool stay = true, ok = true;
        for (int64_t J = Jlo; J < Jhi && ok; ++J) {
          const int64_t mJ = (c + a * J - cstar) / bp - off;   // owner lane - off - n*e
          const int64_t K = mJ - n * fdiv(mJ, n), de = -fdiv(mJ, n);
          const int64_t qlo = off + K + n * (-de), qhi = off + K + n * (ne - de - 1) + 1;
          if (qlo < 0 || qhi > g_.ext[0]) ok = false;
          const bool home = o.fold == 2 ? (K == o.pin_K && de == o.pin_de) : (K == J && de == 0);
          if (!home) stay = false;
          plan.push_back({J, K, de});
        }
        if (!ok || stay) continue;
        o.grad_mode[s] = -1;
        for (auto &pl : plan) moved[{(int64_t)o.tpl, pl[0], pl[1], pl[2]}].push_back((int)s);
      }
    }
    for (auto &mv : moved) {
      Pin pin{mv.first[1], mv.first[2], mv.first[3]};
      Output clone = make_output((int)mv.first[0], false, 0, &pin);
      std::vector<char> keep(clone.vals.size(), 0);
      for (int s : mv.second) keep[s] = 1;
      for (size_t s = 0; s < clone.vals.size(); ++s)
        if (!keep[s]) clone.grad_mode[s] = -1;
      outs_.push_back(std::move(clone));
    }
  }

  // Scatter kinds, stencil neighbours.  A backward difference row at lane i adds into x[i] AND into x[i-1] — the
  // entry of the lane next door, which that lane also writes: both must be atomics, and the output must be zeroed
  // first.  Instead the NEIGHBOUR computes that addend itself: slots of this kernel whose destinations are the same
  // affine map of the lane up to a small shift along dim 0 form a family; the shift most items use is canonical,
  // and a slot at another shift is taken out of its template and computed by a clone of the template that is
  // shifted by as many lanes (geo(shift0): every index, load and guard moves with it), so that its destination
  // coincides with the canonical one — merge_scatter then sums the family in registers and the entry leaves
  // through ONE exclusive store.  The clone re-evaluates the template for the neighbouring item: a derivative
  // approximation (src/transform.jl:511-562 of the reference) is linear, its partials are item data.
  template <class MakeOutput>
  void pull_neighbours(MakeOutput &make_output) {
    if (g_.flat) return;
    struct Ref { int out, slot; int64_t c, items; };
    std::map<std::array<int64_t, 4>, std::vector<Ref>> fam;   // (space, k0, k1, k2) -> slots
    auto box_items = [](const Output &o) { int64_t n = 1; for (int d = 0; d < 3; ++d) n *= std::max<int64_t>(o.qhi[d] - o.qlo[d], 0); return n; };
    for (size_t oi = 0; oi < outs_.size(); ++oi) {
      const Output &o = outs_[oi];
      if (o.scalar || o.fold) continue;   // (folded templates: pull_folded)
      for (size_t s = 0; s < o.grad_idx.size(); ++s) {
        const IdxVal &iv = idx_[o.grad_idx[s]];
        if (!iv.ind.empty() || iv.aff.k[0] == 0) continue;
        fam[{(int64_t)iv.aff.space, iv.aff.k[0], iv.aff.k[1], iv.aff.k[2]}].push_back(Ref{(int)oi, (int)s, iv.aff.c, box_items(o)});
      }
    }
    const int64_t max_shift = 4;
    std::map<std::pair<int, int64_t>, std::vector<int>> moved;   // (output, shift) -> its slots to move
    for (auto &kv : fam) {
      const int64_t k0 = kv.first[1];
      auto &refs = kv.second;
      // destination at lane 0 of dim 0 (the other dims enter with the slot's own box: same k1, k2 for the whole family)
      std::vector<char> taken(refs.size(), 0);
      for (size_t a = 0; a < refs.size(); ++a) {
        if (taken[a]) continue;
        // cluster: slots within max_shift lanes of slot a's map (chains of neighbours stay in one cluster)
        std::vector<size_t> cl{a};
        taken[a] = 1;
        for (size_t grow = 0; grow < cl.size(); ++grow)
          for (size_t b = 0; b < refs.size(); ++b) {
            if (taken[b]) continue;
            const int64_t dc = refs[b].c - refs[cl[grow]].c;
            if (dc % k0 == 0 && std::llabs(dc / k0) <= max_shift) { taken[b] = 1; cl.push_back(b); }
          }
        std::map<int64_t, int64_t> weight;   // c -> items using it
        for (size_t i : cl) weight[refs[i].c] += refs[i].items;
        if (weight.size() < 2) continue;
        int64_t canon = weight.begin()->first;
        for (auto &w : weight) if (w.second > weight[canon] || (w.second == weight[canon] && w.first > canon)) canon = w.first;
        for (size_t i : cl) {
          if (refs[i].c == canon) continue;
          const int64_t dq = (refs[i].c - canon) / k0;   // the slot's entry belongs to lane q + dq
          const Output &o = outs_[refs[i].out];
          if (o.qlo[0] + dq < 0 || o.qhi[0] + dq > g_.ext[0]) continue;   // the clone would leave the launch domain
          moved[{refs[i].out, dq}].push_back(refs[i].slot);
        }
      }
    }
    for (auto &mv : moved) {
      const int oi = mv.first.first;
      Output clone = make_output(outs_[oi].tpl, false, mv.first.second, nullptr);
      if (clone.vals.size() != outs_[oi].vals.size()) throw std::runtime_error("internal: pulled clone differs in shape");
      std::vector<char> keep(clone.vals.size(), 0);
      for (int s : mv.second) { keep[s] = 1; outs_[oi].grad_mode[s] = -1; }
      for (size_t s = 0; s < clone.vals.size(); ++s)
        if (!keep[s]) clone.grad_mode[s] = -1;
      outs_.push_back(std::move(clone));
    }
  }

  // Scatter kinds: slots of DIFFERENT templates of this kernel that hit the same destination from
  // the same lane (identical index value: x7[i] is read by six quadrotor templates) are summed in
  // registers, in template order, and leave through ONE slot — the one whose item box contains
  // the others (their values masked by their own guards).  One store or one atomic per lane and
  // entry instead of several: what remains for an entry is at most the adds of DIFFERENT lanes
  // (a backward-difference neighbour: two addends, whose order cannot change the sum).
  void merge_scatter() {
    std::map<int, std::vector<std::pair<int, int>>> by_dest;   // destination IdxVal id -> (output, slot)
    for (size_t oi = 0; oi < outs_.size(); ++oi) {
      if (outs_[oi].scalar) continue;
      for (size_t s = 0; s < outs_[oi].grad_idx.size(); ++s)
        if (idx_[outs_[oi].grad_idx[s]].ind.empty() && outs_[oi].grad_mode[s] >= 0) by_dest[outs_[oi].grad_idx[s]].emplace_back((int)oi, (int)s);
    }
    for (auto &kv : by_dest) {
      auto &parts = kv.second;
      if (parts.size() < 2) continue;
      auto inside = [&](const Output &a, const Output &h) {
        for (int d = 0; d < 3; ++d) if (a.qlo[d] < h.qlo[d] || a.qhi[d] > h.qhi[d]) return false;
        return true;
      };
      int host = -1;   // the part whose box contains every other part's
      for (size_t c = 0; c < parts.size() && host < 0; ++c) {
        if (outs_[parts[c].first].fold == 2) continue;   // a pinned clone stores on one lane in fold_n: never the carrier of others
        bool all = true;
        for (auto &p : parts) if (!inside(outs_[p.first], outs_[parts[c].first])) all = false;
        if (all) host = (int)c;
      }
      if (host < 0) {
        // no part contains the others (a difference row and its pulled clone: lanes [1, S) and [0, S-1)): when the
        // boxes agree in dims 1, 2 and their dim-0 intervals form ONE interval, a new output on the union carries the sum
        const Output &P0 = outs_[parts[0].first];
        bool ok = !g_.flat;
        std::vector<std::pair<int64_t, int64_t>> iv0;
        for (auto &p : parts) {
          const Output &o = outs_[p.first];
          for (int d = 1; d < 3; ++d) if (o.qlo[d] != P0.qlo[d] || o.qhi[d] != P0.qhi[d]) ok = false;
          iv0.emplace_back(o.qlo[0], o.qhi[0]);
        }
        std::sort(iv0.begin(), iv0.end());
        int64_t lo0 = iv0[0].first, hi0 = iv0[0].second;
        for (auto &r : iv0) { if (r.first > hi0) ok = false; hi0 = std::max(hi0, r.second); }
        if (!ok) continue;
        Output u;
        u.kind = kind_; u.tpl = P0.tpl; u.pos_idx = -1; u.scalar = false;
        for (int d = 0; d < 3; ++d) { u.qlo[d] = P0.qlo[d]; u.qhi[d] = P0.qhi[d]; }
        u.qlo[0] = lo0; u.qhi[0] = hi0;
        std::ostringstream os;
        os << "inb";
        for (int d = 0; d < g_.nd; ++d) {
          if (u.qlo[d] > 0) os << " && q" << d << " >= " << coefstr(u.qlo[d]);
          if (u.qhi[d] < g_.ext[d]) os << " && q" << d << " < " << ip(u.qhi[d]);
        }
        u.guard = guard_id(os.str());
        int acc = -1;
        for (auto &p : parts) {
          const Output &o = outs_[p.first];
          int v = mk(VGUARD, o.guard, o.vals[p.second], -1, -1, 0);
          acc = acc < 0 ? v : add(acc, v);
        }
        u.vals = {acc}; u.grad_idx = {kv.first}; u.grad_tidx = {-1}; u.grad_mode = {2};
        for (auto &p : parts) outs_[p.first].grad_mode[p.second] = -1;
        outs_.push_back(std::move(u));
        continue;
      }
      const Output &H = outs_[parts[host].first];
      int acc = -1;
      for (auto &p : parts) {
        const Output &o = outs_[p.first];
        int v = o.vals[p.second];
        if (o.guard != H.guard) v = mk(VGUARD, o.guard, v, -1, -1, 0);
        acc = acc < 0 ? v : add(acc, v);
      }
      for (size_t c = 0; c < parts.size(); ++c) {
        Output &o = outs_[parts[c].first];
        if ((int)c == host) o.vals[parts[c].second] = acc;
        else o.grad_mode[parts[c].second] = -1;   // folded into the host slot
      }
    }
    // several single-item templates adding into one entry (the terms of a first-stage cost): one sum, one slot
    {
      std::map<int, std::vector<std::pair<int, int>>> sc;
      for (size_t oi = 0; oi < outs_.size(); ++oi) {
        if (!outs_[oi].scalar) continue;
        for (size_t s = 0; s < outs_[oi].grad_idx.size(); ++s)
          if (idx_[outs_[oi].grad_idx[s]].ind.empty() && outs_[oi].grad_mode[s] >= 0) sc[outs_[oi].grad_idx[s]].emplace_back((int)oi, (int)s);
      }
      for (auto &kv : sc) {
        auto &parts = kv.second;
        if (parts.size() < 2) continue;
        int acc = -1;
        for (auto &p : parts) {
          int v = outs_[p.first].vals[p.second];
          if (outs_[p.first].guard != outs_[parts[0].first].guard) v = mk(VGUARD, outs_[p.first].guard, v, -1, -1, 0);
          acc = acc < 0 ? v : add(acc, v);
        }
        outs_[parts[0].first].vals[parts[0].second] = acc;
        for (size_t c = 1; c < parts.size(); ++c) outs_[parts[c].first].grad_mode[parts[c].second] = -1;
      }
    }
    // single-item templates (point constraints x_k(0) == 0) run on the lane of grid point 0: fold their
    // contribution into the slot of that lane that hits the same entry
    for (size_t oi = 0; oi < outs_.size(); ++oi) {
      if (!outs_[oi].scalar) continue;
      for (size_t s = 0; s < outs_[oi].grad_idx.size(); ++s) {
        const IdxVal &iv = idx_[outs_[oi].grad_idx[s]];
        if (!iv.ind.empty() || outs_[oi].grad_mode[s] < 0) continue;
        for (size_t hj = 0; hj < outs_.size(); ++hj) {
          Output &h = outs_[hj];
          if (h.scalar || h.fold == 2) continue;
          bool has0 = true;
          for (int d = 0; d < 3; ++d) if (h.qlo[d] > 0 || h.qhi[d] < 1) has0 = false;
          if (!has0) continue;
          bool done_ = false;
          for (size_t hs = 0; hs < h.grad_idx.size() && !done_; ++hs) {
            const IdxVal &hv = idx_[h.grad_idx[hs]];
            if (h.grad_mode[hs] < 0 || !hv.ind.empty() || hv.aff.c != iv.aff.c) continue;
            if (g_.fold_n > 0 && ((g_.nd == 1 && hv.aff.k[1] != 0) || hv.aff.k[2] != 0)) continue;   // (derived coordinates are not 0 on lane 0)
            h.vals[hs] = add(h.vals[hs], mk(VGUARD, outs_[oi].guard, outs_[oi].vals[s], -1, -1, 0));
            outs_[oi].grad_mode[s] = -1;
            done_ = true;
          }
          if (done_) break;
        }
      }
    }
  }

  // Opt-in merged Hessian layout: templates of this kernel that cover the same lanes form a
  // class; slots of a class with the same unordered index pair always hit the same (row, col),
  // so the lane sums them in registers and the class owns one contiguous block
  // [o, o + n_items*nslots).  Not ExaModels' layout: for solvers that only need a consistent
  // hess_structure!/hess_coord! pair.
  void merge_hess(int64_t &o2m, std::vector<HessClass> &classes) {
    std::vector<Output> merged;
    std::vector<char> used(outs_.size(), 0);
    for (size_t a = 0; a < outs_.size(); ++a) {
      if (used[a]) continue;
      Output m = outs_[a];
      m.vals.clear(); m.slot_ia.clear(); m.slot_ib.clear(); m.slot_ti.clear(); m.slot_tj.clear();
      std::map<std::pair<int, int>, int> slot_of;
      for (size_t b = a; b < outs_.size(); ++b) {
        const Output &ob = outs_[b];
        if (used[b] || ob.guard != outs_[a].guard || ob.scalar != outs_[a].scalar) continue;
        if (ob.scalar && b != a) continue;   // single-item templates stay separate
        used[b] = 1;
        for (size_t s = 0; s < ob.vals.size(); ++s) {
          auto key = std::make_pair(std::min(ob.slot_ia[s], ob.slot_ib[s]), std::max(ob.slot_ia[s], ob.slot_ib[s]));
          auto it = slot_of.find(key);
          if (it == slot_of.end()) {
            slot_of.emplace(key, (int)m.vals.size());
            m.vals.push_back(ob.vals[s]);
            m.slot_ia.push_back(ob.slot_ia[s]); m.slot_ib.push_back(ob.slot_ib[s]);
            // representative index expressions, re-expressed in the FIRST template's tables is not
            // possible in general, so each slot remembers the template it came from
            m.slot_ti.push_back((int)b * 65536 + ob.slot_ti[s]); m.slot_tj.push_back((int)b * 65536 + ob.slot_tj[s]);
          } else {
            m.vals[it->second] = add(m.vals[it->second], ob.vals[s]);
          }
        }
      }
      const Template &t = m_.tpl[m.tpl];
      HessClass hc;
      hc.tpl = m.tpl; hc.n_items = m.scalar ? 1 : t.n_items; hc.o = o2m;
      for (size_t s = 0; s < m.vals.size(); ++s) {
        hc.idx_i.push_back(m.slot_ti[s]); hc.idx_j.push_back(m.slot_tj[s]);
      }
      // the class position as a function of q: same item box as its first template
      TGeo G = geo(m.tpl, m.scalar);
      IdxVal iv; iv.aff = klin_aff(t, G, (int64_t)m.vals.size(), o2m);
      m.pos_idx = idxval(iv); m.pos_off = o2m;
      o2m += hc.n_items * (int64_t)m.vals.size();
      alg_w_ += hc.n_items * (int64_t)m.vals.size();
      // resolve (output index, idx id) → (template id, idx id)
      for (size_t s = 0; s < hc.idx_i.size(); ++s) {
        int ob_i = hc.idx_i[s] / 65536, ob_j = hc.idx_j[s] / 65536;
        hc.idx_i[s] = outs_[ob_i].tpl * 65536 + hc.idx_i[s] % 65536;
        hc.idx_j[s] = outs_[ob_j].tpl * 65536 + hc.idx_j[s] % 65536;
      }
      classes.push_back(hc);
      merged.push_back(std::move(m));
    }
    outs_ = std::move(merged);
  }

  // Deterministic shared-entry reduction of this kernel's kind (grad / jtprod / hprod; grad_mode 3):
  // filled in by generate() once every slot of the kind is classified.
  struct SharedInfo {
    bool on = false;
    std::vector<std::tuple<int, int, int>> mine;       // (output, slot, value id) parked by THIS kernel
    int64_t nv_total = 0, n_wg = 0, red_off = 0;        // values / workgroups of the whole call; this kernel's first workgroup
    std::vector<std::pair<int64_t, int64_t>> owner;     // per value id: {first workgroup, count} that park it
    std::vector<std::pair<int64_t, std::vector<int>>> dests;   // destination entry (0-based) -> value ids summed into it
  };
  void set_shared(const SharedInfo &si) { shared_ = si; }
  // LDS doubles the reduction needs: park uses NV_mine x waves, the last workgroup NV_total, + the flag
  int shared_lds_doubles() const {
    if (!shared_.on) return 0;
    return (int)std::max<int64_t>((int64_t)shared_.mine.size() * (opt_.block / 64), shared_.nv_total) + 1;
  }
  // tables of the last workgroup's epilogue: {ticket offset, n_wg, offs[nv], first[nv], count[nv], ND x {dest, start, cnt}, value ids}
  static std::vector<int64_t> shared_final_table(const SharedInfo &si) {
    std::vector<int64_t> t;
    t.push_back(si.nv_total * si.n_wg);   // ticket words sit behind the parked values
    t.push_back(si.n_wg);
    for (int64_t v = 0; v < si.nv_total; ++v) t.push_back(v * si.n_wg);
    for (auto &o : si.owner) t.push_back(o.first);
    for (auto &o : si.owner) t.push_back(o.second);
    int64_t start = 0;
    for (auto &d : si.dests) { t.push_back(d.first); t.push_back(start); t.push_back((int64_t)d.second.size()); start += (int64_t)d.second.size(); }
    for (auto &d : si.dests) for (int v : d.second) t.push_back(v);
    return t;
  }
  // `base`: text of the table's first element (e.g. "A.ip + 17"); `wg`: text of the workgroup's id within the call
  static std::string shared_epilogue(const SharedInfo &si, const std::string &base, const std::string &wg, const std::string &lds, int lds_doubles) {
    std::ostringstream e;
    const int64_t nv = si.nv_total, nd = (int64_t)si.dests.size();
    e << "  { const long long* st_ = " << base << ";\n"
      << "    if (iem_shared_last(AUX + st_[0], " << wg << ", st_[1], " << lds << " + " << (lds_doubles - 1) << ")) {\n"
      << "      iem_shared_totals(" << nv << ", AUX, st_ + 2, st_ + " << (2 + nv) << ", st_ + " << (2 + 2 * nv) << ", " << lds << ");\n"
      << "      iem_shared_write(OUT, " << lds << ", st_ + " << (2 + 3 * nv) << ", st_ + " << (2 + 3 * nv + 3 * nd) << ", " << nd << ");\n"
      << "    } }\n";
    return e.str();
  }
  const SharedInfo &shared() const { return shared_; }

  // gradient store classification (needs every objective slot of the model)
  std::vector<Output> &outputs() { return outs_; }
  const std::vector<IdxVal> &idxvals() const { return idx_; }
  const std::vector<ILoad> &iloads() const { return iloads_; }
  const std::vector<int> &ia_arrays() const { return iav_; }
  const Group &group() const { return g_; }
  // after emit(): index ranges of every array the kernel's live loads touch (key: 0 x, 1 theta, 2 y, 4 v, 100 + slot item
  // columns by MODEL array id) — generate() takes the union over the bodies of one launch, so that an input two bodies
  // load is counted once in the launch's algorithmic bytes
  std::map<int, std::vector<std::pair<int64_t, int64_t>>> read_ranges() const {
    std::map<int, std::vector<std::pair<int64_t, int64_t>>> out;
    for (auto &kv : ranges_) {
      const int key = kv.first >= 100 ? 100 + fav_[kv.first - 100] : kv.first;   // local fa slot -> model array id
      auto &dst = out[key];
      dst.insert(dst.end(), kv.second.begin(), kv.second.end());
    }
    return out;
  }
  int64_t iload_elems() const { return iload_elems_; }

  // ---- emission -----------------------------------------------------------------
  std::string aff_str(const AffQ &a) {
    std::ostringstream os;
    bool any = false;
    if (g_.flat && g_.nd == 2 && a.k[0] != 0 && a.k[1] == a.k[0] * g_.ext[0] && a.k[2] == 0) {
      // a slab walked at full stride on a flat 2-D grid: c + k*(q0 + E0*q1) is c + k*q — no 64-bit multiply per index
      if (a.c != 0) os << ip(a.c, a.space) << " + ";
      if (a.k[0] == 1) os << "q"; else os << coefstr(a.k[0]) << " * q";
      return os.str();
    }
    if (a.c != 0) { os << ip(a.c, a.space); any = true; }
    for (int d = 0; d < 3; ++d) {
      if (a.k[d] == 0) continue;
      if (any) os << " + ";
      if (a.k[d] == 1) os << "q" << d;
      else os << coefstr(a.k[d]) << " * q" << d;
      any = true;
    }
    if (!any) os << "0LL";
    return os.str();
  }

  std::string guard_or(const std::set<int> &gs) {
    // weakest guard: plain `inb` subsumes every non-scalar guard
    int inb_id = -1;
    auto it = guard_ids_.find("inb");
    if (it != guard_ids_.end()) inb_id = it->second;
    if (inb_id >= 0 && gs.count(inb_id)) return "g" + std::to_string(inb_id);
    std::ostringstream os;
    bool first = true;
    for (int g : gs) { os << (first ? "" : " || ") << "g" << g; first = false; }
    return gs.size() > 1 ? "(" + os.str() + ")" : os.str();
  }

  std::string load_stmt(int i) {
    const Load &l = loads_[i];
    std::string arr = l.arr == 0 ? "X" : l.arr == 1 ? "TH" : l.arr == 2 ? "Y" : l.arr == 4 ? "V" : "FA[" + std::to_string(l.slot) + "]";
    return "const double l" + std::to_string(i) + " = " + guard_or(l.guards) + " ? " + arr + "[i" + std::to_string(l.idxval) + "] : 0.0;\n";
  }

  void emit_val(int id, std::ostringstream &os, std::vector<char> &done, const std::vector<char> &live) {
    if (done[id]) return;
    const VNode &n = v_[id];
    // iterative post-order
    std::vector<std::pair<int, int>> st;
    st.emplace_back(id, 0);
    while (!st.empty()) {
      int cur = st.back().first;
      int &phase = st.back().second;
      if (done[cur]) { st.pop_back(); continue; }
      const VNode &c = v_[cur];
      if (phase == 0) {
        phase = 1;
        if (c.op == VUN || c.op == VGUARD) { if (!done[c.a]) st.emplace_back(c.a, 0); }
        else if (c.op == VBIN) { if (!done[c.b]) st.emplace_back(c.b, 0); if (!done[c.a]) st.emplace_back(c.a, 0); }
        else if (c.op == VSEL) { if (!done[c.b]) st.emplace_back(c.b, 0); if (!done[c.a]) st.emplace_back(c.a, 0); }
        continue;
      }
      emit_one(cur, os, done, live);
      st.pop_back();
    }
    (void)n;
  }

  static const char *fn_name(int op) {
    switch (op) {
      case IEM_OP_SQRT: return "sqrt"; case IEM_OP_CBRT: return "cbrt"; case IEM_OP_ABS: return "fabs";
      case IEM_OP_EXP: return "exp"; case IEM_OP_EXP2: return "exp2"; case IEM_OP_LOG: return "log";
      case IEM_OP_LOG2: return "log2"; case IEM_OP_LOG10: return "log10"; case IEM_OP_LOG1P: return "log1p";
      case IEM_OP_SIN: return "sin"; case IEM_OP_COS: return "cos"; case IEM_OP_TAN: return "tan";
      case IEM_OP_ASIN: return "asin"; case IEM_OP_ACOS: return "acos"; case IEM_OP_ATAN: return "atan";
      case IEM_OP_SINH: return "sinh"; case IEM_OP_COSH: return "cosh"; case IEM_OP_TANH: return "tanh";
      case IEM_OP_ATANH: return "atanh";
    }
    return nullptr;
  }

  void emit_one(int id, std::ostringstream &os, std::vector<char> &done, const std::vector<char> &live) {
    const VNode &n = v_[id];
    std::string nm = "v" + std::to_string(id);
    switch (n.op) {
      case VC: os << "  const double " << nm << " = " << hexf(n.imm) << ";\n"; break;
      case VDP: os << "  const double " << nm << " = A.dp[" << n.sub << "];\n"; break;
      case VW: os << "  const double " << nm << " = A.w;\n"; break;
      case VLD:
        if (n.sub < (int)lazy_load_.size() && lazy_load_[n.sub] == 1) {   // first use: the load itself goes here, not into the head
          lazy_load_[n.sub] = 2;
          os << "  " << load_stmt(n.sub);
        }
        os << "  const double " << nm << " = l" << n.sub << ";\n";
        break;
      case VGUARD: os << "  const double " << nm << " = g" << n.sub << " ? v" << n.a << " : 0.0;\n"; break;
      case VSEL: {
        const auto &pr = sels_[n.sub];
        os << "  const double " << nm << " = (i" << pr.first << " == i" << pr.second << ") ? v" << n.a << " : v" << n.b << ";\n";
        break;
      }
      case VUN: {
        if (n.sub == IEM_OP_NEG) { os << "  const double " << nm << " = -v" << n.a << ";\n"; break; }
        if (n.sub == U_SGN) { os << "  const double " << nm << " = (v" << n.a << " >= 0.0) ? 1.0 : -1.0;\n"; break; }
        if (n.sub == IEM_OP_SIN || n.sub == IEM_OP_COS) {
          int other = n.sub == IEM_OP_SIN ? IEM_OP_COS : IEM_OP_SIN;
          uint64_t zb = 0;
          auto it = memo_.find(VKey{VUN, other, n.a, -1, -1, zb});
          if (it != memo_.end() && live[it->second] && !done[it->second]) {
            int oid = it->second;
            std::string s = n.sub == IEM_OP_SIN ? nm : "v" + std::to_string(oid);
            std::string c = n.sub == IEM_OP_SIN ? "v" + std::to_string(oid) : nm;
            if (opt_.ablate & 2) os << "  double " << s << " = v" << n.a << " * 0.5, " << c << " = v" << n.a << " * 0.25;\n";
            else os << "  double " << s << ", " << c << "; sincos(v" << n.a << ", &" << s << ", &" << c << ");\n";
            done[oid] = 1;
            break;
          }
        }
        const char *f = fn_name(n.sub);
        if (!f) throw std::runtime_error("codegen: no device function for opcode " + std::to_string(n.sub));
        if (opt_.ablate & 2) os << "  const double " << nm << " = v" << n.a << " * 0.75;\n";
        else os << "  const double " << nm << " = " << f << "(v" << n.a << ");\n";
        break;
      }
      case VBIN: {
        const char *o = n.sub == IEM_OP_ADD ? "+" : n.sub == IEM_OP_SUB ? "-" : n.sub == IEM_OP_MUL ? "*" : n.sub == IEM_OP_DIV ? "/" : nullptr;
        if (o) os << "  const double " << nm << " = v" << n.a << " " << o << " v" << n.b << ";\n";
        else os << "  const double " << nm << " = pow(v" << n.a << ", v" << n.b << ");\n";
        break;
      }
    }
    done[id] = 1;
  }

  void mark_live(int id, std::vector<char> &live) {
    std::vector<int> st{id};
    while (!st.empty()) {
      int c = st.back(); st.pop_back();
      if (live[c]) continue;
      live[c] = 1;
      const VNode &n = v_[c];
      if (n.op == VUN || n.op == VGUARD) st.push_back(n.a);
      else if (n.op == VBIN || n.op == VSEL) { st.push_back(n.a); st.push_back(n.b); }
    }
  }

  // as_body: emit a __device__ function `<name>_body` for a multi-group kernel (generate() writes the
  // __global__ wrapper that owns the LDS and decodes the workgroup id) instead of a kernel of its own
  std::string emit(KernelDesc &kd, bool as_body = false) {
    std::ostringstream body;
    std::vector<char> live(v_.size(), 0), done(v_.size(), 0);
    // Scatter kinds with many loads (the OPF's jtprod!: 36 inputs + 62 rows of v): all loads at the head of the
    // kernel keep ~200 VGPRs alive from the first instruction.  lazy_loads 1: the rows of v / y are loaded where
    // the template that uses them starts; 2: every load.
    lazy_load_.assign(loads_.size(), 0);
    if (opt_.lazy_loads > 0 && (opt_.lazy_all_kinds || kind_ == KK_GRAD || kind_ == KK_JTPROD || kind_ == KK_HPROD || kind_ == KK_JPROD) && (int)loads_.size() >= opt_.lazy_min_loads)
      for (size_t i = 0; i < loads_.size(); ++i)
        if (opt_.lazy_loads >= 2 || loads_[i].arr == 4 || loads_[i].arr == 2) lazy_load_[i] = 1;
    for (auto &o : outs_)
      for (size_t s = 0; s < o.vals.size(); ++s)
        if (o.grad_mode.size() != o.vals.size() || o.grad_mode[s] >= 0) mark_live(o.vals[s], live);   // not the slots merge_scatter folded away
    // which idx values must exist as variables: load positions (all), output positions, grad idx, sel operands
    std::set<int> need_idx;
    for (auto &l : loads_) need_idx.insert(l.idxval);
    for (auto &o : outs_) { if (o.pos_idx >= 0) need_idx.insert(o.pos_idx); for (int g : o.grad_idx) need_idx.insert(g); }
    for (auto &pr : sels_) { need_idx.insert(pr.first); need_idx.insert(pr.second); }

    // templates with more slots than LDS can stage (huge nonlinear rows) keep the direct form
    const int ns_cap = std::max(1, (opt_.store_mode == 1 ? 96 * 1024 / (8 * 64 * (opt_.block / 64)) : 96 * 1024 / (8 * opt_.block)));
    int max_ns = 1;
    for (auto &o : outs_)
      if ((kind_ == KK_JAC || kind_ == KK_HESS) && (int)o.vals.size() <= ns_cap) max_ns = std::max<int>(max_ns, (int)o.vals.size());
    // values per lane this kernel ever stages: a body with one small template (the side-by-side
    // shape of small grids) must not reserve the full staging batch — LDS decides how many
    // workgroups share a CU, and those kernels are latency-bound
    int total_ns = 0;
    for (auto &o : outs_) {
      const int ns = (kind_ == KK_JAC || kind_ == KK_HESS) ? (int)o.vals.size() : 1;
      if (!o.scalar && ns <= ns_cap) total_ns += ns;
    }
    bool use_lds = (kind_ == KK_JAC || kind_ == KK_HESS) && opt_.store_mode == 1;
    bool use_blk = (kind_ == KK_JAC || kind_ == KK_HESS || kind_ == KK_CONS || kind_ == KK_JPROD) && opt_.store_mode == 2;

    // stores/outputs first into `tail` so that every ip() they need is registered before the struct is printed
    std::ostringstream tail;
    if (kind_ == KK_OBJ) tail << "  double acc = 0.0;\n";
    // Emission order: cheapest outputs first (greedy on the incremental cost of the DAG
    // nodes still to be computed).  Blocks of one launch start in lockstep, so a kernel that
    // computes everything before its first store leaves HBM idle for the whole prologue of
    // the first round; emitting the templates with constant/affine partials first lets the
    // store stream start while the transcendental-heavy templates are still being computed.
    std::vector<int> order;
    if (opt_.reorder && (kind_ == KK_JAC || kind_ == KK_HESS || kind_ == KK_CONS || kind_ == KK_JPROD)) {
      std::vector<char> sim = done;
      std::vector<char> taken(outs_.size(), 0);
      for (size_t step = 0; step < outs_.size(); ++step) {
        long best_cost = -1;
        int best = -1;
        for (size_t oi = 0; oi < outs_.size(); ++oi) {
          if (taken[oi]) continue;
          long c = 0;
          std::vector<int> st(outs_[oi].vals.begin(), outs_[oi].vals.end());
          std::vector<int> seen;
          while (!st.empty()) {
            int id = st.back(); st.pop_back();
            if (sim[id]) continue;
            sim[id] = 2; seen.push_back(id);
            const VNode &n = v_[id];
            if (n.op == VUN) { c += (n.sub == IEM_OP_NEG || n.sub == U_SGN) ? 1 : 40; st.push_back(n.a); }
            else if (n.op == VBIN) { c += n.sub == IEM_OP_DIV ? 12 : n.sub == IEM_OP_POW ? 80 : 1; st.push_back(n.a); st.push_back(n.b); }
            else if (n.op == VSEL) { c += 1; st.push_back(n.a); st.push_back(n.b); }
            else if (n.op == VLD) c += 4;
          }
          for (int id : seen) sim[id] = 0;
          if (best < 0 || c < best_cost) { best = (int)oi; best_cost = c; }
        }
        taken[best] = 1;
        order.push_back(best);
        std::vector<int> st(outs_[best].vals.begin(), outs_[best].vals.end());
        while (!st.empty()) {
          int id = st.back(); st.pop_back();
          if (sim[id]) continue;
          sim[id] = 1;
          const VNode &n = v_[id];
          if (n.op == VUN) st.push_back(n.a);
          else if (n.op == VBIN || n.op == VSEL) { st.push_back(n.a); st.push_back(n.b); }
        }
      }
    } else {
      for (size_t oi = 0; oi < outs_.size(); ++oi) order.push_back((int)oi);
    }
    int batch_slots = 0;
    const int lds_budget = stage_budget(max_ns, total_ns);
    std::vector<std::string> pending_flush;
    auto flush_batch = [&]() {
      if (pending_flush.empty()) return;
      tail << "  __syncthreads();\n";
      for (auto &f : pending_flush) tail << f;
      if (opt_.store_wait) tail << "  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n";
      tail << "  __syncthreads();\n";
      pending_flush.clear();
      batch_slots = 0;
    };
    for (int oi : order) {
      auto &o = outs_[oi];
      const Template &t = m_.tpl[o.tpl];
      tail << "  // template " << o.tpl << " (" << (t.kind == IEM_T_OBJ ? "objective" : "constraint") << ", " << o.vals.size() << " value(s))\n";
      for (size_t s = 0; s < o.vals.size(); ++s)
        if (o.grad_mode.size() != o.vals.size() || o.grad_mode[s] >= 0) emit_val(o.vals[s], tail, done, live);
      std::string g = "g" + std::to_string(o.guard);
      switch (kind_) {
        case KK_JPROD:
        case KK_CONS:
          if (!(opt_.store_mode == 2 && !o.scalar)) {
            tail << "  if (" << g << ") OUT[i" << o.pos_idx << "] = v" << o.vals[0] << ";\n";
            break;
          }
          [[fallthrough]];   // store_mode 2: rows go out through the aligned block store (1 value per lane)
        case KK_JAC:
        case KK_HESS: {
          int ns = (int)o.vals.size();
          bool scalar_tpl = o.scalar;
          if (ns > ns_cap) scalar_tpl = true;   // direct strided stores for this template
          bool full_box = true;
          for (int d = 0; d < g_.nd; ++d) if (o.qlo[d] != 0 || o.qhi[d] != g_.ext[d]) full_box = false;
          const bool by_ordinal = g_.flat && !full_box && g_.nd == 2 && opt_.store_mode == 2 && !scalar_tpl;
          if (g_.flat && !full_box && !by_ordinal) scalar_tpl = true;   // item ordinal is not linear in the flat lane index
          if (by_ordinal) {
            // sub-box template on a flat 2-D grid: stage by item ordinal (iem_device.h: iem_stage_ord / iem_flush_ord)
            if (batch_slots + ns > lds_budget) flush_batch();
            const std::string rn = "r" + std::to_string(oi), on = "ob" + std::to_string(oi);
            const int64_t w0 = o.qhi[0] - o.qlo[0], h1 = o.qhi[1] - o.qlo[1];
            const std::string box = ip(o.qlo[0]) + ", " + ip(w0) + ", " + ip(o.qlo[1]) + ", " + ip(h1);
            tail << "  const double " << rn << "[" << ns << "] = {";
            for (int s = 0; s < ns; ++s) tail << (s ? ", " : "") << "v" << o.vals[s];
            tail << "};\n";
            tail << "  const long long " << on << " = iem_ord_lt(fr1_, fr0_, " << box << ");\n";
            tail << "  const long long sl" << oi << " = (q0 - " << ip(o.qlo[0]) << ") + " << ip(w0) << " * (q1 - " << ip(o.qlo[1]) << ") - " << on << ";\n";
            tail << "  iem_stage_ord<" << ns << ">(" << rn << ", lds_blk + " << (batch_slots * opt_.block) << ", sl" << oi << ", " << g << ");\n";
            std::ostringstream fl;
            fl << "  { long long r1_, r0_;\n"
               << "    iem_flat_split(fr1_, fr0_, 16, " << ip(g_.ext[0]) << ", r1_, r0_); const long long o16 = iem_ord_lt(r1_, r0_, " << box << ");\n"
               << "    iem_flat_split(fr1_, fr0_, " << qstep_str() << ", " << ip(g_.ext[0]) << ", r1_, r0_); const long long ou = iem_ord_lt(r1_, r0_, " << box << ");\n"
               << "    iem_flat_split(fr1_, fr0_, IEM_TILE, " << ip(g_.ext[0]) << ", r1_, r0_); const long long oa = iem_ord_lt(r1_, r0_, " << box << ");\n"
               << "    iem_flush_ord<" << ns << ", " << qstep_str() << ">(OUT, " << ip(o.pos_off, 3) << ", " << on << ", o16, ou, oa, lds_blk + "
               << (batch_slots * opt_.block) << ", sl" << oi << ", " << g << "); }\n";
            pending_flush.push_back(fl.str());
            batch_slots += ns;
            break;
          }
          if (opt_.store_mode == 2 && !scalar_tpl) {
            // stage now, flush with the rest of the batch (one barrier pair per batch)
            if (batch_slots + ns > lds_budget) flush_batch();
            std::string rn = "r" + std::to_string(oi);
            tail << "  const double " << rn << "[" << ns << "] = {";
            for (int s = 0; s < ns; ++s) tail << (s ? ", " : "") << "v" << o.vals[s];
            tail << "};\n";
            tail << "  iem_stage<" << ns << ">(" << rn << ", lds_blk + " << (batch_slots * opt_.block) << ");\n";
            // block-uniform position of lane 0 / slot 0 and the valid lane interval of this workgroup
            const IdxVal &pv = idx_[o.pos_idx];
            AffQ pb = pv.aff;
            std::ostringstream gb, fl;
            bool any = false;
            for (int d = 1; d < g_.nd; ++d) {
              if (o.qlo[d] > 0) { gb << (any ? " && " : "") << "q" << d << " >= " << coefstr(o.qlo[d]); any = true; }
              // a folded second dimension (blockIdx.y + gridDim.y*blockIdx.z) overshoots ext[1]
              bool folded = d == 1 && g_.nd == 2 && g_.ext[1] > 65535;
              if (o.qhi[d] < g_.ext[d] || folded) { gb << (any ? " && " : "") << "q" << d << " < " << ip(std::min(o.qhi[d], g_.ext[d])); any = true; }
            }
            int64_t k0 = pb.k[0];
            pb.k[0] = 0;
            if (g_.flat) {
              // full-box template in a flat group: item ordinal == flat lane index
              fl << "  { const long long pb = " << ip(pv.aff.c, pv.aff.space) << " + " << coefstr(k0) << " * qb0;\n";
              fl << "    const int v0 = 0;\n";
              fl << "    const int v1 = iem_clamp256(" << ip(g_.ext[0] * g_.ext[1] * g_.ext[2]) << " - qb0);\n";
            } else {
            fl << "  { const long long pb = " << aff_str(pb) << " + " << coefstr(k0) << " * qb0;\n";
            fl << "    const int v0 = iem_clamp256(" << coefstr(o.qlo[0]) << " - qb0);\n";
            fl << "    const int v1 = " << (any ? "(" + gb.str() + ") ? " : "") << "iem_clamp256(" << ip(std::min(o.qhi[0], g_.ext[0])) << " - qb0)"
               << (any ? " : v0" : "") << ";\n";
            }
            fl << "    iem_flush<" << ns << ", " << qstep_str() << ">(OUT, pb, v0, v1, qb0 <= " << coefstr(g_.flat ? 0 : o.qlo[0]) << ", lds_blk + "
               << (batch_slots * opt_.block) << "); }\n";
            pending_flush.push_back(fl.str());
            batch_slots += ns;
            break;
          }
          tail << "  { const double r[" << ns << "] = {";
          for (int s = 0; s < ns; ++s) tail << (s ? ", " : "") << "v" << o.vals[s];
          tail << "};\n";
          if (use_lds && !scalar_tpl) tail << "    iem_store_rows<" << ns << ">(OUT, i" << o.pos_idx << ", " << g << ", r, lds_wave); }\n";
          else tail << "    iem_store_rows_direct<" << ns << ">(OUT, i" << o.pos_idx << ", " << g << ", r); }\n";
          break;
        }
        case KK_OBJ:
          tail << "  acc += " << g << " ? v" << o.vals[0] << " : 0.0;\n";
          break;
        case KK_JTPROD:
        case KK_HPROD:
        case KK_GRAD:
          for (size_t s = 0; s < o.vals.size(); ++s) {
            int mode = o.grad_mode[s];
            if (mode == 3 || mode < 0) continue;   // parked below (deterministic shared-entry reduction) / folded into another slot
            if (mode == 0) tail << "  if (" << g << ") OUT[i" << o.grad_idx[s] << "] = v" << o.vals[s] << ";\n";
            else if (mode == 5) {   // parked for the plan-driven gather: one slot of the aux buffer per lane of the launch domain
              if (o.scalar) tail << "  if (" << g << ") AUX[" << ip(o.axis_off.at((int)s), 6) << "] = v" << o.vals[s] << ";\n";
              else if (g_.flat) tail << "  if (" << g << ") AUX[" << ip(o.axis_off.at((int)s), 6) << " + q] = v" << o.vals[s] << ";\n";
              else if (g_.fold_n > 0 && g_.nd == 1) tail << "  if (" << g << ") AUX[" << ip(o.axis_off.at((int)s), 6) << " + q0] = v" << o.vals[s] << ";\n";
              else if (g_.fold_n > 0) tail << "  if (" << g << ") AUX[" << ip(o.axis_off.at((int)s), 6) << " + q0 + " << ip(g_.ext[0], 6) << " * q1] = v" << o.vals[s] << ";\n";
              else tail << "  if (" << g << ") AUX[" << ip(o.axis_off.at((int)s), 6) << " + q0 + " << ip(g_.ext[0], 6) << " * (q1 + " << ip(g_.ext[1], 6) << " * q2)] = v" << o.vals[s] << ";\n";
            }
            else if (mode == 4) {
              // row = position in dims 1, 2 of this output's box, lane = position in dim 0
              const int64_t n0 = o.qhi[0] - o.qlo[0], n1 = o.qhi[1] - o.qlo[1];
              if (g_.fold_n > 0)      // (q2 is the node of the lane there, not a grid coordinate: the grid has two dimensions)
                tail << "  if (" << g << ") AUX[" << ip(o.axis_off.at((int)s), 6) << " + (q1 - " << coefstr(o.qlo[1]) << ") * " << ip(n0, 6) << " + (q0 - "
                     << coefstr(o.qlo[0]) << ")] = v" << o.vals[s] << ";\n";
              else
              tail << "  if (" << g << ") AUX[" << ip(o.axis_off.at((int)s), 6) << " + ((q1 - " << coefstr(o.qlo[1]) << ") + " << ip(n1, 6) << " * (q2 - "
                   << coefstr(o.qlo[2]) << ")) * " << ip(n0, 6) << " + (q0 - " << coefstr(o.qlo[0]) << ")] = v" << o.vals[s] << ";\n";
            }
            else if (mode == 1) tail << "  iem_grad_wave_uniform(OUT, i" << o.grad_idx[s] << ", v" << o.vals[s] << ", " << g << ");\n";
            else tail << "  iem_grad_atomic(OUT, i" << o.grad_idx[s] << ", v" << o.vals[s] << ", " << g << ");\n";
          }
          break;
      }
    }
    flush_batch();
    if (kind_ == KK_OBJ) {
      if (!as_body) throw std::runtime_error("internal: objective bodies are always called from the tile-walking wrapper");
      tail << "  return acc;\n";
    }
    const std::string wg_txt = "(long long)blockIdx.x + (long long)gridDim.x * ((long long)blockIdx.y + (long long)gridDim.y * (long long)blockIdx.z)";
    if (shared_.on && !shared_.mine.empty()) {
      // this kernel's contributions to entries many items share: one value per lane each, reduced per
      // workgroup in a fixed order and parked for the last workgroup of the call (iem_device.h)
      std::vector<int64_t> offs;
      tail << "  { const double sh_[" << shared_.mine.size() << "] = {";
      bool first_v = true;
      for (auto &mv : shared_.mine) {
        const Output &o = outs_[std::get<0>(mv)];
        tail << (first_v ? "" : ", ") << "g" << o.guard << " ? v" << o.vals[std::get<1>(mv)] << " : 0.0";
        first_v = false;
        offs.push_back((int64_t)std::get<2>(mv) * shared_.n_wg);
      }
      const size_t ob = ip_block(offs);
      tail << "};\n    iem_shared_park<" << shared_.mine.size() << ">(sh_, AUX, A.ip + " << ob << ", " << ip(shared_.red_off) << " + " << wg_txt << ", lds_blk); }\n";
    }
    if (shared_.on && !as_body) {
      const size_t tb = ip_block(shared_final_table(shared_));
      tail << shared_epilogue(shared_, "A.ip + " + std::to_string(tb), ip(shared_.red_off) + " + " + wg_txt, "lds_blk", shared_lds_doubles());
    }

    // head: coordinates, guards, integer loads, index values, loads
    std::ostringstream head;
    if (!zero_fill_.empty()) {
      // fused memset of the output entries no template of this call writes (disjoint from every store)
      head << "  {\n    const long long nb_ = (long long)gridDim.x * gridDim.y * gridDim.z;\n"
           << "    const long long b_ = (long long)blockIdx.x + (long long)gridDim.x * ((long long)blockIdx.y + (long long)gridDim.y * blockIdx.z);\n";
      for (auto &z : zero_fill_) {
        head << "    iem_zero_fill(OUT + " << ip(z.first, 5) << ", " << ip(z.second - z.first, 5) << ", b_, nb_);\n";
        alg_w_ += z.second - z.first;
      }
      head << "  }\n";
    }
    if (g_.flat && g_.nd == 2) {
      // one division per WORKGROUP (block-uniform), then a 32-bit step per lane: (row, column) of the flat index
      head << "  const long long q = (long long)blockIdx.x * " << qstep_str() << " + threadIdx.x;\n";
      head << "  const bool inb = q < " << ip(g_.ext[0] * g_.ext[1]) << ";\n";
      head << "  const long long fqb_ = (long long)blockIdx.x * " << qstep_str() << ";\n";
      head << "  long long fr1_, fr0_; iem_flat_base(fqb_, " << ip(g_.ext[0]) << ", fr1_, fr0_);\n";
      head << "  long long q1, q0; iem_flat_split(fr1_, fr0_, (int)threadIdx.x, " << ip(g_.ext[0]) << ", q1, q0);\n";
      head << "  const long long q2 = 0; (void)fr1_; (void)fr0_;\n";
    } else if (g_.flat) {
      head << "  const long long q = (long long)blockIdx.x * " << qstep_str() << " + threadIdx.x;\n";
      head << "  const bool inb = q < " << ip(g_.ext[0] * g_.ext[1] * g_.ext[2]) << ";\n";
      head << "  const long long q0 = q % " << ip(g_.ext[0]) << ", qr = q / " << ip(g_.ext[0]) << ";\n";
      head << "  const long long q1 = qr % " << ip(g_.ext[1]) << ", q2 = qr / " << ip(g_.ext[1]) << ";\n";
    } else if (g_.fold_n > 0) {
      // element / node of the lane: one 64-bit division per WORKGROUP (block-uniform), 32-bit steps per lane
      const std::string n = std::to_string(g_.fold_n);
      head << "  const long long q0 = (long long)blockIdx.x * " << qstep_str() << " + threadIdx.x;\n";
      head << "  const long long fb_ = (long long)blockIdx.x * " << qstep_str() << " - " << ip(g_.fold_off) << ";\n";
      head << "  const long long fq_ = (fb_ >= 0 ? fb_ : fb_ - " << (g_.fold_n - 1) << ") / " << n << ";\n";
      head << "  const int ft_ = (int)(fb_ - " << n << " * fq_) + (int)threadIdx.x;\n";
      head << "  const long long qe = fq_ + ft_ / " << n << ", q2 = ft_ % " << n << "; (void)qe; (void)q2;\n";
      if (g_.nd == 1) {
        head << "  const long long q1 = qe; (void)q1;\n";      // (a 1-D grid has no second dimension: q1 carries the element)
        head << "  const bool inb = q0 < " << ip(g_.ext[0]) << ";\n";
      } else if (g_.ext[1] > 65535) {
        head << "  const long long q1 = (long long)blockIdx.y + (long long)gridDim.y * blockIdx.z;\n";
        head << "  const bool inb = q0 < " << ip(g_.ext[0]) << " && q1 < " << ip(g_.ext[1]) << ";\n";
      } else {
        head << "  const long long q1 = blockIdx.y;\n";
        head << "  const bool inb = q0 < " << ip(g_.ext[0]) << ";\n";
      }
    } else {
    head << "  const long long q0 = (long long)blockIdx.x * " << qstep_str() << " + threadIdx.x;\n";
    if (g_.nd == 2 && g_.ext[1] > 65535) {
      // second grid dimension longer than gridDim.y allows: fold it over blockIdx.z
      head << "  const long long q1 = (long long)blockIdx.y + (long long)gridDim.y * blockIdx.z, q2 = 0;\n";
      head << "  const bool inb = q0 < " << ip(g_.ext[0]) << " && q1 < " << ip(g_.ext[1]) << ";\n";
    } else {
      head << "  const long long q1 = blockIdx.y, q2 = blockIdx.z;\n";
      head << "  const bool inb = q0 < " << ip(g_.ext[0]) << ";\n";
    }
    }
    for (size_t gi = 0; gi < guards_.size(); ++gi) head << "  const bool g" << gi << " = " << guards_[gi] << ";\n";
    for (size_t i = 0; i < iloads_.size(); ++i) {
      head << "  const long long il" << i << " = " << guard_or(iloads_[i].guards) << " ? IA" << "[" << iloads_[i].ia_slot << "]["
           << aff_str(iloads_[i].pos) << "] : 0LL;\n";
    }
    for (int id : need_idx) {
      const IdxVal &iv = idx_[id];
      head << "  const long long i" << id << " = " << aff_str(iv.aff);
      for (auto &pr : iv.ind) head << " + " << coefstr(pr.first) << " * il" << pr.second;
      head << ";\n";
    }
    for (size_t i = 0; i < loads_.size(); ++i) {
      const Load &l = loads_[i];
      bool used = false;
      for (size_t v = 0; v < v_.size(); ++v) if (v_[v].op == VLD && v_[v].sub == (int)i && live[v]) { used = true; break; }
      if (!used) continue;
      if (!(i < lazy_load_.size() && lazy_load_[i])) head << "  " << load_stmt((int)i);
      alg_r_loads_++;
      // algorithmic read footprint: index range of this load over the launch domain
      const IdxVal &iv = idx_[l.idxval];
      int64_t lo = iv.aff.c, hi = iv.aff.c;
      for (int d = 0; d < 3; ++d) {
        int64_t clo, chi;
        coord_range(d, clo, chi);
        const int64_t e0 = iv.aff.k[d] * clo, e1 = iv.aff.k[d] * chi;
        lo += std::min(e0, e1); hi += std::max(e0, e1);
      }
      if (!iv.ind.empty()) { lo = 0; hi = g_.ext[0] * g_.ext[1] * g_.ext[2] - 1; }
      int akey = l.arr == 3 ? 100 + l.slot : l.arr;   // 0 x, 1 theta, 2 y, 4 v, 100+ item columns
      ranges_[akey].emplace_back(lo, hi);
      // what a sharded handle needs to know: can this kernel touch a halo entry of x (or of a variable-space v)?
      const bool var_space_v = l.arr == 4 && (kind_ == KK_JPROD || kind_ == KK_HPROD);
      if (l.arr == 0 || var_space_v) {
        auto &dst = l.arr == 0 ? kd.x_ranges : kd.v_ranges;
        if (iv.ind.empty()) dst.emplace_back(lo, hi);
        else dst.emplace_back(0, std::max<int64_t>(m_.nvar, 1) - 1);   // a gathered index: anywhere
      }
    }
    {
      int64_t elems = 0;
      for (auto &kv : ranges_) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end());
        int64_t cur_lo = v[0].first, cur_hi = v[0].second;
        for (size_t i = 1; i < v.size(); ++i) {
          if (v[i].first <= cur_hi + 1) cur_hi = std::max(cur_hi, v[i].second);
          else { elems += cur_hi - cur_lo + 1; cur_lo = v[i].first; cur_hi = v[i].second; }
        }
        elems += cur_hi - cur_lo + 1;
      }
      iload_elems_ = (int64_t)iloads_.size() * g_.ext[0] * g_.ext[1] * g_.ext[2];
      kd.alg_bytes_read = 8 * (elems + iload_elems_);
      if (kind_ == KK_GRAD || kind_ == KK_JTPROD || kind_ == KK_HPROD) {
        // scatter kinds: one write per item of every slot that is still its own (merge_scatter sums a lane's
        // addends into one slot; entries reduced by the last workgroup are a handful) plus the fused zero fill
        int64_t w = 0;
        for (auto &o : outs_) {
          int64_t n = 1;
          if (!o.scalar) for (int d = 0; d < 3; ++d) n *= std::max<int64_t>(o.qhi[d] - o.qlo[d], 1);
          for (size_t sl = 0; sl < o.vals.size(); ++sl)
            if (sl < o.grad_mode.size() && o.grad_mode[sl] >= 0 && o.grad_mode[sl] != 3) w += n;
        }
        for (auto &z : zero_fill_) w += z.second - z.first;
        alg_w_ = w;
      }
      kd.alg_bytes_written = 8 * alg_w_;
    }

    // assemble
    std::ostringstream os;
    size_t nip = std::max<size_t>(1, ipv_.size()), ndp = std::max<size_t>(1, dpv_.size());
    size_t nfa = std::max<size_t>(1, fav_.size()), nia = std::max<size_t>(1, iav_.size());
    // the kernel-argument segment is limited (4 KB): big tables move to device memory and the
    // struct carries pointers instead — `A.ip[i]` reads the same either way (uniform scalar loads)
    kd.tables_in_memory = (nip + ndp + nfa + nia) > 320;
    if (as_body) {
      os << "__device__ __forceinline__ " << (kind_ == KK_OBJ ? "double " : "void ") << name_ << "_body(const double* __restrict__ X, const double* __restrict__ TH, "
         << "const double* __restrict__ Y, const double* __restrict__ V, double* __restrict__ OUT, const double w_, double* __restrict__ AUX,\n"
         << "    const long long* ip_, const double* dp_, const double* const* FA, const long long* const* IA, double* lds_blk, double* lds4,\n"
         << "    const long long BX_, const long long BY_, const long long BZ_, const long long GX_, const long long GY_, const long long GZ_) {\n";
      os << "  const struct { const long long* ip; const double* dp; double w; } A = {ip_, dp_, w_};\n";
      os << "  (void)X; (void)TH; (void)Y; (void)V; (void)FA; (void)IA; (void)A; (void)AUX; (void)lds_blk; (void)lds4; (void)BY_; (void)BZ_; (void)GX_; (void)GY_; (void)GZ_;\n";
      kd.tables_in_memory = false;
    } else {
    os << "struct Args_" << name_ << " {\n  const double* x; const double* th; const double* y; const double* v; double* out; double w; double* aux; const IemHaloArgs* comm;\n"
       << "  double* p2; double* p3; double* p4; double* p5; double* p6;\n";
    if (kd.tables_in_memory)
      os << "  const long long* ip; const double* dp; const double* const* fa; const long long* const* ia;\n};\n";
    else
      os << "  long long ip[" << nip << "]; double dp[" << ndp << "]; const double* fa[" << nfa << "]; const long long* ia[" << nia << "];\n};\n";
    os << "extern \"C\" __global__ __launch_bounds__(IEM_TILE" << (opt_.min_waves > 0 ? ", " + std::to_string(opt_.min_waves) : std::string())
       << ") void " << name_ << "(const Args_" << name_ << " A) {\n";
    os << "  const double* __restrict__ X = A.x; const double* __restrict__ TH = A.th; const double* __restrict__ Y = A.y;\n";
    os << "  const double* __restrict__ V = A.v; (void)V;\n";
    os << "  double* __restrict__ OUT = A.out; double* __restrict__ AUX = A.aux; (void)AUX;\n";
    os << "  const double* const* FA = A.fa; const long long* const* IA = A.ia;\n";
    os << "  (void)X; (void)TH; (void)Y; (void)FA; (void)IA;\n";
    kd.carries = carrier();
    if (carrier()) {
      // a pending asynchronous halo exchange rides on this launch: one EXTRA leading workgroup (column 0 of the grid;
      // one per row on 2-D / 3-D grids, only the first works) runs it while the others evaluate — they see workgroup
      // column blockIdx.x - 1.  A.comm is null on every other launch (and on handles that are not sharded).
      os << "  const long long cb_ = A.comm != nullptr ? 1 : 0;\n"
         << "  if (cb_ && blockIdx.x == 0) { if (blockIdx.y == 0 && blockIdx.z == 0) iem_halo_wg(*A.comm, const_cast<double*>(A.x)); return; }\n"
         << "  const long long BX_ = (long long)blockIdx.x - cb_;\n";
    }
    if (opt_.xcd_remap) {
      // (with a carried halo exchange the evaluating workgroups are columns 1 .. of the launch: workgroups that share
      // b & 7 still share an XCD on a 1-D grid — the labels rotate by one)
      const std::string bx = carrier() ? "BX_" : "(long long)blockIdx.x", gx = carrier() ? "((long long)gridDim.x - cb_)" : "(long long)gridDim.x";
      os << "  const long long GX_ = " << gx << ", GY_ = gridDim.y, GZ_ = gridDim.z;\n"
         << "  const long long L_ = iem_xcd_remap(" << bx << " + GX_ * ((long long)blockIdx.y + GY_ * (long long)blockIdx.z), GX_ * GY_ * GZ_);\n"
         << "  const long long RX_ = L_ % GX_, BY_ = (L_ / GX_) % GY_, BZ_ = L_ / (GX_ * GY_); (void)BY_; (void)BZ_; (void)GZ_;\n";
    }
    }
    if (use_lds) {
      if (as_body) os << "  double* lds_all = lds_blk;\n";
      else os << "  __shared__ double lds_all[" << (opt_.block * max_ns) << "];\n";
      os << "  double* lds_wave = lds_all + iem_wave() * " << (64 * max_ns) << ";\n";
      kd.lds_bytes = opt_.block * max_ns * 8;
    }
    if (use_blk) {
      const int budget = stage_budget(max_ns, total_ns);
      if (!as_body) os << "  __shared__ double lds_blk[" << (opt_.block * budget) << "];\n";
      os << "  const long long qb0 = (long long)blockIdx.x * " << qstep_str() << ";\n";
      kd.lds_bytes = opt_.block * budget * 8;
    }
    if (shared_.on) {
      if (!as_body) os << "  __shared__ double lds_blk[" << shared_lds_doubles() << "];\n";
      kd.lds_bytes = shared_lds_doubles() * 8;
    }
    os << head.str() << tail.str() << "}\n\n";
    if (!as_body && !opt_.xcd_remap && carrier()) {
      std::string t = os.str();
      const std::string from = "blockIdx.x", to = "BX_";
      for (size_t pos = t.find("const long long BX_ = (long long)blockIdx.x - cb_;\n") + 52; (pos = t.find(from, pos)) != std::string::npos; pos += to.size()) t.replace(pos, from.size(), to);
      kd.ip = ipv_; kd.dp = dpv_; kd.fa = fav_; kd.ia = iav_;
      return t;
    }
    if (as_body || opt_.xcd_remap) {
      // the body sees LOGICAL workgroup coordinates: of its own grid, decoded by the wrapper
      // (as_body), and/or remapped so that neighbours share an XCD (xcd_remap)
      std::string t = os.str();
      const size_t keep = as_body ? 0 : t.find("(void)GZ_;\n");   // the remap prologue itself reads the hardware ids
      auto subst = [&](const std::string &from, const std::string &to) {
        for (size_t pos = keep; (pos = t.find(from, pos)) != std::string::npos; pos += to.size()) t.replace(pos, from.size(), to);
      };
      subst("blockIdx.x", as_body ? "BX_" : "RX_"); subst("blockIdx.y", "BY_"); subst("blockIdx.z", "BZ_");
      subst("gridDim.x", "GX_"); subst("gridDim.y", "GY_"); subst("gridDim.z", "GZ_");
      kd.ip = ipv_; kd.dp = dpv_; kd.fa = fav_; kd.ia = iav_;
      return t;
    }
    kd.ip = ipv_; kd.dp = dpv_; kd.fa = fav_; kd.ia = iav_;
    return os.str();
  }

  // values per lane staged per barrier pair: `lds_slots` is quoted for 256-thread workgroups
  // (2 KB of LDS per slot) and scaled so the LDS per workgroup stays the same for other sizes
  // Grid points per workgroup along the first dimension.  Kernels that write COO / row blocks
  // through the block store overlap their tiles by 16 lanes: the halo items are computed twice so
  // that every 128-byte line of a block is written WHOLE by one workgroup (iem_flush), instead
  // of two workgroups each writing a part of the line at every seam.
  // kinds whose kernels can carry a pending halo exchange as an extra leading workgroup: the block-store kinds (their
  // bodies never look at gridDim.x); the objective's and the pair's wrappers add theirs in generate()
  bool carrier() const {
    return opt_.carrier && (kind_ == KK_CONS || kind_ == KK_JAC || kind_ == KK_HESS || kind_ == KK_JPROD);
  }
  // templates of this builder whose output values are pure ITEM DATA: constants and item columns only — nothing of x, theta,
  // y or v is loaded for them (the partials of a linear row).  generate() gives them a body of their own (Options::jac_split).
  std::set<int> data_only_templates() const {
    std::set<int> data, computed;
    for (const Output &o : outs_) {
      bool pure = true;
      std::vector<int> st(o.vals.begin(), o.vals.end());
      std::set<int> seen;
      while (!st.empty() && pure) {
        const int id = st.back(); st.pop_back();
        if (id < 0 || !seen.insert(id).second) continue;
        const VNode &n = v_[id];
        if (n.op == VLD) { if (loads_[n.sub].arr != 3) pure = false; }
        else if (n.op == VW) pure = false;
        else if (n.op == VUN || n.op == VGUARD) st.push_back(n.a);
        else if (n.op == VBIN || n.op == VSEL) { st.push_back(n.a); st.push_back(n.b); }
      }
      (pure ? data : computed).insert(o.tpl);
    }
    for (int t : computed) data.erase(t);
    return data;
  }
  // values coordinate d takes over the launch domain (a folded group derives q1, q2 from q0)
  void coord_range(int d, int64_t &lo, int64_t &hi) const {
    lo = 0; hi = g_.ext[d] - 1;
    if (g_.fold_n > 0 && g_.nd == 1 && d == 1) { lo = -((g_.fold_off + g_.fold_n - 1) / g_.fold_n); hi = std::max<int64_t>(g_.ext[0] - 1 - g_.fold_off, 0) / g_.fold_n; }
    if (g_.fold_n > 0 && d == 2) { lo = 0; hi = g_.fold_n - 1; }
  }
  int qstep() const {
    const bool blk = (kind_ == KK_JAC || kind_ == KK_HESS || kind_ == KK_CONS || kind_ == KK_JPROD) && opt_.store_mode == 2;
    return (blk && opt_.overlap && opt_.block >= 256) ? opt_.block - 16 : opt_.block;
  }
  std::string qstep_str() const { return qstep() == opt_.block ? "IEM_TILE" : "(IEM_TILE - 16)"; }
  void set_zero_fill(const std::vector<std::pair<int64_t, int64_t>> &ranges) { zero_fill_ = ranges; }

  int stage_budget(int max_ns, int total_ns) const {
    int b = std::max(1, opt_.lds_slots * 256 / opt_.block);
    b = std::min(b, std::max(1, total_ns));   // never more than the kernel stages in total
    b = std::max(b, max_ns);
    if ((long long)b * opt_.block * 8 > 160 * 1024) throw std::runtime_error("internal: LDS staging budget exceeds a CU");
    return b;
  }

  int ip_index(int64_t v) {
    ip(v);
    return ip_ids_[std::make_pair(0, v)];
  }

  static constexpr int IEM_BLOCK_WAVES = 4;

  const Model &m_;
  const Group &g_;
  int kind_;
  Options opt_;
  std::string name_;
  std::vector<VNode> v_;
  std::map<VKey, int> memo_;
  std::vector<IdxVal> idx_;
  std::map<IdxVal, int> idx_ids_;
  std::vector<ILoad> iloads_;
  std::vector<Load> loads_;
  std::map<std::tuple<int, int, int>, int> load_ids_;
  std::vector<std::pair<int, int>> sels_;
  std::vector<std::string> guards_;
  std::map<std::string, int> guard_ids_;
  std::vector<Output> outs_;
  std::map<std::pair<int, int64_t>, int> ip_ids_;
  std::vector<int64_t> ipv_;
  std::map<uint64_t, int> dp_ids_;
  std::vector<double> dpv_;
  std::map<int, int> fa_ids_, ia_ids_;
  std::vector<int> fav_, iav_;
  int64_t alg_w_ = 0, alg_r_loads_ = 0, iload_elems_ = 0;
  std::vector<char> lazy_load_;   // per load: 0 in the head, 1 at first use (not yet emitted), 2 emitted
  std::vector<std::pair<int64_t, int64_t>> zero_fill_;  // [lo, hi) ranges of OUT this kernel zeroes itself
  SharedInfo shared_;
  std::map<int, std::vector<std::pair<int64_t, int64_t>>> ranges_;
};

// ---------------------------------------------------------------------------
// grouping + validation
// ---------------------------------------------------------------------------
// `solo(ti)`: template ti gets a launch domain of its own (its item box) instead of joining the
// lanes of its support grid
std::vector<Group> make_groups(const Model &m, const std::function<bool(size_t)> &solo, bool flat2d) {
  std::vector<Group> groups;
  std::map<std::pair<int64_t, int>, int> by_key;
  std::vector<int> scalars;
  for (size_t ti = 0; ti < m.tpl.size(); ++ti) {
    const Template &t = m.tpl[ti];
    if (t.grid_id == 0 || (t.n_items == 1 && t.ifields.empty() && t.ffields.empty())) {
      scalars.push_back((int)ti);
      continue;
    }
    int gi;
    if (t.grid_id > 0 && !solo(ti)) {
      auto key = std::make_pair(t.grid_id, t.nd);
      auto it = by_key.find(key);
      if (it == by_key.end()) {
        gi = (int)groups.size();
        groups.emplace_back();
        groups[gi].grid_id = t.grid_id;
        groups[gi].nd = t.nd;
        for (int d = 0; d < 3; ++d) { groups[gi].lo[d] = d < t.nd ? t.origin[d] : 0; groups[gi].ext[d] = d < t.nd ? t.origin[d] + t.dims[d] : 1; }
        by_key.emplace(key, gi);
      } else {
        gi = it->second;
        for (int d = 0; d < t.nd; ++d) {
          groups[gi].lo[d] = std::min(groups[gi].lo[d], t.origin[d]);
          groups[gi].ext[d] = std::max(groups[gi].ext[d], t.origin[d] + t.dims[d]);  // hi for now
        }
      }
    } else {
      gi = (int)groups.size();
      groups.emplace_back();
      groups[gi].grid_id = -1 - (int64_t)ti;
      groups[gi].nd = t.nd;
      // own launch domain = the template's item box (at its grid origin when it sits on a support grid)
      for (int d = 0; d < 3; ++d) { groups[gi].lo[d] = (t.grid_id > 0 && d < t.nd) ? t.origin[d] : 0; groups[gi].ext[d] = d < t.nd ? t.dims[d] : 1; }
    }
    groups[gi].tpls.push_back((int)ti);
  }
  for (auto &g : groups)
    if (g.grid_id > 0)
      for (int d = 0; d < g.nd; ++d) g.ext[d] -= g.lo[d];  // hi -> extent
  for (auto &g : groups)
    if (g.nd > 1 && g.ext[0] < 64 && g.ext[1] * g.ext[2] > 1) g.flat = true;
  // 2-D grids: one linear lane index over the whole grid, so that no row ends in a partly filled
  // workgroup; sub-box templates are stored by item ordinal (iem_flush_ord)
  if (flat2d)
    for (auto &g : groups)
      if (g.nd == 2 && g.ext[1] > 1) g.flat = true;
  if (!scalars.empty()) {
    if (groups.empty()) {
      groups.emplace_back();
      groups[0].grid_id = 0;
    }
    size_t best = 0;
    int64_t vol = -1;
    for (size_t i = 0; i < groups.size(); ++i) {
      int64_t v = groups[i].ext[0] * groups[i].ext[1] * groups[i].ext[2];
      if (v > vol) { vol = v; best = i; }
    }
    if (vol <= 0) {   // every grid is empty (templates without items): the scalars need a launch of their own
      groups.emplace_back();
      best = groups.size() - 1;
      groups[best].grid_id = 0;
    }
    groups[best].scalars = scalars;
  }
  return groups;
}

// Orthogonal collocation (Group::fold_n): put the node x element boxes of the derivative rows — and the element lists
// of constant_over_collocation — onto the lanes of the 1-D support grid they differentiate over.  `partial`: also boxes
// with fewer nodes per element than the fold and element lists (scatter kinds only: their items are not contiguous in
// the lane index, which the block store of the COO kinds needs).
void fold_groups(const Model &m, std::vector<Group> &groups, int max_n, bool partial) {
  typedef std::array<int64_t, 4> Map;   // variable index (1-based) = c + a*j + b*e + x*xi
  // `lead`: dimensions of the node x element part of the box (2 for a full box, 1 for an element list); a further
  // dimension is the second dimension of the grid the box is folded onto
  auto var_maps = [&](const Template &t, int lead, std::vector<Map> &out) {
    for (const Node &nd : t.nodes) {
      if (nd.op != IEM_OP_VAR) continue;
      const IdxExpr &ix = t.idx[nd.a];
      Map mp{ix.c0, 0, 0, 0};
      for (int j = 0; j < ix.nterms; ++j) {
        const FieldDesc &f = t.ifields[ix.field[j]];
        if (f.mode != IEM_F_AFFINE) return false;
        mp[0] += ix.coef[j] * f.base;
        if (lead >= 2) { mp[1] += ix.coef[j] * f.step[0]; mp[2] += ix.coef[j] * f.step[1]; }
        else mp[2] += ix.coef[j] * f.step[0];
        if (t.nd > lead) mp[3] += ix.coef[j] * f.step[lead];
      }
      out.push_back(mp);
    }
    return true;
  };
  auto is_line = [](const Group &g) { return g.grid_id > 0 && g.nd == 1 && !g.flat; };
  auto is_sheet = [](const Group &g) { return g.grid_id > 0 && g.nd == 2 && !g.flat; };
  // lane-affine variable maps of a grid's own templates: (c at lane 0 [row 0], lane stride, stride of the second dimension)
  auto line_maps = [&](const Group &g, std::vector<std::array<int64_t, 3>> &out) {
    for (int ti : g.tpls) {
      if (g.folded.count(ti)) continue;
      const Template &u = m.tpl[ti];
      std::vector<Map> mm;
      if (u.nd != g.nd || !var_maps(u, 1, mm)) continue;
      const int64_t sh0 = g.lo[0] - u.origin[0], sh1 = g.nd == 2 ? g.lo[1] - u.origin[1] : 0;
      for (const Map &mp : mm) if (mp[2] != 0) out.push_back({mp[0] + mp[2] * sh0 + mp[3] * sh1, mp[2], mp[3]});
    }
  };
  // 1. full boxes (fold_n nodes x ne elements [x the second dimension of a 2-D grid], all templates of the flat group):
  // they decide fold_n and fold_off by vote
  for (size_t fi = 0; fi < groups.size(); ++fi) {
    const Group &F = groups[fi];
    if (F.grid_id <= 0 || !F.flat || (F.nd != 2 && F.nd != 3) || F.ext[0] < 2 || F.ext[0] > max_n || F.ext[1] < 1 || F.fold_n > 0) continue;
    const int64_t n = F.ext[0], ne = F.ext[1];
    bool full = true;
    std::vector<Map> fm;
    for (int ti : F.tpls) {
      const Template &t = m.tpl[ti];
      if (t.nd != F.nd || !var_maps(t, 2, fm)) { full = false; break; }
      for (int d = 0; d < F.nd; ++d) if (t.dims[d] != F.ext[d] || t.origin[d] != F.lo[d]) full = false;
      // (a box on a 2-D grid: every integer field must advance by whole elements — fold_aff has no element coordinate there)
      if (F.nd == 3) for (const FieldDesc &f : t.ifields) if (f.step[1] % n != 0) full = false;
      if (F.nd == 3) for (const FieldDesc &f : t.ffields) if (f.step[1] % n != 0) full = false;
    }
    if (!full) continue;
    std::map<std::pair<size_t, int64_t>, int64_t> votes;   // (target group, offset) -> matching maps
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      const Group &G = groups[gi];
      if (!(F.nd == 2 ? is_line(G) : is_sheet(G)) || (G.fold_n > 0 && G.fold_n != n) || G.ext[0] < n * ne) continue;
      if (F.nd == 3 && (F.lo[2] < G.lo[1] || F.lo[2] + F.ext[2] > G.lo[1] + G.ext[1])) continue;   // the box's last dimension lies on the grid's second
      std::vector<std::array<int64_t, 3>> gm;
      line_maps(G, gm);
      for (const Map &mp : fm) {
        if (mp[1] == 0 || mp[2] != n * mp[1]) continue;   // the row's own node: full stride
        for (auto &cg : gm) {
          if (cg[1] != mp[1] || cg[2] != mp[3]) continue;
          const int64_t c = mp[0] + (F.nd == 3 ? mp[3] * (G.lo[1] - F.lo[2]) : 0);     // the box's map at the grid's row 0
          if ((c - cg[0]) % mp[1] != 0) continue;
          const int64_t off = (c - cg[0]) / mp[1];
          if (off < 0 || off + n * ne > G.ext[0] || (G.fold_n > 0 && off != G.fold_off)) continue;
          ++votes[{gi, off}];
        }
      }
    }
    if (votes.empty()) continue;
    auto best = votes.begin();
    for (auto it = votes.begin(); it != votes.end(); ++it) if (it->second > best->second) best = it;
    Group &G = groups[best->first.first];
    G.fold_n = n; G.fold_off = best->first.second;
    for (int ti : F.tpls) { G.tpls.push_back(ti); G.folded.insert(ti); }
    G.scalars.insert(G.scalars.end(), F.scalars.begin(), F.scalars.end());
    groups.erase(groups.begin() + (long)fi);
    --fi;
  }
  if (!partial) return;
  // 2. element lists and narrower boxes whose variable maps advance by fold_n entries of a slab of a (1-D) line per element
  for (size_t fi = 0; fi < groups.size(); ++fi) {
    const Group &F = groups[fi];
    if (F.grid_id <= 0 || F.fold_n > 0 || F.nd > 2 || (F.nd == 2 && !F.flat) || F.ext[2] != 1) continue;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      Group &G = groups[gi];
      if (gi == fi || G.fold_n == 0 || G.nd != 1) continue;
      const int64_t n = G.fold_n, w = F.nd == 2 ? F.ext[0] : 1, ne = F.nd == 2 ? F.ext[1] : F.ext[0];
      if (w >= n || ne < 1 || n * ne > G.ext[0] + n) continue;
      std::vector<std::array<int64_t, 3>> gm;
      line_maps(G, gm);
      bool fits = true;
      for (int ti : F.tpls) {
        const Template &t = m.tpl[ti];
        std::vector<Map> fm;
        int64_t tw = t.nd == 2 ? t.dims[0] : 1, tne = t.nd == 2 ? t.dims[1] : t.dims[0];
        if (t.nd != F.nd || tw != w || tne != ne || !var_maps(t, t.nd, fm) || fm.empty()) { fits = false; break; }
        for (const Map &mp : fm) {
          bool hit = false;
          if (mp[2] != 0 && mp[2] % n == 0)
            for (auto &cg : gm) {
              if (cg[1] != mp[2] / n || (mp[0] - cg[0]) % cg[1] != 0 || mp[1] % cg[1] != 0) continue;
              // owner lanes of the corner items lie on the line
              const int64_t l0 = (mp[0] - cg[0]) / cg[1], l1 = l0 + (mp[1] * (w - 1)) / cg[1] + n * (ne - 1);
              if (std::min(l0, l1) >= 0 && std::max(l0, l1) < G.ext[0]) hit = true;
            }
          if (!hit) { fits = false; break; }
        }
        if (!fits) break;
      }
      if (!fits) continue;
      for (int ti : F.tpls) { G.tpls.push_back(ti); G.folded.insert(ti); }
      G.scalars.insert(G.scalars.end(), F.scalars.begin(), F.scalars.end());
      groups.erase(groups.begin() + (long)fi);
      --fi;
      break;
    }
  }
}

// min / max of an index expression over the template's item box — host-side safety check
// (128-bit arithmetic: blobs come from foreign producers and may hold anything)
void validate_indices(const Model &m) {
  typedef __int128 wide;
  for (size_t ti = 0; ti < m.tpl.size(); ++ti) {
    const Template &t = m.tpl[ti];
    if (t.n_items == 0) continue;
    std::vector<std::pair<wide, wide>> frange(t.ifields.size());
    for (size_t f = 0; f < t.ifields.size(); ++f) {
      const FieldDesc &fd = t.ifields[f];
      if (fd.mode == IEM_F_AFFINE) {
        wide lo = fd.base, hi = fd.base;
        for (int d = 0; d < 3; ++d) {
          wide e = (wide)fd.step[d] * (t.dims[d] - 1);
          if (e < 0) lo += e; else hi += e;
        }
        frange[f] = {lo, hi};
      } else {
        const ArrayDesc &a = m.arrs[fd.arr];
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        if (a.kind == IEM_A_I64_RANGE) {
          if (a.n > 0) {
            wide last = (wide)a.r0 + (wide)a.rstep * (a.n - 1);
            if (last > INT64_MAX || last < INT64_MIN) throw std::runtime_error("template " + std::to_string(ti) + ": range array overflows");
            lo = std::min(a.r0, (int64_t)last); hi = std::max(a.r0, (int64_t)last);
          }
        } else {
          for (int64_t j = 0; j < a.n; ++j) { int64_t v = a.i(j); lo = std::min(lo, v); hi = std::max(hi, v); }
        }
        frange[f] = {lo, hi};
      }
    }
    std::vector<std::pair<wide, wide>> irange(t.idx.size());
    for (size_t i = 0; i < t.idx.size(); ++i) {
      const IdxExpr &ix = t.idx[i];
      wide lo = ix.c0, hi = ix.c0;
      // affine terms over the same box are correlated; evaluate the box corners exactly when all affine
      bool all_aff = true;
      for (int j = 0; j < ix.nterms; ++j) if (t.ifields[ix.field[j]].mode != IEM_F_AFFINE) all_aff = false;
      if (all_aff) {
        wide c = ix.c0, k[3] = {0, 0, 0};
        for (int j = 0; j < ix.nterms; ++j) {
          const FieldDesc &fd = t.ifields[ix.field[j]];
          c += (wide)ix.coef[j] * fd.base;
          for (int d = 0; d < 3; ++d) k[d] += (wide)ix.coef[j] * fd.step[d];
        }
        lo = hi = c;
        for (int d = 0; d < 3; ++d) {
          if (k[d] > ((wide)1 << 62) || k[d] < -((wide)1 << 62)) throw std::runtime_error("template " + std::to_string(ti) + ": index stride overflows");
          wide e = k[d] * (t.dims[d] - 1);
          if (e < 0) lo += e; else hi += e;
        }
      } else {
        for (int j = 0; j < ix.nterms; ++j) {
          wide a = (wide)ix.coef[j] * frange[ix.field[j]].first, b = (wide)ix.coef[j] * frange[ix.field[j]].second;
          lo += std::min(a, b); hi += std::max(a, b);
        }
      }
      irange[i] = {lo, hi};
    }
    for (const Node &nd : t.nodes) {
      if (nd.op != IEM_OP_VAR && nd.op != IEM_OP_PAR) continue;
      int64_t lim = nd.op == IEM_OP_VAR ? m.nvar : m.npar;
      if (irange[nd.a].first < 1 || irange[nd.a].second > lim)
        throw std::runtime_error("template " + std::to_string(ti) + ": " + (nd.op == IEM_OP_VAR ? "variable" : "parameter") +
                                 " index range [" + std::to_string((double)irange[nd.a].first) + ", " + std::to_string((double)irange[nd.a].second) +
                                 "] outside 1:" + std::to_string(lim));
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// Workgroup size when the caller leaves it open (block = 0).  512 lanes halve the partial cache lines
// at block seams — but on a grid of two or more dimensions every ROW ends in a partly
// filled workgroup: 5 000 time supports are 10.08 tiles of 496 (the 11th workgroup of each row runs
// 8 % full: +9 % lanes) but 20.8 tiles of 240 (+0.8 %).  Pick the size that wastes fewer lanes, 512
// unless 256 saves more than 2 % (measured on pandemic 5000 x 100: cons! 13.9 -> 13.2 us,
// jac_coord! 21.6 -> 20.6 us, profiles/r02_ab_pandemic_tiles.txt).
static int choose_block(const Model &m, const Options &opt) {
  const bool overlap = opt.store_mode == 2 && opt.overlap;
  double lanes[2] = {0.0, 0.0}, items = 0.0;
  const int cand[2] = {512, 256};
  for (const Template &t : m.tpl) {
    if (t.nd < 2 || t.grid_id <= 0 || t.dims[0] < 64 || t.n_items == 0) continue;
    if (t.nd == 2 && opt.flat2d) continue;   // flat 2-D grids have no per-row waste
    items += (double)t.n_items;
    for (int c = 0; c < 2; ++c) {
      const int64_t step = overlap ? cand[c] - 16 : cand[c];
      lanes[c] += (double)((t.dims[0] + step - 1) / step * step) * (double)(t.n_items / t.dims[0]);
    }
  }
  if (items == 0.0) return 512;
  return lanes[1] < lanes[0] - 0.02 * items ? 256 : 512;
}

// workgroups of a kernel over support grid g that advances `qs` grid points per workgroup along dim 0
static void launch_grid(const Group &g, int64_t qs, KernelDesc &kd) {
  kd.grid[0] = (g.ext[0] + qs - 1) / qs; kd.grid[1] = g.ext[1]; kd.grid[2] = g.ext[2];
  if (g.flat) { kd.grid[0] = (g.ext[0] * g.ext[1] * g.ext[2] + qs - 1) / qs; kd.grid[1] = kd.grid[2] = 1; }
  if (!g.flat && g.nd == 2 && g.ext[1] > 65535) { kd.grid[1] = 65535; kd.grid[2] = (g.ext[1] + 65534) / 65535; }
  kd.n_blocks = kd.grid[0] * kd.grid[1] * kd.grid[2];
}

// The options the kernels of one KIND are generated with: the handle's, except that jac_coord! / hess_coord! get the
// LARGE-GRID shape (Options::big_*) when any support grid of the model gives that kind at least big_batch_jac / _hess
// workgroups at the model's own tile.  A function of the kind and the grid sizes alone — never of a timer or of the process
// — and one decision per kind: the kernels (bodies) of a kind share launches, hence a workgroup size.
static Options kind_options(const Options &opt, const std::vector<Group> &groups, int kind) {
  Options ko = opt;
  // cons! on a 2-D support grid: one value per row and template, lane-consecutive — the plain stores are coalesced as they are, and on
  // the short rows of such a grid (256-lane tiles, 16 of them recomputed per seam) the LDS re-cut costs more than the whole lines it
  // buys: pandemic 5 000 x 100: 15.7 -> 12.3 us (on a 1-D grid of 512-lane tiles the re-cut wins, quadrotor: 55.7 against 60.2;
  // profiles/r04_kind_ab_cons_store_modes.txt).  Decided by the LARGEST grid of the kind — its shape, not its size.
  if ((kind == KK_CONS || kind == KK_JPROD) && opt.store_mode == 2 && opt.cons_direct_2d && !opt.no_fuse) {      // (J v has cons!'s output shape)
    const Group *big = nullptr;
    int64_t pts = 0;
    for (const Group &g : groups) {
      if (g.grid_id <= 0) continue;
      const int64_t n = g.ext[0] * g.ext[1] * g.ext[2];
      if (n > pts) { pts = n; big = &g; }
    }
    if (big && big->nd >= 2 && !big->flat) ko.store_mode = 1;
    return ko;
  }
  if ((kind != KK_JAC && kind != KK_HESS) || opt.store_mode != 2 || opt.no_fuse || opt.big_batch_slots <= opt.lds_slots) return ko;
  const int64_t thr = kind == KK_JAC ? opt.big_batch_jac : opt.big_batch_hess;
  if (thr <= 0) return ko;
  int64_t biggest = 0;
  for (const Group &g : groups) {
    if (g.grid_id <= 0) continue;
    KernelDesc probe;
    launch_grid(g, (opt.overlap && opt.block >= 256) ? opt.block - 16 : opt.block, probe);
    biggest = std::max(biggest, probe.n_blocks);
  }
  if (biggest >= thr) {
    ko.lds_slots = opt.big_batch_slots;
    if (opt.big_xcd) ko.xcd_remap = 1;
    if (opt.big_tile > 0) ko.block = opt.big_tile;
  }
  return ko;
}

// algorithmic bytes READ by one launch of several bodies: every distinct input element once (the union of the bodies' ranges)
template <class Builders>
static int64_t union_read_bytes(const Builders &bs) {
  std::map<int, std::vector<std::pair<int64_t, int64_t>>> all;
  int64_t elems = 0;
  for (const auto *b : bs) {
    for (auto &kv : b->read_ranges()) { auto &d = all[kv.first]; d.insert(d.end(), kv.second.begin(), kv.second.end()); }
    elems += b->iload_elems();
  }
  for (auto &kv : all) {
    auto &v = kv.second;
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    int64_t lo = v[0].first, hi = v[0].second;
    for (size_t i = 1; i < v.size(); ++i) {
      if (v[i].first <= hi + 1) hi = std::max(hi, v[i].second);
      else { elems += hi - lo + 1; lo = v[i].first; hi = v[i].second; }
    }
    elems += hi - lo + 1;
  }
  return 8 * elems;
}

// Workgroup-id dispatch over a few bodies (<= 4) of one launch: body j owns the workgroups [first_j, first_j + n_j) —
// except a LEADING RUN of `run` bodies with one common grid, whose workgroups are interleaved so that all of them are
// resident together (Options::jac_split / pair_inter).  mode 1: workgroup r of the run -> body r % run, tile r / run;
// mode 2: in runs of 8 consecutive workgroups (hardware deals workgroups round-robin over the 8 XCDs: every XCD then
// works on every body), the last n % 8 tiles of each body behind them.  `dec`: decode table {first, gx, gy, gz} per body.
static void emit_dispatch_chain(std::ostream &src, size_t nb, size_t dec, size_t run, int mode, const std::vector<bool> &remap, const std::string &b,
                                const std::function<std::string(size_t, const std::string &)> &call,
                                const std::function<std::string(size_t)> &extra, bool shift_first = false) {
  size_t j0 = 0;
  if (run >= 2) {
    const size_t e = dec;
    src << "  ";
    if (run < nb) src << "if (" << b << " < A.ip[" << (dec + 4 * run) << "]) ";
    src << "{\n    const long long gx = A.ip[" << (e + 1) << "], gy = A.ip[" << (e + 2) << "], gz = A.ip[" << (e + 3) << "];\n"
        << "    const long long n_ = gx * gy * gz, r_ = " << b << " - A.ip[" << e << "];\n    long long j_, lq_;\n";
    if (mode == 1)
      src << "    j_ = r_ % " << run << "; lq_ = r_ / " << run << ";\n";
    else
      src << "    { const long long f_ = (n_ >> 3) * " << (8 * run) << ";\n"
          << "      if (r_ < f_) { j_ = (r_ >> 3) % " << run << "; lq_ = (r_ / " << (8 * run) << ") * 8 + (r_ & 7); }\n"
          << "      else { const long long t_ = n_ & 7, q_ = r_ - f_; j_ = q_ / t_; lq_ = (n_ & ~7LL) + q_ % t_; } }\n";
    if (shift_first) src << "    if (j_ == 0) { lq_ += n_ >> 1; if (lq_ >= n_) lq_ -= n_; }\n";
    src << "    const long long lb = " << (remap[0] ? "iem_xcd_remap(lq_, n_)" : "lq_") << ";\n";
    for (size_t j = 0; j < run; ++j) {
      src << "    " << (j ? "else " : "");
      if (j + 1 < run) src << "if (j_ == " << j << ") ";
      src << "{\n" << extra(j) << call(j, "      ") << "    }\n";
    }
    src << "  }\n";
    j0 = run;
  }
  for (size_t j = j0; j < nb; ++j) {
    const size_t e = dec + 4 * j;
    src << "  " << (j ? "else " : "");
    if (j + 1 < nb) src << "if (" << b << " < A.ip[" << (e + 4) << "]) ";
    src << "{\n    const long long gx = A.ip[" << (e + 1) << "], gy = A.ip[" << (e + 2) << "], gz = A.ip[" << (e + 3) << "];\n"
        << "    const long long lb = " << (remap[j] ? "iem_xcd_remap(" + b + " - A.ip[" + std::to_string(e) + "], gx * gy * gz)" : b + " - A.ip[" + std::to_string(e) + "]") << ";\n"
        << extra(j) << call(j, "    ") << "  }\n";
  }
}

Program generate(const Model &m, const Options &opt_in) {
  validate_indices(m);
  Options opt = opt_in;
  if (opt.block == 0) opt.block = choose_block(m, opt);
  Program P;
  P.block = opt.block;
  std::vector<Group> groups = make_groups(m, [&](size_t) { return opt.no_fuse != 0; }, opt.flat2d != 0);
  // The scatter kinds (grad, J'v, Hv) always keep the lane-fused groups: when several templates add
  // into the same entry, ONE lane issues those adds in program order (deterministic); side by side
  // they would come from different workgroups in arrival order.
  std::vector<Group> groups_fused = groups;
  bool split = false;
  if (!opt.no_fuse && opt.split_small > 0 && !opt.hess_merge) {   // the merged Hessian layout is defined on fused lanes
    // A support grid that fills only a fraction of the chip (<= split_small workgroups) is
    // latency-bound: one wave runs the whole fused lane program while most CUs idle.  Its
    // templates then run side by side — one body each, all in the same launch (the workgroup-id
    // dispatch below) — trading the cross-template CSE for parallelism.  Measured crossover on
    // MI355X: 64 workgroups (32 768 supports at 512 lanes), profiles/r01_ab_small_grids.txt.
    std::set<size_t> solo;
    for (const Group &g : groups) {
      if (g.grid_id <= 0 || g.tpls.size() < 2) continue;
      const int64_t nb = g.flat ? (g.ext[0] * g.ext[1] * g.ext[2] + opt.block - 1) / opt.block
                                : (g.ext[0] + opt.block - 1) / opt.block * g.ext[1] * g.ext[2];
      if (nb <= opt.split_small) solo.insert(g.tpls.begin(), g.tpls.end());
    }
    if (!solo.empty()) { groups = make_groups(m, [&](size_t ti) { return solo.count(ti) != 0; }, opt.flat2d != 0); split = true; }
  }
  // Orthogonal collocation: the scatter kinds put the node x element boxes (and the element lists) on the lanes of the
  // support grid itself, so that every entry has ONE writer (pull_folded); fold_colloc = 2: the full boxes for every kind.
  if (opt.fold_colloc > 0 && !opt.no_fuse) {
    const size_t before = groups_fused.size();
    fold_groups(m, groups_fused, opt.fold_max_n, true);
    if (groups_fused.size() != before) split = true;   // (the scatter kinds then run on groups of their own)
    if (opt.fold_colloc >= 2) fold_groups(m, groups, opt.fold_max_n, false);
  }
  std::ostringstream src;
  src << "// generated by libiem_hip (iem_codegen.cpp) — do not edit\n";
  static const char *kname[] = {"cons", "jac", "hess", "obj", "grad", "jprod", "jtprod", "hprod"};

  // gradient slot classification needs a global view of every objective slot's index range
  struct GSlot { int kind; int kernel; int out; int slot; int64_t lo, hi; bool injective, uniform0; AffQ aff; bool pure; int64_t count = 0;
                 int64_t box_lo[3] = {0, 0, 0}, box_n[3] = {1, 1, 1}; bool axis_ok = false; bool empty = false; };
  std::vector<GSlot> gslots;
  std::vector<std::unique_ptr<KernelBuilder>> builders;
  std::vector<KernelDesc> descs;
  std::vector<Options> kopts;   // the options each builder was made with (kind_options)
  std::deque<Group> sub_groups;  // Options::jac_split: the two halves of a support grid's templates (builders keep references)
  std::map<size_t, std::pair<const Group *, std::string>> whole_of;   // desc index of a split body's FIRST half -> (the whole grid, its kernel name)
  std::set<size_t> second_half;

  auto is_scatter = [](int kind) { return kind == KK_GRAD || kind == KK_JTPROD || kind == KK_HPROD; };
  // a handle's second code object (the tuner's large store batch) carries a tag in its kernel names, so that a
  // profile of a run that used both tells them apart
  const std::string name_tag = opt.name_tag > 0 ? "_b" + std::to_string(opt.name_tag) : std::string();
  for (int pass = 0; pass < (split ? 2 : 1); ++pass)
  for (size_t gi = 0; gi < (pass ? groups_fused.size() : groups.size()); ++gi) {
    const Group &g = pass ? groups_fused[gi] : groups[gi];
    if (!g.flat && ((g.nd > 2 && g.ext[1] > 65535) || g.ext[2] > 65535 || g.ext[1] > 65535LL * 65535LL))
      throw std::runtime_error("support grid too large in dims 2/3 (limit 65535 per dimension for 3-D grids)");
    for (int kind = 0; kind < KK_COUNT; ++kind) {
      if (split && is_scatter(kind) != (pass == 1)) continue;   // pass 1: the scatter kinds on the fused groups
      std::string name = std::string("iem_") + kname[kind] + "_g" + std::to_string(gi) + name_tag;
      const Options ko = kind_options(opt, pass ? groups_fused : groups, kind);
      auto kb = std::make_unique<KernelBuilder>(m, g, kind, ko, name);
      if (!kb->build(nullptr)) continue;
      if (kind == KK_JAC && opt.jac_split > 0 && !opt.no_fuse && ko.store_mode == 2 && g.grid_id > 0) {
        // a large grid: the templates whose partials are item data get a body of their own (Options::jac_split)
        KernelDesc probe;
        launch_grid(g, kb->qstep(), probe);
        const std::set<int> data = kb->data_only_templates();
        int n_data = 0, n_comp = 0;   // lane templates on either side (scalars follow their class, but never make a body)
        for (int ti : g.tpls) if (kb->relevant(m.tpl[ti])) (data.count(ti) ? n_data : n_comp)++;
        if (probe.n_blocks > opt.jac_split_min && n_data > 0 && n_comp > 0) {
          for (int half = 0; half < 2; ++half) {
            sub_groups.push_back(g);
            Group &sg = sub_groups.back();
            sg.tpls.clear(); sg.scalars.clear();
            for (int ti : g.tpls) if ((data.count(ti) != 0) == (half == 0)) sg.tpls.push_back(ti);
            for (int ti : g.scalars) if ((data.count(ti) != 0 || !kb->relevant(m.tpl[ti])) == (half == 0)) sg.scalars.push_back(ti);
            const std::string hname = name + (half ? "b" : "a");
            auto hb = std::make_unique<KernelBuilder>(m, sg, kind, ko, hname);
            if (!hb->build(nullptr)) throw std::runtime_error("internal: empty half of a split jac_coord! body");
            KernelDesc hd;
            hd.name = hname; hd.kind = kind; hd.block = ko.block; hd.lds_slots = ko.lds_slots; hd.inter = (int)gi;
            launch_grid(sg, hb->qstep(), hd);
            if (half == 0) whole_of[descs.size()] = {&g, name}; else second_half.insert(descs.size());
            builders.push_back(std::move(hb));
            descs.push_back(hd);
            kopts.push_back(ko);
          }
          continue;
        }
      }
      if (kind == KK_HESS && opt.hess_merge) kb->merge_hess(P.nnzh_merged, P.hess_classes);
      KernelDesc kd;
      kd.name = name;
      kd.kind = kind;
      kd.block = ko.block;
      kd.lds_slots = ko.lds_slots;
      launch_grid(g, kb->qstep(), kd);
      if (kind == KK_GRAD || kind == KK_JTPROD || kind == KK_HPROD) {
        auto &outs = kb->outputs();
        for (size_t oi = 0; oi < outs.size(); ++oi) {
          const bool scalar = outs[oi].scalar;
          int64_t box_n[3], box_items = 1;   // the output's own item box in the launch domain (a pulled clone's is shifted, a union's is wider than its template's)
          for (int d = 0; d < 3; ++d) { box_n[d] = scalar ? 1 : std::max<int64_t>(outs[oi].qhi[d] - outs[oi].qlo[d], 1); box_items *= box_n[d]; }
          for (size_t s = 0; s < outs[oi].grad_idx.size(); ++s) {
            if (outs[oi].grad_mode[s] < 0) continue;   // folded into another slot (merge_scatter)
            const IdxVal &iv = kb->idxvals()[outs[oi].grad_idx[s]];
            GSlot gs{kind, (int)builders.size(), (int)oi, (int)s, 0, 0, false, false, iv.aff, iv.ind.empty()};
            // a folded group derives q1, q2 from the lane: a destination that still depends on them is not an interval of the lane
            if (g.fold_n > 0 && ((g.nd == 1 && iv.aff.k[1] != 0) || iv.aff.k[2] != 0)) gs.pure = false;
            for (int d = 0; d < 3; ++d) { gs.box_lo[d] = scalar ? 0 : outs[oi].qlo[d]; gs.box_n[d] = box_n[d]; if (!scalar && outs[oi].qhi[d] <= outs[oi].qlo[d]) gs.empty = true; }
            gs.count = gs.empty ? 0 : box_items;
            if (gs.pure) {
              int64_t lo = iv.aff.c, hi = iv.aff.c;
              // q range of this template: q_d in [origin-lo, origin-lo+dims)
              bool inj = true;
              int64_t reach = 0;  // largest offset reachable through the lower dims
              for (int d = 0; d < 3; ++d) {
                int64_t n = box_n[d];
                int64_t q_lo = scalar ? 0 : outs[oi].qlo[d];
                int64_t a = iv.aff.k[d] * q_lo, b = iv.aff.k[d] * (q_lo + n - 1);
                lo += std::min(a, b); hi += std::max(a, b);
                if (n > 1) {
                  if (iv.aff.k[d] <= reach) inj = false;
                  reach += iv.aff.k[d] * (n - 1);
                }
              }
              gs.lo = lo; gs.hi = hi; gs.injective = inj;
              gs.count = box_items;
              if (outs[oi].fold == 2) gs.count = (box_n[0] - 1) / g.fold_n + 1;   // a pinned clone: one lane in fold_n
              for (int d = 0; d < 3; ++d) { gs.box_lo[d] = scalar ? 0 : outs[oi].qlo[d]; gs.box_n[d] = box_n[d]; }
              // a sum over the non-lane axes: the entry depends on the lane only, and every row of dims 1, 2 adds into it
              gs.axis_ok = !g.flat && !scalar && iv.aff.k[0] != 0 && iv.aff.k[1] == 0 && iv.aff.k[2] == 0 && box_n[1] * box_n[2] > 1 &&
                           box_n[1] * box_n[2] <= (1LL << 20);
              // wave-uniform destination: every lane of a wave shares q1/q2 — not true for flat groups
              gs.uniform0 = !g.flat && !scalar && iv.aff.k[0] == 0 && box_n[0] > 1;
            } else {
              gs.lo = INT64_MIN; gs.hi = INT64_MAX;
              if (outs[oi].fold == 2) gs.count = (box_n[0] - 1) / g.fold_n + 1;
            }
            gslots.push_back(gs);
          }
        }
      }
      builders.push_back(std::move(kb));
      descs.push_back(kd);
      kopts.push_back(ko);
    }
  }
  bool accumulates[KK_COUNT] = {};
  std::vector<int> atomic_slots[KK_COUNT];   // gslots that ended as atomics (modes 1, 2)
  std::vector<int> parked_slots[KK_COUNT];   // gslots whose addends go through the plan-driven gather (mode 5)
  // Entries MANY items share (one constant destination: finite / first-stage variables): when every
  // slot of the kind that can reach the entry is such a constant slot, the entry is reduced
  // deterministically (grad_mode 3) and written once by the last workgroup of the call.
  std::vector<int> shared_dest(gslots.size(), 0);   // 1: slot i is a mode-3 slot
  std::map<int64_t, std::vector<int>> dest_slots[KK_COUNT];
  if (opt.det_shared)
    for (int kind : {(int)KK_GRAD, (int)KK_JTPROD, (int)KK_HPROD}) {
      std::map<int64_t, std::vector<int>> cand;
      for (size_t i = 0; i < gslots.size(); ++i)
        if (gslots[i].kind == kind && gslots[i].pure && gslots[i].lo == gslots[i].hi) cand[gslots[i].lo].push_back((int)i);
      for (auto &kv : cand) {
        bool many = false, clean = true;
        for (int i : kv.second) if (gslots[i].count > 1) many = true;
        for (size_t j = 0; j < gslots.size() && clean; ++j) {
          const GSlot &b = gslots[j];
          if (b.kind != kind || (b.pure && b.lo == b.hi && b.lo == kv.first)) continue;
          if (!(b.hi < kv.first || b.lo > kv.first)) clean = false;
        }
        if (!many || !clean) continue;
        for (int i : kv.second) shared_dest[i] = 1;
        dest_slots[kind][kv.first] = kv.second;
      }
    }
  // A FEW items against many (pandemic: the initial conditions on the xi grid hit s(0, xi), which the path rows on the
  // t x xi grid write too — another workgroup of the same launch): instead of turning the big slot's coalesced stores
  // into atomics, the small slots are DEFERRED — parked like the plan-driven gather's addends and ADDED to the entries
  // after the kind's kernels, in plan order.  A slot is small when it has at most 1/32 of the items of the kind's
  // largest slot (and at most 65536); a big injective slot that only small slots reach keeps its exclusive stores.
  std::vector<char> deferred(gslots.size(), 0);
  if (opt.det_scatter >= 1) {
    int64_t maxcount[KK_COUNT] = {};
    for (const GSlot &b : gslots) maxcount[b.kind] = std::max(maxcount[b.kind], b.count);
    auto is_small = [&](const GSlot &b) { return b.pure && b.count > 0 && b.count <= 65536 && b.count * 32 <= maxcount[b.kind]; };
    for (size_t i = 0; i < gslots.size(); ++i) {
      const GSlot &a = gslots[i];
      // (a sum over a non-lane axis is written once per entry by iem_axis_sum_kernel: the few items that also reach its entries
      //  — pandemic with collocation: the rows that hold u constant over an element — are added behind it, like behind a store)
      if (shared_dest[i] || !a.pure || !(a.injective || (a.axis_ok && opt.det_axis)) || a.count == 0 || is_small(a)) continue;
      bool big_clash = false;
      std::vector<size_t> smalls;
      for (size_t j = 0; j < gslots.size() && !big_clash; ++j) {
        const GSlot &b = gslots[j];
        if (i == j || b.kind != a.kind || b.hi < a.lo || b.lo > a.hi) continue;
        if (shared_dest[j] || !is_small(b)) big_clash = true; else smalls.push_back(j);
      }
      if (!big_clash) for (size_t j : smalls) deferred[j] = 1;
    }
  }
  std::vector<int> deferred_slots[KK_COUNT];
  // classify gradient slots: exclusive iff injective and its range meets no other (non-deferred) slot's range
  for (size_t i = 0; i < gslots.size(); ++i) {
    GSlot &a = gslots[i];
    if (shared_dest[i]) { builders[a.kernel]->outputs()[a.out].grad_mode[a.slot] = 3; continue; }
    if (deferred[i]) { deferred_slots[a.kind].push_back((int)i); continue; }   // mode 5 below, with the accumulate flag
    int mode = 2;
    bool clash = false;
    if (a.pure)
      for (size_t j = 0; j < gslots.size() && !clash; ++j) {
        if (i == j || deferred[j]) continue;
        const GSlot &b = gslots[j];
        if (b.kind != a.kind) continue;   // different output vectors
        if (!(b.hi < a.lo || b.lo > a.hi)) clash = true;
      }
    if (a.pure && !a.injective && a.axis_ok && !clash && opt.det_axis) {
      // nobody else reaches these entries: park the rows, iem_axis_sum_kernel writes each entry once (offsets assigned below)
      mode = 4;
      const int64_t k0 = a.aff.k[0];
      P.axis[a.kind].push_back(Program::AxisSum{a.aff.c + k0 * a.box_lo[0], k0, a.box_n[0], a.box_n[1] * a.box_n[2], -(int64_t)i - 1});
      if ((k0 == 1 || k0 == -1) && a.hi - a.lo + 1 == a.box_n[0]) P.covered[a.kind].emplace_back(a.lo, a.hi);
      builders[a.kernel]->outputs()[a.out].grad_mode[a.slot] = 4;
      continue;
    }
    if (a.pure && a.injective) {
      if (!clash) mode = 0;
      // a slot whose items tile its index range without gaps fully overwrites that range:
      // iem_grad need not zero it first
      if (mode == 0 && a.hi - a.lo + 1 == a.count) {
        P.covered[a.kind].emplace_back(a.lo, a.hi);
        if (a.kind == KK_GRAD) P.grad_covered.emplace_back(a.lo, a.hi);
      }
    }
    if (mode == 2 && a.pure && a.uniform0) mode = 1;
    builders[a.kernel]->outputs()[a.out].grad_mode[a.slot] = mode;
    if (mode == 1 || mode == 2) atomic_slots[a.kind].push_back((int)i);
    else if (!(mode == 0 && a.hi - a.lo + 1 == a.count)) accumulates[a.kind] = true;
  }
  // What is still an atomic (entries written from two support grids of a call, collocation stencils, gathered
  // indices): with `det_scatter` every such addend is PARKED — aux[off + lane of the launch domain], a coalesced
  // exclusive store — and a follow-up kernel sums each entry's addends in a fixed order from a plan built here, on
  // the host, from the index expressions (entry -> parked positions).  No float atomics remain, every kind is
  // bitwise reproducible on every model; the price is 8 bytes of plan and of scratch per addend.
  for (int kind : {(int)KK_GRAD, (int)KK_JTPROD, (int)KK_HPROD}) {
    auto &as = atomic_slots[kind];
    auto &ds = deferred_slots[kind];
    if (as.empty() && ds.empty()) continue;
    int64_t total = 0;
    for (int i : as) total += gslots[i].count;
    bool park_atomics = !as.empty() && opt.det_scatter && total <= opt.det_scatter_max;
    // destinations of one slot's items, in item order; `park` = the slot's first position in the gather region
    auto collect = [&](int i, int64_t park, std::vector<int64_t> &dest_of, std::vector<int64_t> *pos_of) {
      const GSlot &a = gslots[i];
      KernelBuilder &kb = *builders[a.kernel];
      const Output &o = kb.outputs()[a.out];
      const Group &g = kb.group();
      const IdxVal &iv = kb.idxvals()[o.grad_idx[a.slot]];
      int64_t q[3];
      if (!a.empty)
      for (q[2] = a.box_lo[2]; q[2] < a.box_lo[2] + a.box_n[2]; ++q[2])
        for (q[1] = a.box_lo[1]; q[1] < a.box_lo[1] + a.box_n[1]; ++q[1])
          for (q[0] = a.box_lo[0]; q[0] < a.box_lo[0] + a.box_n[0]; ++q[0]) {
            int64_t c1 = q[1], c2 = q[2];
            if (g.fold_n > 0) {   // derived coordinates; a pinned clone's items sit on the lanes with q2 == K
              const int64_t l = q[0] - g.fold_off;
              const int64_t el = (l >= 0 ? l : l - (g.fold_n - 1)) / g.fold_n;
              c2 = l - g.fold_n * el;
              if (g.nd == 1) c1 = el;      // (a 2-D grid keeps its own q1)
              if (o.fold == 2 && c2 != o.pin_K) continue;
            }
            int64_t d = iv.aff.c + iv.aff.k[0] * q[0] + iv.aff.k[1] * c1 + iv.aff.k[2] * c2;
            for (auto &t : iv.ind) {
              const ILoad &il = kb.iloads()[t.second];
              const int64_t p = il.pos.c + il.pos.k[0] * q[0] + il.pos.k[1] * c1 + il.pos.k[2] * c2;
              const ArrayDesc &arr = m.arrs[kb.ia_arrays()[il.ia_slot]];
              if (p < 0 || p >= arr.n) throw std::runtime_error("index array position out of range");
              d += t.first * arr.i(p);
            }
            if (d < 0 || d >= m.nvar) throw std::runtime_error("scatter destination out of range");
            dest_of.push_back(d);
            if (pos_of) pos_of->push_back(park + (o.scalar ? 0 : q[0] + g.ext[0] * (q[1] + g.ext[1] * q[2])));
          }
      return o.scalar ? (int64_t)1 : g.ext[0] * g.ext[1] * g.ext[2];
    };
    if (park_atomics && opt.det_scatter < 2) {
      // at most TWO addends per entry: a + b = b + a, those atomics are already order-independent — and cheaper than
      // a second launch
      std::vector<int64_t> d0;
      d0.reserve((size_t)total);
      for (int i : as) collect(i, 0, d0, nullptr);
      std::sort(d0.begin(), d0.end());
      int64_t run = 0, best = 0;
      for (size_t e = 0; e < d0.size(); ++e) { run = (e && d0[e] == d0[e - 1]) ? run + 1 : 1; best = std::max(best, run); }
      if (best <= 2) park_atomics = false;
    }
    if (!as.empty() && !park_atomics) accumulates[kind] = true;
    std::vector<std::pair<int, bool>> chosen;   // (gslot, accumulate onto what the kernels stored)
    for (int i : ds) chosen.emplace_back(i, true);
    if (park_atomics) for (int i : as) chosen.emplace_back(i, false);
    if (chosen.empty()) continue;
    Program::Gather &G = P.gather[kind];
    std::vector<int64_t> dest_of, pos_of;
    std::vector<char> acc_of;
    int64_t park = 0;   // relative to the start of the gather region of the aux buffer
    for (auto &ch : chosen) {
      const GSlot &a = gslots[ch.first];
      Output &o = builders[a.kernel]->outputs()[a.out];
      o.grad_mode[a.slot] = 5;
      o.axis_off[a.slot] = park;   // made absolute below
      const size_t before = dest_of.size();
      park += collect(ch.first, park, dest_of, &pos_of);
      acc_of.insert(acc_of.end(), dest_of.size() - before, ch.second ? 1 : 0);
      parked_slots[kind].push_back(ch.first);
    }
    // sort by entry; within an entry the addends keep slot order, then item order: a fixed order (a stable sort; a
    // model — or a crafted blob — with few addends in a huge variable space must not allocate by nvar)
    const size_t na = dest_of.size();
    std::vector<size_t> order(na);
    for (size_t e = 0; e < na; ++e) order[e] = e;
    if (m.nvar <= 4 * (int64_t)na + (1 << 20)) {   // counting sort
      std::vector<int64_t> start((size_t)m.nvar + 1, 0);
      for (int64_t d : dest_of) ++start[(size_t)d + 1];
      for (int64_t d = 0; d < m.nvar; ++d) start[(size_t)d + 1] += start[(size_t)d];
      for (size_t e = 0; e < na; ++e) order[(size_t)start[(size_t)dest_of[e]]++] = e;
    } else {
      std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return dest_of[a] < dest_of[b]; });
    }
    G.perm.reserve(na);
    for (size_t k = 0; k < na; ++k) {
      const size_t e = order[k];
      if (k == 0 || dest_of[e] != dest_of[order[k - 1]]) { G.dest.push_back(dest_of[e]); G.seg.push_back((int64_t)k); }
      if (acc_of[e] && G.dest.back() >= 0) G.dest.back() = ~G.dest.back();   // an entry with a deferred addend: ADD to what the kernels stored
      G.perm.push_back(pos_of[e]);
    }
    G.seg.push_back((int64_t)na);
    G.park_doubles = park;
  }
  for (int kind : {(int)KK_GRAD, (int)KK_JTPROD, (int)KK_HPROD})
    for (auto &kv : dest_slots[kind]) P.covered[kind].emplace_back(kv.first, kv.first);   // written (not accumulated) by the last workgroup
  // Entries of the scatter outputs (g, Jᵀv, Hv) that no template overwrites completely must be
  // zero before the kernels run.  When NOTHING of a kind accumulates (every slot stores
  // exclusively and tiles its range), the complement is disjoint from every store and the largest
  // kernel of the kind zeroes it itself (fused memset); otherwise the runtime issues memsets.
  for (int kind : {(int)KK_GRAD, (int)KK_JTPROD, (int)KK_HPROD}) {
    auto cov = P.covered[kind];
    std::sort(cov.begin(), cov.end());
    std::vector<std::pair<int64_t, int64_t>> holes;
    int64_t pos = 0;
    for (auto &c : cov) {
      if (c.first > pos) holes.emplace_back(pos, c.first);
      pos = std::max(pos, c.second + 1);
    }
    if (pos < m.nvar) holes.emplace_back(pos, m.nvar);
    int best = -1;
    for (size_t k = 0; k < descs.size(); ++k)
      if (descs[k].kind == kind && (best < 0 || descs[k].n_blocks > descs[best].n_blocks)) best = (int)k;
    int64_t hole_len = 0;
    for (auto &h : holes) hole_len += h.second - h.first;
    // per-workgroup share bounded (64 rounds of the block) so a small kernel never serialises a big memset
    const bool fusable = opt.fuse_zero && !accumulates[kind] && best >= 0 && !holes.empty() && holes.size() <= 16;
    if (fusable && hole_len <= descs[best].n_blocks * 64 * (int64_t)opt.block)
      builders[best]->set_zero_fill(holes);
    else if (fusable && descs[best].grid[1] == 1 && descs[best].grid[2] == 1 && (hole_len + 16 * (int64_t)opt.block - 1) / (16 * (int64_t)opt.block) <= 4096) {
      // a kernel far smaller than what it must zero (an objective over a handful of variables: the OPF's
      // first-stage cost, pandemic's ∫u dt): launch EXTRA workgroups — past the grid, every lane's guard is
      // false, they only take their share of the zero fill — instead of a memset launch of its own
      const int64_t need = (hole_len + 16 * (int64_t)opt.block - 1) / (16 * (int64_t)opt.block);
      if (need > descs[best].grid[0]) { descs[best].grid[0] = need; descs[best].n_blocks = need; }
      builders[best]->set_zero_fill(holes);
    } else
      P.zero_ranges[kind] = holes;
  }
  for (int kind : {(int)KK_GRAD, (int)KK_JTPROD, (int)KK_HPROD}) {
    if (dest_slots[kind].empty()) continue;
    KernelBuilder::SharedInfo si;
    si.on = true;
    std::map<int, int> vid;   // gslot -> value id
    std::vector<int> kernel_of;
    for (size_t i = 0; i < gslots.size(); ++i)
      if (gslots[i].kind == kind && shared_dest[i]) { vid[(int)i] = (int)kernel_of.size(); kernel_of.push_back(gslots[i].kernel); }
    si.nv_total = (int64_t)kernel_of.size();
    std::map<int, int64_t> off_of;   // kernel -> first workgroup of the call
    for (size_t k = 0; k < descs.size(); ++k)
      if (descs[k].kind == kind) { off_of[(int)k] = si.n_wg; si.n_wg += descs[k].n_blocks; }
    for (int k : kernel_of) si.owner.emplace_back(off_of[k], descs[k].n_blocks);
    for (auto &kv : dest_slots[kind]) {
      std::vector<int> ids;
      for (int i : kv.second) ids.push_back(vid[i]);
      si.dests.emplace_back(kv.first, ids);
    }
    for (auto &ko : off_of) {
      KernelBuilder::SharedInfo mine = si;
      mine.red_off = ko.second;
      for (size_t i = 0; i < gslots.size(); ++i)
        if (gslots[i].kind == kind && shared_dest[i] && gslots[i].kernel == ko.first)
          mine.mine.emplace_back(gslots[i].out, gslots[i].slot, vid[(int)i]);
      builders[ko.first]->set_shared(mine);
    }
    P.red_values[kind] = si.nv_total;
    P.red_wgs[kind] = si.n_wg;
  }
  // aux buffer of a scatter kind: the shared-entry part (values x workgroups, ticket words), then the parked rows of its axis sums
  for (int kind : {(int)KK_GRAD, (int)KK_JTPROD, (int)KK_HPROD}) {
    int64_t base = P.red_values[kind] > 0 ? P.red_values[kind] * P.red_wgs[kind] + 1 + (P.red_wgs[kind] + 31) / 32 : 0;
    for (auto &ax : P.axis[kind]) {
      const GSlot &a = gslots[(size_t)(-ax.off - 1)];
      ax.off = base;
      builders[a.kernel]->outputs()[a.out].axis_off[a.slot] = base;
      base += ax.n0 * ax.rows;
    }
    if (P.gather[kind].park_doubles > 0) {   // parked addends of the plan-driven gather: offsets become absolute
      P.gather[kind].aux_off = base;
      for (int i : parked_slots[kind]) {
        const GSlot &a = gslots[i];
        builders[a.kernel]->outputs()[a.out].axis_off[a.slot] += base;
      }
      base += P.gather[kind].park_doubles;
    }
    P.aux_doubles[kind] = base;
  }
  // One launch per NLPModels call: when the templates of a call live on several support grids
  // (pandemic: t x xi and t; collocation: the node grids), the per-grid bodies become
  // __device__ functions and ONE kernel dispatches on the workgroup id (block-uniform branch,
  // shared LDS, largest grid first so that its workgroups start first).
  // The objective always takes this form, with a wrapper of its own: at most `obj_wgs` workgroups
  // WALK the tiles of every body, so there is one partial (and one ticket) per workgroup.
  // Kernels of more than one workgroup size in one program (the large-grid shape of jac_coord! / hess_coord!): every kernel
  // then sits in the namespace iem_t<size> that holds the tile-dependent device primitives compiled for its size, with
  // IEM_TILE redefined in front of it (csrc/iem_api.cpp: full_source).  A program of one size is emitted as it always was.
  bool mixed = false;
  for (const KernelDesc &d : descs) mixed = mixed || d.block != opt.block;
  auto ns_begin = [&](int tile) {
    if (mixed) src << "#undef IEM_TILE\n#define IEM_TILE " << tile << "\nnamespace iem_t" << tile << " {\n";
  };
  auto ns_end = [&](int tile) {
    if (mixed) src << "}  // namespace iem_t" << tile << "\n#undef IEM_TILE\n#define IEM_TILE " << opt.block << "\n\n";
  };
  // ---- per-kind launches ------------------------------------------------------------------------------------------------
  // A kind's bodies are emitted ONCE (__device__ functions) together with its tables; the dispatch code that routes a
  // workgroup id to a body is generated from a KindEmit at a TABLE BASE, so that the kind's own kernel (base 0) and the
  // one-launch-per-solver-phase kernels below (iem_eval_trial: obj + cons!; iem_eval_accepted: grad! + jac_coord! +
  // hess_coord! — each member kind at its own base behind one workgroup-id dispatcher) run the very same bodies.
  struct KindEmit {
    int kind = -1, tile = 0;
    std::vector<size_t> ks;                      // descs of the kind, largest grid first
    std::vector<size_t> oip, odp, ofa, oia;      // table offsets of each body
    KernelDesc F;                                // the kind's own launch; F.ip / dp / fa / ia = its tables at base 0
    size_t dec = 0, tbl = 0, sh_tbl = 0, sh_off = 0, nt_slot = 0;
    bool table = false;                          // workgroup -> body table behind the decode table
    const KernelBuilder::SharedInfo *si = nullptr;
    int sh_lds = 0;
    bool any_remap = false;
    std::vector<bool> remap;
    size_t run = 1;                              // leading bodies whose workgroups are interleaved (Options::jac_split)
  };
  std::map<int, KindEmit> emitted;
  const bool phases_on = opt.phase_kernels && !opt.no_fuse && opt.fuse_groups && !opt.hess_merge;
  auto in_a_phase = [&](int kind) { return phases_on && (kind == KK_CONS || kind == KK_GRAD || kind == KK_JAC || kind == KK_HESS); };
  // the code that takes local workgroup id `b` (of the kind's launch) to its body, tables at the given bases; OUT / AUX: the
  // pointer expressions the bodies get as their output and aux buffers.  Objective: `b` = walker index, `nw` = walkers.
  auto dispatch_code = [&](const KindEmit &E, size_t ipb, size_t dpb, size_t fab, size_t iab, const std::string &tblx, const std::string &b, const std::string &nw,
                           const std::string &OUT, const std::string &AUX, const std::string &ind0) {
    std::ostringstream c;
    const bool is_obj = E.kind == KK_OBJ;
    const auto &ks = E.ks;
    auto ipx = [&](size_t i) { return "A.ip[" + std::to_string(ipb + i) + "]"; };
    auto call = [&](size_t j, const std::string &ind) {
      std::ostringstream s;
      s << ind << (is_obj ? "acc += " : "") << descs[ks[j]].name << "_body(A.x, A.th, A.y, A.v, " << OUT << ", A.w, " << AUX << ", A.ip + " << (ipb + E.oip[j])
        << ", A.dp + " << (dpb + E.odp[j]) << ", A.fa + " << (fab + E.ofa[j]) << ", A.ia + " << (iab + E.oia[j])
        << ", lds_blk, lds4, lb % gx, (lb / gx) % gy, lb / (gx * gy), gx, gy, gz);\n";
      return s.str();
    };
    const size_t dec = E.dec;
    if (is_obj) {
      // every lane adds the terms of its tiles b, b + nw, ... in that order; the body index only grows
      c << ind0 << "double acc = 0.0;\n" << ind0 << "int j_ = 0;\n"
        << ind0 << "for (long long t_ = " << b << "; t_ < " << ipx(E.nt_slot) << "; t_ += " << nw << ") {\n";
      if (ks.size() > 1)
        c << ind0 << "  while (j_ + 1 < " << ks.size() << " && t_ >= A.ip[" << (ipb + dec) << " + 4 * (j_ + 1)]) ++j_;\n";
      c << ind0 << "  const long long gx = A.ip[" << (ipb + dec) << " + 4 * j_ + 1], gy = A.ip[" << (ipb + dec) << " + 4 * j_ + 2], gz = A.ip[" << (ipb + dec) << " + 4 * j_ + 3];\n"
        << ind0 << "  const long long lb = t_ - A.ip[" << (ipb + dec) << " + 4 * j_];\n";
      if (ks.size() == 1 && opt.obj_unroll > 1) {
        // one body: two tiles per trip (t_ and t_ + nw) so that the second tile's loads are in flight while the first is
        // summed; a tile index past the end decodes to a workgroup column outside the grid, where every lane's guard is
        // false and the body adds 0
        c << call(0, ind0 + "  ")
          << ind0 << "  { const long long b2 = t_ + " << nw << "; const bool in2 = b2 < " << ipx(E.nt_slot) << ";\n"
          << ind0 << "    const long long lb2 = b2 - A.ip[" << (ipb + dec) << " + 4 * j_];\n"
          << ind0 << "    acc += " << descs[ks[0]].name << "_body(A.x, A.th, A.y, A.v, " << OUT << ", A.w, " << AUX << ", A.ip + " << (ipb + E.oip[0]) << ", A.dp + " << (dpb + E.odp[0])
          << ", A.fa + " << (fab + E.ofa[0]) << ", A.ia + " << (iab + E.oia[0]) << ", lds_blk, lds4, in2 ? lb2 % gx : gx, in2 ? (lb2 / gx) % gy : 0, in2 ? lb2 / (gx * gy) : 0, gx, gy, gz);\n"
          << ind0 << "    t_ += " << nw << "; }\n";
      } else if (ks.size() == 1) c << call(0, ind0 + "  ");
      else {
        c << ind0 << "  switch (j_) {\n";
        for (size_t j = 0; j < ks.size(); ++j) c << ind0 << "    case " << j << ":\n" << call(j, ind0 + "      ") << ind0 << "      break;\n";
        c << ind0 << "  }\n";
      }
      c << ind0 << "}\n"
        << ind0 << "iem_block_partial(acc, " << OUT << ", " << b << ", lds4, " << nw << ", " << AUX << ");\n";
      return c.str();
    }
    if (E.si) c << ind0 << "long long wg_ = 0;\n";
    if (ks.size() > 4) {
      // many bodies (one per template on a small grid): workgroup -> body table (one scalar load), or a binary search of
      // the table of first workgroups, then a jump table — a chain of 80 compares costs microseconds
      if (E.table) c << ind0 << "const int lo_ = (int)A.ip[" << tblx << " + " << b << "];\n";
      else c << ind0 << "int lo_ = 0, hi_ = " << ks.size() << ";\n"
             << ind0 << "while (hi_ - lo_ > 1) { const int mid_ = (lo_ + hi_) >> 1; if (" << b << " >= A.ip[" << (ipb + dec) << " + 4 * mid_]) lo_ = mid_; else hi_ = mid_; }\n";
      c << ind0 << "const long long gx = A.ip[" << (ipb + dec) << " + 4 * lo_ + 1], gy = A.ip[" << (ipb + dec) << " + 4 * lo_ + 2], gz = A.ip[" << (ipb + dec) << " + 4 * lo_ + 3];\n"
        << ind0 << "const long long lb = " << (E.any_remap ? "iem_xcd_remap(" + b + " - A.ip[" + std::to_string(ipb + dec) + " + 4 * lo_], gx * gy * gz)"
                                                            : b + " - A.ip[" + std::to_string(ipb + dec) + " + 4 * lo_]") << ";\n";
      if (E.si) c << ind0 << "wg_ = A.ip[" << (ipb + E.sh_off) << " + lo_] + lb;\n";
      c << ind0 << "switch (lo_) {\n";
      for (size_t j = 0; j < ks.size(); ++j) c << ind0 << "  case " << j << ":\n" << call(j, ind0 + "    ") << ind0 << "    break;\n";
      c << ind0 << "}\n";
    } else {
      std::ostringstream ch;
      emit_dispatch_chain(ch, ks.size(), ipb + dec, E.si ? 1 : E.run, opt.jac_split == 1 ? 1 : 2, E.remap, b, call,
                          [&](size_t j) { return E.si ? "    wg_ = A.ip[" + std::to_string(ipb + E.sh_off + j) + "] + lb;\n" : std::string(); }, opt.split_shift != 0);
      c << ch.str();
    }
    if (E.si) c << KernelBuilder::shared_epilogue(*E.si, "A.ip + " + std::to_string(ipb + E.sh_tbl), "wg_", "lds_blk", E.sh_lds);
    return c.str();
  };
  auto args_struct = [&](const KernelDesc &F) {
    const size_t nip = std::max<size_t>(1, F.ip.size()), ndp = std::max<size_t>(1, F.dp.size());
    const size_t nfa = std::max<size_t>(1, F.fa.size()), nia = std::max<size_t>(1, F.ia.size());
    src << "struct Args_" << F.name << " {\n  const double* x; const double* th; const double* y; const double* v; double* out; double w; double* aux; const IemHaloArgs* comm;\n"
        << "  double* p2; double* p3; double* p4; double* p5; double* p6;\n";
    if (F.tables_in_memory)
      src << "  const long long* ip; const double* dp; const double* const* fa; const long long* const* ia;\n};\n";
    else
      src << "  long long ip[" << nip << "]; double dp[" << ndp << "]; const double* fa[" << nfa << "]; const long long* ia[" << nia << "];\n};\n";
    src << "extern \"C\" __global__ __launch_bounds__(IEM_TILE" << (opt.min_waves > 0 ? ", " + std::to_string(opt.min_waves) : std::string())
        << ") void " << F.name << "(const Args_" << F.name << " A) {\n";
  };
  for (int kind = 0; kind < KK_COUNT; ++kind) {
    std::vector<size_t> ks;
    for (size_t k = 0; k < descs.size(); ++k) if (descs[k].kind == kind) ks.push_back(k);
    if (ks.empty()) continue;
    const bool is_obj = kind == KK_OBJ;
    const int ktile = descs[ks[0]].block;   // one workgroup size per kind (kind_options)
    if (!is_obj && !in_a_phase(kind) && (ks.size() == 1 || !opt.fuse_groups || (opt.no_fuse && opt.fuse_groups < 2))) {   // fuse_groups = 2: experiments (one launch of per-template bodies)
      for (size_t k : ks) { ns_begin(ktile); src << builders[k]->emit(descs[k]); ns_end(ktile); P.kernels.push_back(descs[k]); }
      continue;
    }
    if (!is_obj && (!opt.fuse_groups || (opt.no_fuse && opt.fuse_groups < 2))) {
      for (size_t k : ks) { ns_begin(ktile); src << builders[k]->emit(descs[k]); ns_end(ktile); P.kernels.push_back(descs[k]); }
      continue;
    }
    ns_begin(ktile);
    std::stable_sort(ks.begin(), ks.end(), [&](size_t a, size_t b) { return descs[a].n_blocks > descs[b].n_blocks; });
    KindEmit &E = emitted[kind];
    E.kind = kind; E.tile = ktile; E.ks = ks;
    KernelDesc &F = E.F;
    F.name = std::string("iem_") + kname[kind] + "_all" + name_tag;
    F.kind = kind; F.block = ktile;
    F.grid[0] = 0; F.grid[1] = F.grid[2] = 1;
    for (size_t k : ks) {
      src << builders[k]->emit(descs[k], true);
      const KernelDesc &d = descs[k];
      E.oip.push_back(F.ip.size()); E.odp.push_back(F.dp.size()); E.ofa.push_back(F.fa.size()); E.oia.push_back(F.ia.size());
      F.ip.insert(F.ip.end(), d.ip.begin(), d.ip.end());
      F.dp.insert(F.dp.end(), d.dp.begin(), d.dp.end());
      F.fa.insert(F.fa.end(), d.fa.begin(), d.fa.end());
      F.ia.insert(F.ia.end(), d.ia.begin(), d.ia.end());
      F.grid[0] += d.n_blocks;
      F.lds_bytes = std::max(F.lds_bytes, d.lds_bytes);
      F.alg_bytes_read += d.alg_bytes_read; F.alg_bytes_written += d.alg_bytes_written;
      F.x_ranges.insert(F.x_ranges.end(), d.x_ranges.begin(), d.x_ranges.end());
      F.v_ranges.insert(F.v_ranges.end(), d.v_ranges.begin(), d.v_ranges.end());
      F.lds_slots = std::max(F.lds_slots, d.lds_slots);
      E.remap.push_back(kopts[k].xcd_remap != 0);
      E.any_remap = E.any_remap || kopts[k].xcd_remap;
    }
    if (F.grid[0] > 2147483647LL) throw std::runtime_error("support grids too large for one launch");
    {
      std::vector<const KernelBuilder *> bs;
      for (size_t k : ks) bs.push_back(builders[k].get());
      F.alg_bytes_read = union_read_bytes(bs);
    }
    const int64_t n_tiles = F.grid[0];
    // workgroup decode table: per body {first workgroup, gx, gy, gz} (launch-size dependent -> arguments)
    E.dec = F.ip.size();
    int64_t first = 0;
    for (size_t k : ks) {
      const KernelDesc &d = descs[k];
      F.ip.push_back(first); F.ip.push_back(d.grid[0]); F.ip.push_back(d.grid[1]); F.ip.push_back(d.grid[2]);
      first += d.n_blocks;
    }
    // scatter kinds with shared entries: the last workgroup's tables, and each body's first workgroup of
    // the call in the numbering the bodies park under (descs order, not the sorted one)
    for (size_t k : ks) if (builders[k]->shared().on) E.si = &builders[k]->shared();
    if (E.si) {
      E.sh_tbl = F.ip.size();
      const std::vector<int64_t> t = KernelBuilder::shared_final_table(*E.si);
      F.ip.insert(F.ip.end(), t.begin(), t.end());
      E.sh_off = F.ip.size();
      for (size_t k : ks) { F.ip.push_back(builders[k]->shared().red_off); E.sh_lds = std::max(E.sh_lds, builders[k]->shared_lds_doubles()); }
      F.lds_bytes = std::max(F.lds_bytes, E.sh_lds * 8);
    }
    if (is_obj) {
      E.nt_slot = F.ip.size();
      F.ip.push_back(n_tiles);
      F.grid[0] = std::min<int64_t>(n_tiles, std::max(1, opt.obj_wgs));
      F.alg_bytes_written = 8 * F.grid[0];
      P.n_partials = F.grid[0];
    }
    F.n_blocks = F.grid[0];
    E.tbl = F.ip.size();
    if (!is_obj && ks.size() > 4 && F.grid[0] <= (1 << 18)) {
      E.table = true;
      for (size_t j = 0; j < ks.size(); ++j) F.ip.insert(F.ip.end(), (size_t)descs[ks[j]].n_blocks, (int64_t)j);
    }
    // leading bodies that share an interleave class and a grid (the two halves of a split jac_coord!) take turns
    while (!is_obj && E.run < ks.size() && descs[ks[0]].inter >= 0 && descs[ks[E.run]].inter == descs[ks[0]].inter && descs[ks[E.run]].n_blocks == descs[ks[0]].n_blocks &&
           descs[ks[E.run]].grid[0] == descs[ks[0]].grid[0] && descs[ks[E.run]].grid[1] == descs[ks[0]].grid[1]) ++E.run;
    {
      const size_t nip = std::max<size_t>(1, F.ip.size()), ndp = std::max<size_t>(1, F.dp.size());
      const size_t nfa = std::max<size_t>(1, F.fa.size()), nia = std::max<size_t>(1, F.ia.size());
      // the workgroup->body table has one entry per workgroup: in device memory always, so that the
      // argument struct (hence the source) does not depend on the launch size
      F.tables_in_memory = (nip + ndp + nfa + nia) > 320 || F.ip.size() > E.tbl;
    }
    args_struct(F);
    if (F.lds_bytes > 0) src << "  __shared__ double lds_blk[" << (F.lds_bytes / 8) << "];\n";
    else src << "  double* lds_blk = nullptr;\n";
    if (is_obj) src << "  __shared__ double lds4[IEM_TILE / 64 + 1];\n";
    else src << "  double* lds4 = nullptr;\n";
    if (is_obj) {
      // (a pending halo exchange rides on this launch as one extra leading workgroup; the walkers are the others)
      F.carries = opt.carrier != 0;
      if (opt.carrier)
      src << "  const long long cb_ = A.comm != nullptr ? 1 : 0;\n"
          << "  if (cb_ && blockIdx.x == 0) { iem_halo_wg(*A.comm, const_cast<double*>(A.x)); return; }\n"
          << "  const long long bx_ = (long long)blockIdx.x - cb_, gx_ = (long long)gridDim.x - cb_;\n";
      else src << "  const long long bx_ = (long long)blockIdx.x, gx_ = (long long)gridDim.x;\n";
      src << dispatch_code(E, 0, 0, 0, 0, std::to_string(E.tbl), "bx_", "gx_", "A.out", "A.aux", "  ");
      src << "}\n\n";
      ns_end(ktile);
      P.kernels.push_back(F);
      continue;
    }
    F.carries = opt.carrier && !E.si && (kind == KK_CONS || kind == KK_JAC || kind == KK_HESS || kind == KK_JPROD);
    if (F.carries)
      src << "  const long long cb_ = A.comm != nullptr ? 1 : 0;   // a pending halo exchange rides on this launch: one extra leading workgroup\n"
          << "  if (cb_ && blockIdx.x == 0) { iem_halo_wg(*A.comm, const_cast<double*>(A.x)); return; }\n"
          << "  const long long b = (long long)blockIdx.x - cb_;\n";
    else
    src << "  const long long b = blockIdx.x;\n";
    if (E.si) src << "  double* __restrict__ OUT = A.out; double* __restrict__ AUX = A.aux;\n";
    src << dispatch_code(E, 0, 0, 0, 0, std::to_string(E.tbl), "b", "", "A.out", "A.aux", "  ");
    src << "}\n\n";
    ns_end(ktile);
    P.kernels.push_back(F);
  }
  // ---- one launch per solver phase (KK_TRIAL: obj + cons! at a trial point; KK_ACCEPTED: grad! + jac_coord! + hess_coord!
  // at an accepted point — the call pattern of ext/InfiniteExaModelsMadNLP.jl:49-50,64 and ext/InfiniteExaModelsIpopt.jl:48-49
  // of the reference).  Member kinds keep their workgroup ranges (largest member first), each decoded by its own dispatch
  // code at its own table base; bytes identical to the separate calls (same bodies).  Pointers: trial  out = c, aux = the
  // objective scalar, p2 = the objective's partials;  accepted  out = jac values, aux = hess values, p2 = g, p3 = grad!'s
  // reduction buffer.  Follow-ups of grad! (axis sums, plan-driven gather, runtime memsets) stay with the runtime.
  if (phases_on) {
    struct Member { int kind; std::string out, aux; };
    struct Phase { int id; const char *name; std::vector<Member> mem; };
    const Phase phases[] = {
      {KK_TRIAL, "iem_trial_all", {{KK_CONS, "A.out", "nullptr"}, {KK_OBJ, "A.p2", "A.aux"}}},
      {KK_ACCEPTED, "iem_accepted_all", {{KK_JAC, "A.out", "nullptr"}, {KK_HESS, "A.aux", "nullptr"}, {KK_GRAD, "A.p2", "A.p3"}}},
      // all five evaluations of one point in ONE launch (iem_eval_all: the solver's first trial point is usually accepted —
      // obj, cons!, grad!, jac_coord!, hess_coord! at the same x): p4 = c, p5 = the objective's partials, p6 = its scalar
      {KK_ALL, "iem_point_all", {{KK_JAC, "A.out", "nullptr"}, {KK_HESS, "A.aux", "nullptr"}, {KK_CONS, "A.p4", "nullptr"}, {KK_GRAD, "A.p2", "A.p3"}, {KK_OBJ, "A.p5", "A.p6"}}},
    };
    for (const Phase &ph : phases) {
      bool ok = true;
      int tile = 0;
      std::vector<Member> present;   // (a linear program has no hess_coord! kernel: the accepted point is grad! + jac_coord!)
      for (const Member &mb : ph.mem) {
        auto it = emitted.find(mb.kind);
        if (it == emitted.end()) { if (ph.id == KK_TRIAL || mb.kind == KK_OBJ || mb.kind == KK_CONS) ok = false; continue; }
        if (!tile) tile = it->second.tile;
        ok = ok && it->second.tile == tile;   // (kinds of different workgroup sizes cannot share a launch)
        present.push_back(mb);
      }
      if (!ok || present.size() < 2 || (ph.id == KK_ALL && present.size() < 3)) continue;
      KernelDesc F;
      F.name = std::string(ph.name) + name_tag;
      F.kind = ph.id; F.block = tile;
      F.grid[0] = 0; F.grid[1] = F.grid[2] = 1;
      struct Base { size_t ip, dp, fa, ia; int64_t first, n; size_t tbl; };
      std::vector<Base> base;
      std::vector<const KernelBuilder *> bs;
      bool has_obj = false, has_si = false;
      for (const Member &mb : present) {
        const KindEmit &E = emitted[mb.kind];
        // the member's tables WITHOUT its per-workgroup table (the only part whose length depends on the launch size:
        // those go behind everything else, so that every index the source names is size-independent)
        base.push_back(Base{F.ip.size(), F.dp.size(), F.fa.size(), F.ia.size(), F.grid[0], E.F.grid[0], 0});
        F.ip.insert(F.ip.end(), E.F.ip.begin(), E.F.ip.begin() + (long)E.tbl);
        F.dp.insert(F.dp.end(), E.F.dp.begin(), E.F.dp.end());
        F.fa.insert(F.fa.end(), E.F.fa.begin(), E.F.fa.end());
        F.ia.insert(F.ia.end(), E.F.ia.begin(), E.F.ia.end());
        F.grid[0] += E.F.grid[0];
        F.lds_bytes = std::max(F.lds_bytes, E.F.lds_bytes);
        F.alg_bytes_written += E.F.alg_bytes_written;
        F.x_ranges.insert(F.x_ranges.end(), E.F.x_ranges.begin(), E.F.x_ranges.end());
        F.lds_slots = std::max(F.lds_slots, E.F.lds_slots);
        for (size_t k : E.ks) bs.push_back(builders[k].get());
        has_obj = has_obj || mb.kind == KK_OBJ;
        has_si = has_si || E.si != nullptr;
      }
      if (F.grid[0] > 2147483647LL) continue;
      F.n_blocks = F.grid[0];
      F.alg_bytes_read = union_read_bytes(bs);
      const size_t mdec = F.ip.size();            // per member {first workgroup, workgroups, start of its workgroup -> body table}: launch-size dependent -> arguments
      F.ip.resize(mdec + 3 * base.size());
      for (size_t i = 0; i < base.size(); ++i) {
        const KindEmit &E = emitted[present[i].kind];
        base[i].tbl = F.ip.size();
        F.ip.insert(F.ip.end(), E.F.ip.begin() + (long)E.tbl, E.F.ip.end());
        F.ip[mdec + 3 * i] = base[i].first; F.ip[mdec + 3 * i + 1] = base[i].n; F.ip[mdec + 3 * i + 2] = (int64_t)base[i].tbl;
      }
      F.tables_in_memory = true;                  // (member tables may hold per-workgroup entries; one form for every size)
      ns_begin(tile);
      args_struct(F);
      if (F.lds_bytes > 0) src << "  __shared__ double lds_blk[" << (F.lds_bytes / 8) << "];\n";
      else src << "  double* lds_blk = nullptr;\n";
      if (has_obj) src << "  __shared__ double lds4[IEM_TILE / 64 + 1];\n";
      else src << "  double* lds4 = nullptr;\n";
      F.carries = opt.carrier && !has_si;
      if (F.carries)
        src << "  const long long cb_ = A.comm != nullptr ? 1 : 0;   // a pending halo exchange rides on this launch: one extra leading workgroup\n"
            << "  if (cb_ && blockIdx.x == 0) { iem_halo_wg(*A.comm, const_cast<double*>(A.x)); return; }\n"
            << "  const long long pb_ = (long long)blockIdx.x - cb_;\n";
      else src << "  const long long pb_ = blockIdx.x;\n";
      for (size_t i = 0; i < present.size(); ++i) {
        const KindEmit &E = emitted[present[i].kind];
        src << "  " << (i ? "else " : "");
        if (i + 1 < present.size()) src << "if (pb_ < A.ip[" << (mdec + 3 * (i + 1)) << "]) ";
        src << "{\n    const long long b = pb_ - A.ip[" << (mdec + 3 * i) << "];\n";
        if (E.si) src << "    double* __restrict__ OUT = " << present[i].out << "; double* __restrict__ AUX = " << present[i].aux << ";\n";
        src << dispatch_code(E, base[i].ip, base[i].dp, base[i].fa, base[i].ia, "A.ip[" + std::to_string(mdec + 3 * i + 2) + "]", "b", "A.ip[" + std::to_string(mdec + 3 * i + 1) + "]",
                             present[i].out, present[i].aux, "    ");
        src << "  }\n";
      }
      src << "}\n\n";
      ns_end(tile);
      P.kernels.push_back(F);
    }
  }
  // jac_coord! + hess_coord! in ONE launch (KK_PAIR; iem_jac_hess_coord).  The two calls are independent given x (and y):
  // behind one workgroup-id dispatcher their bodies share a launch — one ramp and one drain instead of two, and on a grid
  // of about one workgroup per CU (a 1/8 shard of the headline problem: 252 + 252 workgroups) both kinds are resident
  // together, 16 waves per CU instead of 8.  `out` = Jacobian values, `aux` = Hessian values; every body is generated
  // again by a builder of its own (a builder emits once), with the options its stand-alone twin got.
  if (opt.pair_kernel && !opt.no_fuse) {
    std::vector<size_t> ks;
    bool have[2] = {false, false};
    for (size_t k = 0; k < descs.size(); ++k)
      if (descs[k].kind == KK_JAC || descs[k].kind == KK_HESS) { ks.push_back(k); have[descs[k].kind == KK_HESS] = true; }
    int ptile = 0;
    bool one_tile = true;
    for (size_t k : ks) { if (!ptile) ptile = descs[k].block; one_tile = one_tile && descs[k].block == ptile; }
    if (have[0] && have[1] && one_tile) {   // (kinds of different workgroup sizes cannot share a launch: the two calls stay)
      ns_begin(ptile);
      std::vector<std::unique_ptr<KernelBuilder>> pb;
      std::vector<KernelDesc> pd;
      std::vector<bool> pxcd;
      int64_t nnz_again = 0;
      std::vector<HessClass> classes_again;
      for (size_t k : ks) {   // in the order of the first pass: the merged Hessian layout's offsets are running counters
        // (jac_coord!'s two halves, Options::jac_split, are ONE body again here unless pair_inter asks for interleaved bodies:
        // measured, the pair is fastest as jac_coord!'s workgroups followed by hess_coord!'s — 0.155 ms against 0.159 - 0.165
        // for any interleaving at 1e6 quadrotor supports, profiles/r04_ab_jac_split.txt)
        const bool rejoin = !opt.pair_inter && whole_of.count(k);
        if (!opt.pair_inter && second_half.count(k)) continue;
        KernelDesc kd;
        kd.name = (rejoin ? whole_of[k].second : descs[k].name) + "_p"; kd.kind = descs[k].kind; kd.block = descs[k].block; kd.lds_slots = descs[k].lds_slots; kd.inter = rejoin ? -1 : descs[k].inter;
        for (int d = 0; d < 3; ++d) kd.grid[d] = descs[k].grid[d];
        kd.n_blocks = descs[k].n_blocks;
        auto kb = std::make_unique<KernelBuilder>(m, rejoin ? *whole_of[k].first : builders[k]->group(), kd.kind, kopts[k], kd.name);
        if (!kb->build(nullptr)) throw std::runtime_error("internal: pair body without outputs");
        if (kd.kind == KK_HESS && opt.hess_merge) kb->merge_hess(nnz_again, classes_again);
        pb.push_back(std::move(kb));
        pd.push_back(kd);
        pxcd.push_back(kopts[k].xcd_remap != 0);
      }
      std::vector<size_t> ord(pd.size());
      for (size_t j = 0; j < ord.size(); ++j) ord[j] = j;
      std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return pd[a].n_blocks > pd[b].n_blocks; });
      KernelDesc F;
      F.name = std::string("iem_pair_all") + name_tag;
      F.kind = KK_PAIR; F.block = ptile;
      F.grid[0] = 0; F.grid[1] = F.grid[2] = 1;
      std::vector<size_t> oip, odp, ofa, oia;
      for (size_t j : ord) {
        src << pb[j]->emit(pd[j], true);
        const KernelDesc &d = pd[j];
        oip.push_back(F.ip.size()); odp.push_back(F.dp.size()); ofa.push_back(F.fa.size()); oia.push_back(F.ia.size());
        F.ip.insert(F.ip.end(), d.ip.begin(), d.ip.end());
        F.dp.insert(F.dp.end(), d.dp.begin(), d.dp.end());
        F.fa.insert(F.fa.end(), d.fa.begin(), d.fa.end());
        F.ia.insert(F.ia.end(), d.ia.begin(), d.ia.end());
        F.grid[0] += d.n_blocks;
        F.lds_bytes = std::max(F.lds_bytes, d.lds_bytes);
        F.alg_bytes_read += d.alg_bytes_read; F.alg_bytes_written += d.alg_bytes_written;
        F.x_ranges.insert(F.x_ranges.end(), d.x_ranges.begin(), d.x_ranges.end());
        F.lds_slots = std::max(F.lds_slots, d.lds_slots);
      }
      {
        std::vector<const KernelBuilder *> bs;
        for (auto &b : pb) bs.push_back(b.get());
        F.alg_bytes_read = union_read_bytes(bs);   // x7..x9, u1..u3, h are loaded by both kinds' bodies: counted once
      }
      if (F.grid[0] <= 2147483647LL) {   // (larger: the two calls stay separate launches)
        F.n_blocks = F.grid[0];
        const size_t dec = F.ip.size();
        int64_t first = 0;
        for (size_t j : ord) {
          F.ip.push_back(first); F.ip.push_back(pd[j].grid[0]); F.ip.push_back(pd[j].grid[1]); F.ip.push_back(pd[j].grid[2]);
          first += pd[j].n_blocks;
        }
        const size_t tbl = F.ip.size();
        const bool table = ord.size() > 4 && F.grid[0] <= (1 << 18);   // many bodies (templates side by side): workgroup -> body table, one scalar load
        if (table)
          for (size_t jj = 0; jj < ord.size(); ++jj) F.ip.insert(F.ip.end(), (size_t)pd[ord[jj]].n_blocks, (int64_t)jj);
        const size_t nip = std::max<size_t>(1, F.ip.size()), ndp = std::max<size_t>(1, F.dp.size());
        const size_t nfa = std::max<size_t>(1, F.fa.size()), nia = std::max<size_t>(1, F.ia.size());
        F.tables_in_memory = (nip + ndp + nfa + nia) > 320 || F.ip.size() > tbl;
        src << "struct Args_" << F.name << " {\n  const double* x; const double* th; const double* y; const double* v; double* out; double w; double* aux; const IemHaloArgs* comm;\n"
            << "  double* p2; double* p3; double* p4; double* p5; double* p6;\n";
        if (F.tables_in_memory)
          src << "  const long long* ip; const double* dp; const double* const* fa; const long long* const* ia;\n};\n";
        else
          src << "  long long ip[" << nip << "]; double dp[" << ndp << "]; const double* fa[" << nfa << "]; const long long* ia[" << nia << "];\n};\n";
        src << "extern \"C\" __global__ __launch_bounds__(IEM_TILE" << (opt.min_waves > 0 ? ", " + std::to_string(opt.min_waves) : std::string())
            << ") void " << F.name << "(const Args_" << F.name << " A) {\n";
        if (F.lds_bytes > 0) src << "  __shared__ double lds_blk[" << (F.lds_bytes / 8) << "];\n";
        else src << "  double* lds_blk = nullptr;\n";
        src << "  double* lds4 = nullptr;\n";
        F.carries = opt.carrier != 0;
        if (opt.carrier)
        src << "  const long long cb_ = A.comm != nullptr ? 1 : 0;   // a pending halo exchange rides on this launch: one extra leading workgroup\n"
            << "  if (cb_ && blockIdx.x == 0) { iem_halo_wg(*A.comm, const_cast<double*>(A.x)); return; }\n"
            << "  const long long b = (long long)blockIdx.x - cb_;\n";
        else src << "  const long long b = blockIdx.x;\n";
        auto call = [&](size_t jj, const std::string &ind) {
          const KernelDesc &d = pd[ord[jj]];
          std::ostringstream c;
          c << ind << d.name << "_body(A.x, A.th, A.y, A.v, " << (d.kind == KK_JAC ? "A.out" : "A.aux") << ", A.w, nullptr, A.ip + " << oip[jj] << ", A.dp + " << odp[jj]
            << ", A.fa + " << ofa[jj] << ", A.ia + " << oia[jj] << ", lds_blk, lds4, lb % gx, (lb / gx) % gy, lb / (gx * gy), gx, gy, gz);\n";
          return c.str();
        };
        bool pair_remap = false;
        for (bool f : pxcd) pair_remap = pair_remap || f;
        if (ord.size() > 4) {
          if (table) src << "  const int lo_ = (int)A.ip[" << tbl << " + b];\n";
          else src << "  int lo_ = 0, hi_ = " << ord.size() << ";\n"
                   << "  while (hi_ - lo_ > 1) { const int mid_ = (lo_ + hi_) >> 1; if (b >= A.ip[" << dec << " + 4 * mid_]) lo_ = mid_; else hi_ = mid_; }\n";
          src << "  const long long gx = A.ip[" << dec << " + 4 * lo_ + 1], gy = A.ip[" << dec << " + 4 * lo_ + 2], gz = A.ip[" << dec << " + 4 * lo_ + 3];\n"
              << "  const long long lb = " << (pair_remap ? "iem_xcd_remap(b - A.ip[" + std::to_string(dec) + " + 4 * lo_], gx * gy * gz)" : "b - A.ip[" + std::to_string(dec) + " + 4 * lo_]")
              << ";\n  switch (lo_) {\n";
          for (size_t jj = 0; jj < ord.size(); ++jj) src << "    case " << jj << ":\n" << call(jj, "      ") << "      break;\n";
          src << "  }\n";
        } else {
          // bodies of one grid (jac_coord!'s halves and hess_coord! of the same support grid) take turns: all resident together
          size_t run = 1;
          auto same = [&](size_t a, size_t b2) {
            return pd[ord[a]].n_blocks == pd[ord[b2]].n_blocks && pd[ord[a]].grid[0] == pd[ord[b2]].grid[0] && pd[ord[a]].grid[1] == pd[ord[b2]].grid[1] &&
                   pxcd[ord[a]] == pxcd[ord[b2]] && (opt.pair_inter || (pd[ord[a]].inter >= 0 && pd[ord[a]].inter == pd[ord[b2]].inter));
          };
          while (run < ord.size() && same(0, run)) ++run;
          std::vector<bool> rm;
          for (size_t jj = 0; jj < ord.size(); ++jj) rm.push_back(pxcd[ord[jj]]);
          emit_dispatch_chain(src, ord.size(), dec, run, opt.jac_split == 1 ? 1 : 2, rm, "b", call, [](size_t) { return std::string(); }, opt.split_shift != 0);
        }
        src << "}\n\n";
        P.kernels.push_back(F);
      }
      ns_end(ptile);
    }
  }
  P.source = src.str();
  P.key = fnv1a64(P.source);
  return P;
}

}  // namespace iem
