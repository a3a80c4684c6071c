// iem_shard.hpp — cut one rank's shard out of a GLOBAL transcribed model, in C++ (behind
// iem_create_sharded / iem_shard_blob), so that a host that only knows the global blob — the
// reference's Julia host, /root/reference/src/infiniteopt_backend.jl:155-156 — reaches the
// multi-GPU path through the C-ABI.
//
// The reference has no sharding (single process, single device).  What is cut follows from its
// data layout: an infinite variable is one contiguous slab, first parameter fastest
// (transform.jl:140-156, test/transcription.jl:44-57), every template item owns its row and
// its COO slots, and the only coupling between supports of one parameter is the derivative
// stencil, whose index expressions are `group_idx ± const` (transform.jl:471-506, 535-557).
//
// Rank r of `world` owns the contiguous block [a, b) of the sharded parameter group's supports.
//   * x: every slab that runs over the group keeps the WINDOW [a - h, b) of that axis (h = the
//     stencil's reach to the left, at most `a`); other slabs (finite / first-stage variables,
//     variables over other parameters) are replicated.  The local x is the concatenation of the
//     local slabs in the global order; `var_map` gives local -> global.
//   * theta and every item-data column stay GLOBAL and resident on every rank (read-only,
//     8 bytes per support each): parameter indices and data positions are untouched, so measure
//     coefficients are those of the global grid by construction.
//   * a template that iterates over the group is cut to the items whose support the rank owns;
//     a template that does not is kept by the rank that owns the supports of its point variables
//     (x(0) == 0 -> the rank holding t = 0), else by rank 0.
//   * variable index expressions are re-based into the local numbering (they become affine
//     fields of the local item box);
//   * templates that walk the sharded axis without iterating over its support grid (orthogonal
//     collocation: node x element boxes with indices like 2e + j, constant-over-collocation pairs) are
//     cut along the dimension with the largest stride, whole elements at a time; an item belongs to
//     the rank owning the LAST support it references, the halo covers the first.
// Supported: templates on support grids (grid hint present) with affine integer fields, stencils
// that reach to the LEFT only (backward differences, orthogonal collocation) — what the reference's
// derivative methods emit; and explicit item lists (a domain restriction filters the iterator,
// transform.jl:448-451) whose variable indices all sit at the item's own support: the list is
// filtered to the owned supports and every column re-gathered.  Anything else throws with a message
// (IEM_E_BLOB at the ABI).
#pragma once
#include <algorithm>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "iem_model.hpp"

namespace iem {

struct ShardTpl {           // a local template and where it sits in the global model
  int64_t gindex = 0;       // global template index
  int64_t klo[3] = {0, 0, 0};   // local item coordinate 0 = global item coordinate klo
  int64_t gdims[3] = {1, 1, 1}; // the global item box
  int64_t go0 = 0, go1 = 0, go2 = 0;
  bool explicit_items = false;  // an explicit item list (domain restriction, transform.jl:448-451): local item j = global item items[j]
  std::vector<int64_t> items;
};

struct HaloSeg {            // one sharded slab of the LOCAL x (window length wn along the sharded axis)
  int64_t loff = 0, inner = 1, outer = 1, wn = 0;
};

struct ShardInfo {
  int group = 0, rank = 0, world = 1;
  int64_t n_global = 0;     // supports of the sharded group
  int64_t own_lo = 0, own_n = 0, halo = 0;   // global index of the first OWNED support, count, halo supports in front
  int64_t halo_reach = 0;   // the model's stencil reach (what ranks > 0 carry)
  int64_t nvar_global = 0, ncon_global = 0, nnzj_global = 0, nnzh_global = 0;
  std::vector<int64_t> var_map;        // local variable -> global variable (0-based)
  std::vector<unsigned char> var_flag; // bit 0: owned by this rank, bit 1: replicated on every rank, bit 2: halo copy
  std::vector<ShardTpl> tpl;
  std::vector<HaloSeg> segs;           // sharded slabs, in local order
  int64_t halo_doubles = 0;            // doubles one neighbour sends the next: sum over segs of outer * halo_reach * inner
};

inline void partition_block(int64_t n, int world, int rank, int64_t &a, int64_t &b) {
  const int64_t base = n / world, rem = n % world;
  a = rank * base + std::min<int64_t>(rank, rem);
  b = a + base + (rank < rem ? 1 : 0);
}

namespace shard_detail {

struct Affine {   // value(k) = c + sum_d k[d] * kd  over the template's item box
  __int128 c = 0;
  int64_t k[3] = {0, 0, 0};
  bool ok = true;   // false: a gathered field takes part
};

inline Affine idx_affine(const Template &t, const IdxExpr &ix) {
  // blobs come from foreign producers: 128-bit arithmetic, and anything that does not fit an index is refused
  Affine a;
  __int128 c = ix.c0, k[3] = {0, 0, 0};
  for (int j = 0; j < ix.nterms; ++j) {
    const FieldDesc &f = t.ifields[ix.field[j]];
    if (f.mode != IEM_F_AFFINE) { a.ok = false; continue; }
    c += (__int128)ix.coef[j] * f.base;
    for (int d = 0; d < 3; ++d) k[d] += (__int128)ix.coef[j] * f.step[d];
  }
  const __int128 lim = (__int128)1 << 50;
  if (c > lim || c < -lim) throw std::runtime_error("index expression out of range");
  for (int d = 0; d < 3; ++d) {
    if (k[d] > lim || k[d] < -lim) throw std::runtime_error("index stride out of range");
    a.k[d] = (int64_t)k[d];
  }
  a.c = c;
  return a;
}

inline int dim_group(const Template &t, int d) {   // group id of item dimension d from the grid hint (0: unknown)
  if (t.grid_id <= 0 || t.lattice_recovered) return 0;
  int64_t g = t.grid_id;
  for (int e = t.nd - 1; e > d; --e) g /= 4096;
  return (int)(g % 4096) - 1;
}

inline int find_slab(const std::vector<Slab> &slabs, int64_t v0 /*0-based variable*/) {
  int lo = 0, hi = (int)slabs.size();
  while (hi - lo > 1) {
    const int mid = (lo + hi) / 2;
    if (slabs[mid].off <= v0) lo = mid; else hi = mid;
  }
  return lo;
}

// how one variable index expression of a template walks its slab
struct Walk {
  int slab = -1;
  int64_t i0[3] = {0, 0, 0};    // slab coordinates (0-based) at item k = 0
  int64_t m[3][3] = {};         // m[d][a]: slab axis a advances by m per unit of item coordinate d
};

inline Walk walk_of(const Model &G, const Template &t, size_t ti, const IdxExpr &ix) {
  const Affine a = idx_affine(t, ix);
  const std::string where = "template " + std::to_string(ti);
  if (!a.ok) throw std::runtime_error(where + ": sharding needs affine item fields in variable indices (explicit index columns are not supported)");
  Walk w;
  const int64_t v0 = (int64_t)a.c - 1;
  if (v0 < 0 || v0 >= G.nvar) throw std::runtime_error(where + ": variable index out of range");
  w.slab = find_slab(G.slabs, v0);
  const Slab &sl = G.slabs[w.slab];
  const int64_t stride[3] = {1, sl.dims[0], sl.dims[0] * sl.dims[1]};
  int64_t rem = v0 - sl.off;
  for (int ax = 2; ax >= 0; --ax) { w.i0[ax] = stride[ax] ? rem / stride[ax] : 0; rem -= w.i0[ax] * stride[ax]; }
  for (int d = 0; d < t.nd; ++d) {
    if (a.k[d] == 0 || t.dims[d] <= 1) continue;
    int ax = -1;
    const int g = dim_group(t, d);
    if (g > 0)
      for (int e = 0; e < sl.nd; ++e) if (sl.group[e] == g) ax = e;
    if (ax < 0)   // no hint: the largest slab stride that divides the item stride
      for (int e = sl.nd - 1; e >= 0; --e)
        if (stride[e] != 0 && a.k[d] % stride[e] == 0 && sl.dims[e] > 1) { ax = e; break; }
    if (ax < 0 || a.k[d] % stride[ax] != 0) throw std::runtime_error(where + ": cannot match an item dimension to a slab axis");
    w.m[d][ax] = a.k[d] / stride[ax];
  }
  // the whole box must stay inside the slab (validate_indices only bounds it by nvar)
  for (int ax = 0; ax < 3; ++ax) {
    int64_t lo = w.i0[ax], hi = w.i0[ax];
    for (int d = 0; d < t.nd; ++d) {
      const int64_t e = w.m[d][ax] * (t.dims[d] - 1);
      if (e < 0) lo += e; else hi += e;
    }
    if (lo < 0 || hi >= sl.dims[ax]) throw std::runtime_error(where + ": a variable index expression leaves its slab");
  }
  return w;
}

}  // namespace shard_detail

// In place: `m` (the parsed GLOBAL model, slab table required) becomes rank `rank`'s shard.
inline void shard_model(Model &m, int group, int rank, int world, ShardInfo &info) {
  using namespace shard_detail;
  if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("bad rank / world");
  if (group < 1 || group > 4094) throw std::runtime_error("bad sharded group id");
  if (m.slabs.empty()) throw std::runtime_error("sharding needs the blob's slab table (header word 9)");
  info = ShardInfo();
  info.group = group; info.rank = rank; info.world = world;
  info.nvar_global = m.nvar; info.ncon_global = m.ncon; info.nnzj_global = m.nnzj; info.nnzh_global = m.nnzh;
  // supports of the sharded group
  int64_t ng = -1;
  std::vector<int> sax(m.slabs.size(), -1);   // sharded axis of each slab
  for (size_t s = 0; s < m.slabs.size(); ++s)
    for (int a = 0; a < m.slabs[s].nd; ++a)
      if (m.slabs[s].group[a] == group) {
        if (sax[s] >= 0) throw std::runtime_error("a variable runs over the sharded group twice");
        sax[s] = a;
        if (ng >= 0 && ng != m.slabs[s].dims[a]) throw std::runtime_error("slabs disagree on the sharded group's support count");
        ng = m.slabs[s].dims[a];
      }
  if (ng < 0) throw std::runtime_error("no variable runs over the sharded group");
  if (ng < world) throw std::runtime_error("fewer supports than ranks");
  info.n_global = ng;
  int64_t a0, b0;
  partition_block(ng, world, rank, a0, b0);
  info.own_lo = a0; info.own_n = b0 - a0;

  // an index expression shared by a variable node and a parameter node (same numbers, different
  // arrays) must not be re-based for both: give the parameter nodes a copy
  for (Template &t : m.tpl) {
    std::vector<char> v(t.idx.size(), 0), p(t.idx.size(), 0);
    for (const Node &nd : t.nodes) { if (nd.op == IEM_OP_VAR) v[nd.a] = 1; if (nd.op == IEM_OP_PAR) p[nd.a] = 1; }
    for (size_t i = 0, n0 = t.idx.size(); i < n0; ++i) {
      if (!(v[i] && p[i])) continue;
      const int copy = (int)t.idx.size();
      t.idx.push_back(t.idx[i]);
      for (Node &nd : t.nodes) if (nd.op == IEM_OP_PAR && nd.a == (int)i) nd.a = copy;
    }
  }
  // pass 1: which item dimension of each template runs over the group; stencil reach
  const size_t nt = m.tpl.size();
  std::vector<int> sdim(nt, -1);
  std::vector<int> gdim(nt, -1);            // un-hinted templates that walk the sharded axis: the item dimension to cut,
  std::vector<int64_t> gm(nt, 0), ganchor(nt, 0);   // its stride in supports, and the last referenced support at item 0
  std::vector<char> explicit_tpl(nt, 0);
  std::vector<std::vector<Walk>> walks(nt);
  int64_t reach = 0;
  for (size_t ti = 0; ti < nt; ++ti) {
    const Template &t = m.tpl[ti];
    std::vector<char> is_var(t.idx.size(), 0);
    for (const Node &nd : t.nodes) if (nd.op == IEM_OP_VAR) is_var[nd.a] = 1;
    for (int d = 0; d < t.nd; ++d) if (dim_group(t, d) == group) sdim[ti] = d;
    walks[ti].resize(t.idx.size());
    if (t.n_items == 0) continue;
    bool gathered = false;
    for (size_t i = 0; i < t.idx.size(); ++i)
      if (is_var[i] && !idx_affine(t, t.idx[i]).ok) gathered = true;
    if (gathered) {   // explicit item list: handled item by item in pass 2
      if (t.nd != 1 || sdim[ti] >= 0) throw std::runtime_error("template " + std::to_string(ti) + ": explicit index columns on a support-grid template cannot be sharded");
      explicit_tpl[ti] = 1;
      continue;
    }
    const bool hinted = sdim[ti] >= 0;   // iterates over the sharded group's own support grid
    int64_t a_max = INT64_MIN, c_min = INT64_MAX;
    for (size_t i = 0; i < t.idx.size(); ++i) {
      if (!is_var[i]) continue;
      walks[ti][i] = walk_of(m, t, ti, t.idx[i]);
      const Walk &w = walks[ti][i];
      const int ax = sax[w.slab];
      if (ax < 0) continue;
      if (hinted) {
        for (int d = 0; d < t.nd; ++d) {
          if (w.m[d][ax] == 0) continue;
          if (sdim[ti] != d) throw std::runtime_error("template " + std::to_string(ti) + ": two item dimensions run over the sharded group");
          if (w.m[d][ax] != 1) throw std::runtime_error("template " + std::to_string(ti) + ": strided access along the sharded group's own grid is not supported");
        }
        if (w.m[sdim[ti]][ax] == 1) {
          const int64_t shift = w.i0[ax] - t.origin[sdim[ti]];   // slab coordinate minus the item's own support
          if (shift > 0) throw std::runtime_error("template " + std::to_string(ti) + ": stencils that reach to the right of their support are not supported (backward differences only)");
          reach = std::max(reach, -shift);
        }
        continue;
      }
      // No support-grid hint, but the template walks the sharded axis (collocation rows over node x element
      // boxes, transform.jl:535-557: indices like 2e + j): the item dimension with the LARGEST stride
      // (the element) is the one to cut, whole elements at a time; an item belongs to the rank that
      // owns the LAST support it references (for a backward difference that is the row's own support),
      // and the stencil's reach is how far the first referenced support lies before it.
      int dstar = -1;
      for (int d = 0; d < t.nd; ++d) {
        if (w.m[d][ax] < 0) throw std::runtime_error("template " + std::to_string(ti) + ": an index runs backwards along the sharded group");
        if (w.m[d][ax] > 0 && (dstar < 0 || w.m[d][ax] > w.m[dstar][ax])) dstar = d;
      }
      if (dstar < 0) continue;   // a fixed support (point variable): decided in pass 2
      if (gdim[ti] >= 0 && (gdim[ti] != dstar || gm[ti] != w.m[dstar][ax]))
        throw std::runtime_error("template " + std::to_string(ti) + ": index expressions walk the sharded group with different strides");
      gdim[ti] = dstar; gm[ti] = w.m[dstar][ax];
      int64_t hi = w.i0[ax];
      for (int d = 0; d < t.nd; ++d) if (d != dstar) hi += w.m[d][ax] * (t.dims[d] - 1);
      a_max = std::max(a_max, hi);
      c_min = std::min(c_min, w.i0[ax]);
    }
    if (gdim[ti] >= 0) {
      // every referenced support that does NOT move with the cut dimension must move with it too (all walks
      // into sharded slabs share the stride): a walk that stays put would leave the window
      for (size_t i = 0; i < t.idx.size(); ++i) {
        if (!is_var[i]) continue;
        const Walk &w = walks[ti][i];
        if (sax[w.slab] >= 0 && w.m[gdim[ti]][sax[w.slab]] != gm[ti])
          throw std::runtime_error("template " + std::to_string(ti) + ": mixes moving and fixed supports of the sharded group");
      }
      ganchor[ti] = a_max;
      reach = std::max(reach, a_max - c_min);
    }
  }
  info.halo_reach = reach;
  const int64_t h = std::min(reach, a0);
  info.halo = h;
  const int64_t wlo = a0 - h, wn = b0 - a0 + h;

  // local slabs, variable map
  std::vector<Slab> ls(m.slabs.size());
  int64_t lnvar = 0;
  for (size_t s = 0; s < m.slabs.size(); ++s) {
    ls[s] = m.slabs[s];
    ls[s].off = lnvar;
    if (sax[s] >= 0) ls[s].dims[sax[s]] = wn;
    lnvar += ls[s].length();
  }
  info.var_map.resize((size_t)lnvar);
  info.var_flag.assign((size_t)lnvar, 0);
  for (size_t s = 0; s < m.slabs.size(); ++s) {
    const Slab &g = m.slabs[s], &l = ls[s];
    const int ax = sax[s];
    for (int64_t i2 = 0; i2 < l.dims[2]; ++i2)
      for (int64_t i1 = 0; i1 < l.dims[1]; ++i1)
        for (int64_t i0 = 0; i0 < l.dims[0]; ++i0) {
          const int64_t li[3] = {i0, i1, i2};
          int64_t gi[3] = {i0, i1, i2};
          if (ax >= 0) gi[ax] += wlo;
          const int64_t lv = l.off + i0 + l.dims[0] * (i1 + l.dims[1] * i2);
          info.var_map[(size_t)lv] = g.off + gi[0] + g.dims[0] * (gi[1] + g.dims[1] * gi[2]);
          info.var_flag[(size_t)lv] = ax < 0 ? (unsigned char)(2 | (rank == 0 ? 1 : 0)) : (li[ax] >= h ? 1 : 4);
        }
    if (ax >= 0) {
      HaloSeg sg;
      sg.loff = l.off; sg.wn = wn; sg.inner = 1; sg.outer = 1;
      for (int e = 0; e < ax; ++e) sg.inner *= l.dims[e];
      for (int e = ax + 1; e < 3; ++e) sg.outer *= l.dims[e];
      info.segs.push_back(sg);
      info.halo_doubles += sg.outer * reach * sg.inner;
    }
  }

  // core arrays of the local variables
  auto gather_core = [&](int &arr_id) {
    const ArrayDesc src = m.arrs[arr_id];
    std::vector<double> v((size_t)lnvar);
    for (int64_t i = 0; i < lnvar; ++i) v[(size_t)i] = src.f(info.var_map[(size_t)i]);
    m.synth.push_back(std::move(v));
    ArrayDesc a;
    a.kind = IEM_A_F64_DATA; a.n = lnvar; a.data = m.synth.back().data();
    arr_id = (int)m.arrs.size();
    m.arrs.push_back(a);
  };
  gather_core(m.arr_x0); gather_core(m.arr_lvar); gather_core(m.arr_uvar);

  // pass 2: cut / keep / drop the templates, re-base their variable indices
  std::vector<Template> kept;
  int64_t o0 = 0, o1 = 0, o2 = 0;
  for (size_t ti = 0; ti < nt; ++ti) {
    Template t = m.tpl[ti];
    const std::string where = "template " + std::to_string(ti);
    ShardTpl st;
    st.gindex = (int64_t)ti; st.go0 = t.o0; st.go1 = t.o1; st.go2 = t.o2;
    for (int d = 0; d < 3; ++d) st.gdims[d] = t.dims[d];
    std::vector<char> is_var(t.idx.size(), 0);
    for (const Node &nd : t.nodes) if (nd.op == IEM_OP_VAR) is_var[nd.a] = 1;
    int64_t klo[3] = {0, 0, 0};
    if (explicit_tpl[ti]) {
      // item by item: where does each variable index sit, which items does this rank own
      const int64_t n = t.n_items;
      auto ifield_at = [&](const FieldDesc &f, int64_t k) { const int64_t p = f.base + f.step[0] * k; return f.mode == IEM_F_AFFINE ? p : m.arrs[f.arr].i(p); };
      std::vector<std::vector<int64_t>> lidx(t.idx.size());   // per variable index expression: local 1-based index per item (0: outside the window)
      std::vector<int64_t> coord((size_t)n, -1);              // the item's support on the sharded axis (-1: touches no sharded slab)
      for (size_t i = 0; i < t.idx.size(); ++i) {
        if (!is_var[i]) continue;
        lidx[i].assign((size_t)n, 0);
        for (int64_t k = 0; k < n; ++k) {
          __int128 v128 = t.idx[i].c0;
          for (int j = 0; j < t.idx[i].nterms; ++j) v128 += (__int128)t.idx[i].coef[j] * ifield_at(t.ifields[t.idx[i].field[j]], k);
          if (v128 < 1 || v128 > (__int128)info.nvar_global) throw std::runtime_error(where + ": variable index out of range");
          const int64_t v = (int64_t)v128;
          const int si = find_slab(m.slabs, v - 1);
          const Slab &g = m.slabs[si], &l = ls[si];
          int64_t rem = v - 1 - g.off, c3[3];
          c3[0] = rem % g.dims[0]; rem /= g.dims[0]; c3[1] = rem % g.dims[1]; c3[2] = rem / g.dims[1];
          const int ax = sax[si];
          if (ax >= 0) {
            if (coord[(size_t)k] >= 0 && coord[(size_t)k] != c3[ax]) throw std::runtime_error(where + ": an explicit item list with a stencil along the sharded group is not supported");
            coord[(size_t)k] = c3[ax];
            c3[ax] -= wlo;
            if (c3[ax] < 0 || c3[ax] >= wn) continue;   // outside this rank's window: the item is not kept (checked below)
          }
          lidx[i][(size_t)k] = l.off + 1 + c3[0] + l.dims[0] * (c3[1] + l.dims[1] * c3[2]);
        }
      }
      int64_t cmin = -1, cmax = -1;
      for (int64_t k = 0; k < n; ++k) if (coord[(size_t)k] >= 0) { cmin = cmin < 0 ? coord[(size_t)k] : std::min(cmin, coord[(size_t)k]); cmax = std::max(cmax, coord[(size_t)k]); }
      std::vector<int64_t> keep;
      if (cmin >= 0 && cmin != cmax) {            // runs over the sharded group: the owned supports
        for (int64_t k = 0; k < n; ++k) {
          if (coord[(size_t)k] < 0) throw std::runtime_error(where + ": an explicit item list mixes items on and off the sharded group");
          if (coord[(size_t)k] >= a0 && coord[(size_t)k] < b0) keep.push_back(k);
        }
      } else {                                     // one support (or none): like a point-variable template
        int owner = 0;
        if (cmin >= 0) for (; owner < world; ++owner) { int64_t a, b; partition_block(ng, world, owner, a, b); if (cmin >= a && cmin < b) break; }
        if (owner != rank) continue;
        for (int64_t k = 0; k < n; ++k) keep.push_back(k);
      }
      const int64_t nk = (int64_t)keep.size();
      auto new_i64 = [&](std::vector<int64_t> v) {
        m.synth_i.push_back(std::move(v));
        ArrayDesc a; a.kind = IEM_A_I64_DATA; a.n = (int64_t)m.synth_i.back().size(); a.data = m.synth_i.back().data();
        m.arrs.push_back(a);
        return (int)m.arrs.size() - 1;
      };
      auto new_f64 = [&](std::vector<double> v) {
        m.synth.push_back(std::move(v));
        ArrayDesc a; a.kind = IEM_A_F64_DATA; a.n = (int64_t)m.synth.back().size(); a.data = m.synth.back().data();
        m.arrs.push_back(a);
        return (int)m.arrs.size() - 1;
      };
      auto as_column = [](FieldDesc &f, int arr) { f.mode = IEM_F_GATHER; f.base = 0; f.step[0] = 1; f.step[1] = f.step[2] = 0; f.arr = arr; };
      for (FieldDesc &f : t.ifields) {
        std::vector<int64_t> col((size_t)nk);
        for (int64_t j = 0; j < nk; ++j) col[(size_t)j] = ifield_at(f, keep[(size_t)j]);
        as_column(f, new_i64(std::move(col)));
      }
      for (FieldDesc &f : t.ffields) {
        std::vector<double> col((size_t)nk);
        for (int64_t j = 0; j < nk; ++j) col[(size_t)j] = m.arrs[f.arr].f(f.base + f.step[0] * keep[(size_t)j]);
        as_column(f, new_f64(std::move(col)));
      }
      for (size_t i = 0; i < t.idx.size(); ++i) {
        if (!is_var[i]) continue;
        std::vector<int64_t> col((size_t)nk);
        for (int64_t j = 0; j < nk; ++j) {
          col[(size_t)j] = lidx[i][(size_t)keep[(size_t)j]];
          if (col[(size_t)j] == 0) throw std::runtime_error(where + ": a variable index leaves the rank's window");
        }
        FieldDesc f;
        as_column(f, new_i64(std::move(col)));
        IdxExpr &ix = t.idx[i];
        ix.c0 = 0; ix.nterms = 1; ix.field[0] = (int)t.ifields.size(); ix.coef[0] = 1;
        for (int j = 1; j < IEM_MAX_IDX_TERMS; ++j) { ix.field[j] = 0; ix.coef[j] = 0; }
        t.ifields.push_back(f);
      }
      auto cut_bound = [&](int mode, int &arr) {
        if (mode != 1) return;
        std::vector<double> col((size_t)nk);
        for (int64_t j = 0; j < nk; ++j) col[(size_t)j] = m.arrs[arr].f(keep[(size_t)j]);
        arr = new_f64(std::move(col));
      };
      cut_bound(t.lmode, t.larr); cut_bound(t.umode, t.uarr);
      t.n_items = nk; t.dims[0] = nk; t.dims[1] = t.dims[2] = 1;
      t.grid_id = -1; t.origin[0] = t.origin[1] = t.origin[2] = 0;
      st.explicit_items = true; st.items = keep;
      t.o2 = o2; o2 += t.n_items * t.o2step;
      if (t.kind == IEM_T_CON) { t.o0 = o0; o0 += t.n_items; t.o1 = o1; o1 += t.n_items * t.o1step; }
      kept.push_back(std::move(t));
      info.tpl.push_back(st);
      continue;
    }
    if (sdim[ti] >= 0 || gdim[ti] >= 0) {
      const int d = sdim[ti] >= 0 ? sdim[ti] : gdim[ti];
      int64_t lo, hi;
      if (sdim[ti] >= 0) {
        lo = std::min(std::max<int64_t>(a0 - t.origin[d], 0), t.dims[d]);
        hi = std::min(std::max<int64_t>(b0 - t.origin[d], 0), t.dims[d]);
      } else {   // items whose last referenced support  ganchor + gm*k  this rank owns
        auto ceil_div = [](int64_t p, int64_t q) { return p >= 0 ? (p + q - 1) / q : -((-p) / q); };
        lo = std::min(std::max<int64_t>(ceil_div(a0 - ganchor[ti], gm[ti]), 0), t.dims[d]);
        hi = std::min(std::max<int64_t>(ceil_div(b0 - ganchor[ti], gm[ti]), 0), t.dims[d]);
      }
      klo[d] = lo;
      // per-item bound arrays follow the cut
      auto cut_bound = [&](int mode, int &arr) {
        if (mode != 1) return;
        const ArrayDesc src = m.arrs[arr];
        int64_t nd_[3] = {t.dims[0], t.dims[1], t.dims[2]};
        nd_[d] = std::max<int64_t>(hi - lo, 0);
        std::vector<double> v((size_t)(nd_[0] * nd_[1] * nd_[2]));
        size_t p = 0;
        for (int64_t k2 = 0; k2 < nd_[2]; ++k2)
          for (int64_t k1 = 0; k1 < nd_[1]; ++k1)
            for (int64_t k0 = 0; k0 < nd_[0]; ++k0) {
              int64_t g[3] = {k0, k1, k2};
              g[d] += lo;
              v[p++] = src.f(g[0] + t.dims[0] * (g[1] + t.dims[1] * g[2]));
            }
        m.synth.push_back(std::move(v));
        ArrayDesc a;
        a.kind = IEM_A_F64_DATA; a.n = (int64_t)m.synth.back().size(); a.data = m.synth.back().data();
        arr = (int)m.arrs.size();
        m.arrs.push_back(a);
      };
      cut_bound(t.lmode, t.larr); cut_bound(t.umode, t.uarr);
      for (FieldDesc &f : t.ifields) f.base += f.step[d] * lo;
      for (FieldDesc &f : t.ffields) f.base += f.step[d] * lo;
      t.dims[d] = std::max<int64_t>(hi - lo, 0);
      if (sdim[ti] >= 0) t.origin[d] = t.origin[d] + lo - wlo;   // grid coordinate inside the local window
      t.n_items = t.dims[0] * t.dims[1] * t.dims[2];
    } else {
      // not over the sharded group: the rank that owns the supports of its point variables, else rank 0
      int owner = -1;
      bool any = false;
      if (t.n_items > 0)
        for (size_t i = 0; i < t.idx.size(); ++i) {
          if (!is_var[i]) continue;
          const Walk &w = walks[ti][i];
          const int ax = sax[w.slab];
          if (ax < 0) continue;
          any = true;
          int r = 0;
          for (; r < world; ++r) { int64_t a, b; partition_block(ng, world, r, a, b); if (w.i0[ax] >= a && w.i0[ax] < b) break; }
          if (owner >= 0 && owner != r) throw std::runtime_error(where + ": a finite template couples point variables owned by different ranks");
          owner = r;
        }
      if (!any) owner = 0;
      if (owner != rank) continue;
    }
    // variable indices -> local numbering: one fresh affine field per variable index expression
    if (t.n_items > 0)
      for (size_t i = 0; i < t.idx.size(); ++i) {
        if (!is_var[i]) continue;
        const Walk &w = walks[ti][i];
        const Slab &l = ls[w.slab];
        const int ax = sax[w.slab];
        const int64_t lstride[3] = {1, l.dims[0], l.dims[0] * l.dims[1]};
        int64_t c = l.off + 1, k[3] = {0, 0, 0};
        for (int a = 0; a < 3; ++a) {
          int64_t at0 = w.i0[a];                        // slab coordinate at the first LOCAL item
          for (int d = 0; d < t.nd; ++d) at0 += w.m[d][a] * klo[d];
          if (a == ax) {
            at0 -= wlo;
            int64_t lo = at0, hi = at0;
            for (int d = 0; d < t.nd; ++d) { const int64_t e = w.m[d][a] * (t.dims[d] - 1); if (e < 0) lo += e; else hi += e; }
            if (lo < 0 || hi >= wn) throw std::runtime_error(where + ": a variable index leaves the rank's window (halo too small)");
          }
          c += lstride[a] * at0;
          for (int d = 0; d < t.nd; ++d) k[d] += lstride[a] * w.m[d][a];
        }
        FieldDesc f;
        f.mode = IEM_F_AFFINE; f.base = c; f.arr = -1;
        for (int d = 0; d < 3; ++d) f.step[d] = k[d];
        IdxExpr &ix = t.idx[i];
        ix.c0 = 0; ix.nterms = 1;
        ix.field[0] = (int)t.ifields.size(); ix.coef[0] = 1;
        for (int j = 1; j < IEM_MAX_IDX_TERMS; ++j) { ix.field[j] = 0; ix.coef[j] = 0; }
        t.ifields.push_back(f);
      }
    for (int d = 0; d < 3; ++d) st.klo[d] = klo[d];
    t.o2 = o2; o2 += t.n_items * t.o2step;
    if (t.kind == IEM_T_CON) { t.o0 = o0; o0 += t.n_items; t.o1 = o1; o1 += t.n_items * t.o1step; }
    kept.push_back(std::move(t));
    info.tpl.push_back(st);
  }
  m.tpl = std::move(kept);
  m.slabs = ls;
  m.nvar = lnvar; m.ncon = o0; m.nnzj = o1; m.nnzh = o2;
}

// Re-serialise a parsed (possibly sharded) model as a blob of include/iem_blob.h.  Arrays nothing
// references any more (the global x0 of a shard) are dropped.
inline std::vector<int64_t> serialize_model(const Model &m) {
  std::vector<int> remap(m.arrs.size(), -1);
  std::vector<int> order;
  auto use = [&](int id) {
    if (id < 0) return;
    if (remap[id] < 0) { remap[id] = (int)order.size(); order.push_back(id); }
  };
  use(m.arr_x0); use(m.arr_lvar); use(m.arr_uvar); use(m.arr_theta);
  for (const Template &t : m.tpl) {
    if (t.lmode == 1) use(t.larr);
    if (t.umode == 1) use(t.uarr);
    for (const FieldDesc &f : t.ifields) if (f.mode == IEM_F_GATHER) use(f.arr);
    for (const FieldDesc &f : t.ffields) use(f.arr);
  }
  auto d2w = [](double d) { int64_t w; std::memcpy(&w, &d, 8); return w; };
  std::vector<std::vector<int64_t>> tw(m.tpl.size());
  for (size_t i = 0; i < m.tpl.size(); ++i) {
    const Template &t = m.tpl[i];
    std::vector<int64_t> &w = tw[i];
    w = {t.kind, t.n_items, t.nd, t.dims[0], t.dims[1], t.dims[2], t.lattice_recovered ? -1 : t.grid_id, t.origin[0], t.origin[1], t.origin[2],
         (int64_t)t.ifields.size(), (int64_t)t.ffields.size(), (int64_t)t.idx.size(), (int64_t)t.nodes.size(), t.root,
         t.lmode, d2w(t.lval), t.lmode == 1 ? remap[t.larr] : -1, t.umode, d2w(t.uval), t.umode == 1 ? remap[t.uarr] : -1};
    auto put_field = [&](const FieldDesc &f) {
      w.push_back(f.mode); w.push_back(f.base);
      for (int d = 0; d < 3; ++d) w.push_back(f.step[d]);
      w.push_back(f.mode == IEM_F_GATHER ? remap[f.arr] : -1);
    };
    for (const FieldDesc &f : t.ifields) put_field(f);
    for (const FieldDesc &f : t.ffields) put_field(f);
    for (const IdxExpr &ix : t.idx) {
      w.push_back(ix.c0); w.push_back(ix.nterms);
      for (int j = 0; j < IEM_MAX_IDX_TERMS; ++j) { w.push_back(j < ix.nterms ? ix.field[j] : 0); w.push_back(j < ix.nterms ? ix.coef[j] : 0); }
    }
    for (const Node &nd : t.nodes) { w.push_back(nd.op); w.push_back(nd.a); w.push_back(nd.b); w.push_back(d2w(nd.imm)); }
  }
  const int64_t n_arr = (int64_t)order.size(), n_tpl = (int64_t)m.tpl.size();
  int64_t pos = IEM_HDR_WORDS + IEM_ARR_WORDS * n_arr + n_tpl;
  std::vector<int64_t> toff(n_tpl);
  for (int64_t i = 0; i < n_tpl; ++i) { toff[i] = pos; pos += (int64_t)tw[i].size(); }
  const int64_t slab_off = m.slabs.empty() ? 0 : pos;
  if (!m.slabs.empty()) pos += 1 + IEM_SLAB_WORDS * (int64_t)m.slabs.size();
  std::vector<int64_t> aoff(n_arr, 0);
  for (int64_t i = 0; i < n_arr; ++i) {
    const ArrayDesc &a = m.arrs[order[i]];
    if (a.kind == IEM_A_F64_DATA || a.kind == IEM_A_I64_DATA) { aoff[i] = pos; pos += a.n; }
  }
  std::vector<int64_t> out((size_t)pos);
  out[0] = IEM_BLOB_MAGIC; out[1] = IEM_BLOB_VERSION; out[2] = m.nvar; out[3] = m.npar; out[4] = m.ncon; out[5] = n_tpl; out[6] = n_arr;
  out[7] = m.minimize; out[8] = pos; out[9] = slab_off;
  out[10] = remap[m.arr_x0]; out[11] = remap[m.arr_lvar]; out[12] = remap[m.arr_uvar]; out[13] = remap[m.arr_theta];
  int64_t *aw = out.data() + IEM_HDR_WORDS;
  for (int64_t i = 0; i < n_arr; ++i, aw += IEM_ARR_WORDS) {
    const ArrayDesc &a = m.arrs[order[i]];
    aw[0] = a.kind; aw[1] = a.n; aw[2] = aoff[i];
    aw[3] = a.kind == IEM_A_F64_FILL ? d2w(a.fill) : a.r0; aw[4] = a.rstep; aw[5] = 0;
    if (a.kind == IEM_A_F64_DATA || a.kind == IEM_A_I64_DATA) std::memcpy(out.data() + aoff[i], a.data, (size_t)a.n * 8);
  }
  int64_t *tp = out.data() + IEM_HDR_WORDS + IEM_ARR_WORDS * n_arr;
  for (int64_t i = 0; i < n_tpl; ++i) { tp[i] = toff[i]; std::memcpy(out.data() + toff[i], tw[i].data(), tw[i].size() * 8); }
  if (!m.slabs.empty()) {
    int64_t *sw = out.data() + slab_off;
    *sw++ = (int64_t)m.slabs.size();
    for (const Slab &s : m.slabs) {
      *sw++ = s.off; *sw++ = s.nd;
      for (int d = 0; d < 3; ++d) *sw++ = s.dims[d];
      for (int d = 0; d < 3; ++d) *sw++ = s.group[d];
    }
  }
  return out;
}

}  // namespace iem
