// iem_kkt_host.hpp — host side of the chain KKT solver behind the C-ABI (iem_kkt_create / _assemble / _factor / _solve):
// the grouping of the augmented system's unknowns into chain blocks + border, the narrow coupling, and the gather plan
// that fills the blocks straight from the hess_coord! / jac_coord! value arrays.
//
// The same analysis exists in Python (infiniteexamodels.jl_amd/kkt_chain.py: ChainLayout — where it was developed, and what
// tests/test_kkt_cabi.py compares this file with, field for field): unknown u < nvar is variable u, u >= nvar the
// multiplier of row u - nvar; a support of the chain's parameter group owns its variables and the rows whose LAST touched
// support it is; blocks are `reach` supports long and start where that gives the narrowest coupling; inside a block every
// unknown sits at a FIXED place (variables before rows, by the support's position in the block, then by index; a short
// block leaves holes, filled with a unit diagonal); unknowns off the chain form the border.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <cmath>
#include <set>
#include <stdexcept>
#include <tuple>
#include <string>
#include <vector>

#include "iem_model.hpp"

namespace iem {

struct KktLayout {
  int64_t nvar = 0, ncon = 0, S = 0, n_border = 0;
  int nb = 0, ne = 0, nc = 4, reach = 0, group = 0, phase = 0;
  // LANES (2-D support grids, e.g. ESCAPE34/pandemic.jl: t x xi): when the blocks of one chain support would be too large
  // (17 N_xi + 1 unknowns per time support), every point of the OTHER parameter(s) gets a chain of its own — block index =
  // lane * (blocks per lane) + time block, zero coupling at the seams — and the unknowns that live on the chain's parameter
  // alone (u(t): one per time support, coupled to every lane) form the border.  1: plain chain.
  int64_t lanes = 1;
  std::vector<int64_t> blk, loc;        // per unknown: block (-1: border) and place in the block / the border
  std::vector<int64_t> counts;          // real unknowns per block
  std::vector<int32_t> rowsR, colsC;    // local rows (of block k) / columns (of block k - 1) of the coupling, padded with -1 to nc
  // flat buffer D | Bt | E | G
  int64_t oD() const { return 0; }
  int64_t oB() const { return S * (int64_t)nb * nb; }
  int64_t oE() const { return oB() + S * (int64_t)nc * nc; }
  int64_t oG() const { return oE() + e_doubles(); }
  int64_t total() const { return oG() + g_doubles(); }
  // HUBS (BASELINE config 3, pandemic 5 000 x 100): a laned grid whose border is too large for the dense-border kernels — the
  // border unknowns u(t) become hubs, each owned by one time block; the blocks keep SPAN-SPARSE border columns between the
  // reduction levels (csrc/iem_kkt_device.h: kkt_hub_z / kkt_hub_widen) and the hubs' Schur complement is a dense Hp x Hp matrix.
  // The kernels see ne = 0.  E: [Tp][lanes][nq][hw] on the nq local rows Q that ever hold a border entry; G: [Hp][Hp].
  bool hubs = false;
  int64_t Tp = 0, hw = 0, H = 0, Hp = 0;   // blocks per lane, hubs per time block, hubs (time blocks that own one x hw), row length of G
  int nq = 0, nr = 0, ncq = 0;
  std::vector<int64_t> hub_of;            // per unknown: hub index (time block * hw + ordinal), -1 for the chain's unknowns
  std::vector<int32_t> Q, qR, qC;         // local rows of Q; positions of the coupling rows / columns inside Q
  int64_t e_doubles() const { return hubs ? Tp * lanes * (int64_t)nq * hw : S * (int64_t)nb * ne; }
  int64_t g_doubles() const { return hubs ? Hp * Hp : (int64_t)ne * ne; }
};

inline int64_t ceil4(int64_t n) { return (n + 3) / 4 * 4; }

// gather plan: flat[dest[i]] = sum over k in [seg[i], seg[i+1]) of source(perm[k]); sources are indices into the virtual array
// hess values | jac values | (sigma + delta_w) per variable | -delta_c per row | 1.0 (the padding's unit diagonal)
struct KktPlan {
  std::vector<int64_t> dest;
  std::vector<uint32_t> seg, perm;
  int64_t n_h = 0, n_j = 0;
};

// (hr, hc: the Hessian's structure — an entry across two supports would widen the coupling; none of the reference's models has one)
inline KktLayout kkt_layout(const Model &m, const std::vector<int64_t> &jr, const std::vector<int64_t> &jc, const std::vector<int64_t> &hr,
                            const std::vector<int64_t> &hc, int want_group, int max_nb, int max_ne, int max_nc, bool allow_hubs = false) {
  KktLayout L;
  L.nvar = m.nvar; L.ncon = m.ncon;
  const int64_t nvar = m.nvar, ncon = m.ncon, n = nvar + ncon, nj = (int64_t)jr.size();
  std::set<int> groups;
  for (const Slab &s : m.slabs) for (int a = 0; a < s.nd; ++a) if (s.group[a] > 0) groups.insert(s.group[a]);
  if (groups.empty()) throw std::runtime_error("chain KKT: the model has no infinite-parameter slab table (nothing to chain along)");
  auto coords = [&](int g, std::vector<int64_t> &vc) {
    vc.assign((size_t)nvar, -1);
    for (const Slab &s : m.slabs)
      for (int a = 0; a < s.nd; ++a) {
        if (s.group[a] != g) continue;
        int64_t stride = 1;
        for (int d = 0; d < a; ++d) stride *= s.dims[d];
        const int64_t len = s.length();
        for (int64_t i = 0; i < len; ++i) vc[(size_t)(s.off + i)] = (i / stride) % s.dims[a];
        break;
      }
  };
  auto row_span = [&](const std::vector<int64_t> &vc, std::vector<int64_t> &hi, std::vector<int64_t> &lo) {
    hi.assign((size_t)ncon, -1); lo.assign((size_t)ncon, INT64_MAX);
    for (int64_t k = 0; k < nj; ++k) {
      const int64_t v = vc[(size_t)jc[k]];
      hi[(size_t)jr[k]] = std::max(hi[(size_t)jr[k]], v);
      if (v >= 0) lo[(size_t)jr[k]] = std::min(lo[(size_t)jr[k]], v);
    }
    for (int64_t r = 0; r < ncon; ++r) if (lo[(size_t)r] == INT64_MAX) lo[(size_t)r] = hi[(size_t)r];
  };
  std::vector<int64_t> vc, hi, lo;
  int group = want_group;
  if (group <= 0) {
    bool have = false;
    std::pair<bool, int64_t> best_key;
    for (int g : groups) {
      std::vector<int64_t> v, h, l;
      coords(g, v); row_span(v, h, l);
      int64_t reach = 0, top = -1;
      for (int64_t r = 0; r < ncon; ++r) reach = std::max(reach, h[(size_t)r] - l[(size_t)r]);
      for (int64_t x : v) top = std::max(top, x);
      const std::pair<bool, int64_t> key{reach > 0, top + 1};
      if (!have || key > best_key) { have = true; best_key = key; group = g; vc.swap(v); hi.swap(h); lo.swap(l); }
    }
  } else {
    coords(group, vc); row_span(vc, hi, lo);
  }
  L.group = group;
  // lane of a variable: its flattened position along the OTHER dimensions of a slab that carries the chain's group (-1: the
  // variable has no other dimension — or no chain coordinate at all)
  std::vector<int64_t> vlane((size_t)nvar, -1);
  int64_t nlanes = 1;
  bool lanes_consistent = true;
  for (const Slab &s : m.slabs)
    for (int a = 0; a < s.nd; ++a) {
      if (s.group[a] != group) continue;
      int64_t stride = 1, other = 1;
      for (int d = 0; d < a; ++d) stride *= s.dims[d];
      for (int d = 0; d < s.nd; ++d) if (d != a) other *= s.dims[d];
      if (other > 1) {
        if (nlanes > 1 && nlanes != other) lanes_consistent = false;
        nlanes = std::max(nlanes, other);
        const int64_t len = s.length(), span = stride * s.dims[a];
        for (int64_t i = 0; i < len; ++i) vlane[(size_t)(s.off + i)] = (i / span) * stride + (i % stride);
      }
      break;
    }
  struct Cand { std::vector<int64_t> blk, loc, counts; int nb = 0; std::vector<int64_t> rows, cols; int64_t S = 0; };
  // `use_lanes`: one chain per lane, lane-less chain variables to the border
  auto build = [&](bool use_lanes, bool hub_mode = false) {
    KktLayout B = L;
    std::vector<int64_t> chain((size_t)n, -1), lane((size_t)n, 0);
    std::vector<int64_t> rhi(hi), rlo(lo);
    if (use_lanes) {
      // rows: lane and span from their LANED variables only; a row whose variables sit in two lanes does not fit
      std::vector<int64_t> rl((size_t)ncon, -1);
      rhi.assign((size_t)ncon, -1); rlo.assign((size_t)ncon, INT64_MAX);
      for (int64_t k = 0; k < nj; ++k) {
        const int64_t v = jc[(size_t)k], r = jr[(size_t)k];
        if (vlane[(size_t)v] < 0 || vc[(size_t)v] < 0) continue;
        if (rl[(size_t)r] >= 0 && rl[(size_t)r] != vlane[(size_t)v]) throw std::runtime_error("chain KKT: a constraint row couples two lanes of the support grid");
        rl[(size_t)r] = vlane[(size_t)v];
        rhi[(size_t)r] = std::max(rhi[(size_t)r], vc[(size_t)v]); rlo[(size_t)r] = std::min(rlo[(size_t)r], vc[(size_t)v]);
      }
      for (int64_t i = 0; i < nvar; ++i) if (vlane[(size_t)i] >= 0 && vc[(size_t)i] >= 0) { chain[(size_t)i] = vc[(size_t)i]; lane[(size_t)i] = vlane[(size_t)i]; }
      for (int64_t r = 0; r < ncon; ++r) if (rl[(size_t)r] >= 0) { chain[(size_t)(nvar + r)] = rhi[(size_t)r]; lane[(size_t)(nvar + r)] = rl[(size_t)r]; }
      B.lanes = nlanes;
    } else {
      for (int64_t i = 0; i < nvar; ++i) chain[(size_t)i] = vc[(size_t)i];
      for (int64_t r = 0; r < ncon; ++r) chain[(size_t)(nvar + r)] = hi[(size_t)r];     // a row sits with the LAST support it touches
    }
    B.reach = 0;
    for (int64_t r = 0; r < ncon; ++r) if (rhi[(size_t)r] >= 0 && rlo[(size_t)r] != INT64_MAX) B.reach = (int)std::max<int64_t>(B.reach, rhi[(size_t)r] - rlo[(size_t)r]);
    const int64_t R = std::max(B.reach, 1);
    std::vector<int64_t> ids, border;
    for (int64_t u = 0; u < n; ++u) (chain[(size_t)u] >= 0 ? ids : border).push_back(u);
    auto arrange = [&](int64_t phase) {
      Cand c;
      c.blk.assign((size_t)n, -1); c.loc.assign((size_t)n, -1);
      std::vector<int64_t> off((size_t)n, 0);
      int64_t Sb = 0;
      for (int64_t u : ids) Sb = std::max(Sb, (chain[(size_t)u] + phase) / R + 1);
      for (int64_t u : ids) { c.blk[(size_t)u] = lane[(size_t)u] * Sb + (chain[(size_t)u] + phase) / R; off[(size_t)u] = (chain[(size_t)u] + phase) % R; }
      c.S = ids.empty() ? 0 : B.lanes * Sb;
      auto gkey = [&](int64_t u) { return (c.blk[(size_t)u] * 2 + (u >= nvar ? 1 : 0)) * R + off[(size_t)u]; };
      std::vector<int64_t> order(ids);
      std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return gkey(a) < gkey(b); });
      std::vector<int64_t> ordinal(order.size(), 0), size((size_t)(2 * R), 0);
      for (size_t i = 0; i < order.size(); ++i) {
        if (i && gkey(order[i]) == gkey(order[i - 1])) ordinal[i] = ordinal[i - 1] + 1;
        const int64_t u = order[i], ko = (u >= nvar ? 1 : 0) * R + off[(size_t)u];
        size[(size_t)ko] = std::max(size[(size_t)ko], ordinal[i] + 1);
      }
      std::vector<int64_t> base((size_t)(2 * R), 0);
      int64_t run = 0;
      for (int64_t ko = 0; ko < 2 * R; ++ko) { base[(size_t)ko] = run; run += size[(size_t)ko]; }   // variables by position, then rows by position
      for (size_t i = 0; i < order.size(); ++i) {
        const int64_t u = order[i], ko = (u >= nvar ? 1 : 0) * R + off[(size_t)u];
        c.loc[(size_t)u] = base[(size_t)ko] + ordinal[i];
      }
      for (size_t i = 0; i < border.size(); ++i) c.loc[(size_t)border[i]] = (int64_t)i;
      c.counts.assign((size_t)c.S, 0);
      for (int64_t u : ids) ++c.counts[(size_t)c.blk[(size_t)u]];
      c.nb = (int)ceil4(run);
      std::set<int64_t> rs, cs;
      for (int64_t k = 0; k < nj; ++k) {
        const int64_t ur = nvar + jr[k], uc = jc[k], kr = c.blk[(size_t)ur], kc = c.blk[(size_t)uc];
        if (kr < 0 || kc < 0) continue;
        if (kr == kc + 1) { rs.insert(c.loc[(size_t)ur]); cs.insert(c.loc[(size_t)uc]); }
        else if (kc == kr + 1) { rs.insert(c.loc[(size_t)uc]); cs.insert(c.loc[(size_t)ur]); }
      }
      for (size_t k = 0; k < hr.size(); ++k) {
        const int64_t ur = hr[k], uc = hc[k], kr = c.blk[(size_t)ur], kc = c.blk[(size_t)uc];
        if (kr < 0 || kc < 0 || kr == kc) continue;
        if (kr == kc + 1) { rs.insert(c.loc[(size_t)ur]); cs.insert(c.loc[(size_t)uc]); }
        else if (kc == kr + 1) { rs.insert(c.loc[(size_t)uc]); cs.insert(c.loc[(size_t)ur]); }
      }
      c.rows.assign(rs.begin(), rs.end()); c.cols.assign(cs.begin(), cs.end());
      return c;
    };
    bool have = false;
    std::tuple<int64_t, int, int64_t> best_key;
    Cand best;
    for (int64_t phase = 0; phase < R; ++phase) {
      Cand c = arrange(phase);
      const std::tuple<int64_t, int, int64_t> key{(int64_t)std::max(c.rows.size(), c.cols.size()), c.nb, phase};
      if (!have || key < best_key) { have = true; best_key = key; best = std::move(c); B.phase = (int)phase; }
    }
    if (best.S < 1) throw std::runtime_error("chain KKT: no unknown lies on the chain");
    B.S = best.S; B.blk.swap(best.blk); B.loc.swap(best.loc); B.counts.swap(best.counts);
    B.n_border = (int64_t)border.size();
    B.nb = best.nb; B.ne = hub_mode ? 0 : (int)ceil4((int64_t)border.size());
    if (B.nb > max_nb || (!hub_mode && B.ne > max_ne))
      throw std::runtime_error("chain KKT: blocks of " + std::to_string(*std::max_element(B.counts.begin(), B.counts.end())) + " unknowns / a border of " +
                               std::to_string(border.size()) + " exceed the dense-block solver's limits (" + std::to_string(max_nb) + " / " + std::to_string(max_ne) + ")");
    B.nc = (int)std::max<int64_t>(ceil4((int64_t)std::max(best.rows.size(), best.cols.size())), 4);
    if (B.reach > 0 && (B.nc > max_nc || B.nc > B.nb))
      throw std::runtime_error("chain KKT: the coupling between neighbouring blocks spans " + std::to_string(best.rows.size()) + " rows / " +
                               std::to_string(best.cols.size()) + " columns (limit " + std::to_string(max_nc) + ")");
    B.rowsR.assign((size_t)B.nc, -1); B.colsC.assign((size_t)B.nc, -1);
    for (size_t i = 0; i < best.rows.size(); ++i) B.rowsR[i] = (int32_t)best.rows[i];
    for (size_t i = 0; i < best.cols.size(); ++i) B.colsC[i] = (int32_t)best.cols[i];
    if (hub_mode) {
      if (B.reach != 1) throw std::runtime_error("chain KKT (hub border): only stencils of reach 1 (backward differences) are handled");
      B.hubs = true;
      B.Tp = B.S / B.lanes;
      // every border unknown belongs to the time block of its own support (a row: of the last support it touches)
      std::vector<int64_t> tb(border.size()), cnt((size_t)B.Tp, 0);
      B.hub_of.assign((size_t)n, -1);
      for (size_t i = 0; i < border.size(); ++i) {
        const int64_t u = border[i], c = u < nvar ? vc[(size_t)u] : hi[(size_t)(u - nvar)];
        if (c < 0) throw std::runtime_error("chain KKT (hub border): a border unknown has no place on the chain's axis (finite variables are not handled here)");
        tb[i] = c + B.phase;      // (R = 1)
        if (tb[i] >= B.Tp) throw std::runtime_error("chain KKT (hub border): a border unknown lies beyond the chain's last block");
      }
      int64_t tmax = -1;
      for (size_t i = 0; i < border.size(); ++i) { B.hub_of[(size_t)border[i]] = cnt[(size_t)tb[i]]++; tmax = std::max(tmax, tb[i]); }   // ordinal first ...
      B.hw = 0;
      for (int64_t c : cnt) B.hw = std::max(B.hw, c);
      for (size_t i = 0; i < border.size(); ++i) B.hub_of[(size_t)border[i]] += tb[i] * B.hw;                                          // ... then the time block's base
      B.H = (tmax + 1) * B.hw;
      int64_t P = 1;
      while (P < B.Tp) P *= 2;
      B.Hp = P * B.hw;          // the spans of the last levels reach up to the next power of two of time blocks
      // Q: the local rows a border entry sits on (J, J', H across chain / border), and the coupling rows / columns
      std::set<int32_t> q;
      auto see = [&](int64_t ua, int64_t ub) {
        if (B.blk[(size_t)ua] >= 0 && B.blk[(size_t)ub] < 0) {
          if (B.hub_of[(size_t)ub] / B.hw != B.blk[(size_t)ua] % B.Tp) throw std::runtime_error("chain KKT (hub border): a border unknown couples to a block of another time support");
          q.insert((int32_t)B.loc[(size_t)ua]);
        }
      };
      for (int64_t k = 0; k < nj; ++k) { see(nvar + jr[(size_t)k], jc[(size_t)k]); see(jc[(size_t)k], nvar + jr[(size_t)k]); }
      for (size_t k = 0; k < hr.size(); ++k) { see(hr[k], hc[k]); see(hc[k], hr[k]); }
      for (int32_t r : B.rowsR) if (r >= 0) q.insert(r);
      for (int32_t c : B.colsC) if (c >= 0) q.insert(c);
      B.Q.assign(q.begin(), q.end());
      B.nq = (int)B.Q.size();
      if (B.nq > 24) throw std::runtime_error("chain KKT (hub border): " + std::to_string(B.nq) + " local rows hold border entries (limit 24)");
      std::vector<int32_t> qidx((size_t)B.nb, -1);
      for (int i = 0; i < B.nq; ++i) qidx[(size_t)B.Q[(size_t)i]] = i;
      for (int32_t r : B.rowsR) if (r >= 0) B.qR.push_back(qidx[(size_t)r]);
      for (int32_t c : B.colsC) if (c >= 0) B.qC.push_back(qidx[(size_t)c]);
      B.nr = (int)B.qR.size(); B.ncq = (int)B.qC.size();
    }
    return B;
  };
  // the plain chain first (every model it fits keeps its layout); a 2-D grid whose time blocks are too large gets lanes
  try {
    return build(false);
  } catch (const std::runtime_error &plain) {
    if (nlanes <= 1 || !lanes_consistent) throw;
    try {
      return build(true);
    } catch (const std::runtime_error &laned) {
      if (allow_hubs) {
        try {
          return build(true, true);
        } catch (const std::runtime_error &hub) {
          throw std::runtime_error(std::string(plain.what()) + "; one chain per lane of the support grid (" + std::to_string(nlanes) + " lanes): " + laned.what() +
                                   "; with the border as hubs: " + hub.what());
        }
      }
      throw std::runtime_error(std::string(plain.what()) + "; one chain per lane of the support grid (" + std::to_string(nlanes) + " lanes): " + laned.what());
    }
  }
}

// where the entry (ur, uc) of K goes in the flat buffer (-1: dropped — the upper coupling blocks and the border's row block, by symmetry)
inline int64_t kkt_dest(const KktLayout &L, const std::vector<int32_t> &ridx, const std::vector<int32_t> &cidx, int64_t ur, int64_t uc) {
  const int64_t kr = L.blk[(size_t)ur], kc = L.blk[(size_t)uc], lr = L.loc[(size_t)ur], lc = L.loc[(size_t)uc];
  if (kr >= 0 && kc >= 0) {
    if (kr == kc) return L.oD() + (kr * L.nb + lr) * L.nb + lc;
    if (kr == kc + 1) {
      const int32_t ri = ridx[(size_t)lr], ci = cidx[(size_t)lc];
      if (ri < 0 || ci < 0) throw std::runtime_error("chain KKT: an entry of the Hessian couples neighbouring blocks outside the rows / columns the Jacobian couples them on");
      return L.oB() + (kr * L.nc + ri) * L.nc + ci;
    }
    if (kc == kr + 1) return -1;
    throw std::runtime_error("chain KKT: an entry couples blocks that are not neighbours (the chain grouping does not fit this model)");
  }
  if (L.hubs) {
    if (kr >= 0 && kc < 0) {      // E0[time block][lane][row in Q][hub of the time block]
      int q = -1;
      for (int i = 0; i < L.nq; ++i) if (L.Q[(size_t)i] == lr) q = i;
      const int64_t hub = L.hub_of[(size_t)uc];
      return L.oE() + (((kr % L.Tp) * L.lanes + kr / L.Tp) * L.nq + q) * L.hw + hub % L.hw;
    }
    if (kr < 0 && kc < 0) return L.oG() + L.hub_of[(size_t)ur] * L.Hp + L.hub_of[(size_t)uc];
    return -1;
  }
  if (kr >= 0 && kc < 0) return L.oE() + (kr * L.nb + lr) * L.ne + lc;
  if (kr < 0 && kc < 0) return L.oG() + lr * L.ne + lc;
  return -1;
}

inline KktPlan kkt_plan(const KktLayout &L, const std::vector<int64_t> &hr, const std::vector<int64_t> &hc, const std::vector<int64_t> &jr,
                        const std::vector<int64_t> &jc) {
  KktPlan P;
  const int64_t nvar = L.nvar, ncon = L.ncon, nh = (int64_t)hr.size(), nj = (int64_t)jr.size();
  P.n_h = nh; P.n_j = nj;
  if (nh + nj + nvar + ncon + 1 >= (int64_t)UINT32_MAX) throw std::runtime_error("chain KKT: too many entries for the 32-bit gather plan");
  std::vector<int32_t> ridx((size_t)L.nb, -1), cidx((size_t)L.nb, -1);
  for (int i = 0; i < L.nc; ++i) { if (L.rowsR[(size_t)i] >= 0) ridx[(size_t)L.rowsR[(size_t)i]] = i; if (L.colsC[(size_t)i] >= 0) cidx[(size_t)L.colsC[(size_t)i]] = i; }
  std::vector<std::pair<int64_t, uint32_t>> pairs;
  pairs.reserve((size_t)(2 * nh + 2 * nj + nvar + ncon));
  auto put = [&](int64_t ur, int64_t uc, int64_t src) {
    const int64_t d = kkt_dest(L, ridx, cidx, ur, uc);
    if (d >= 0) pairs.emplace_back(d, (uint32_t)src);
  };
  for (int64_t k = 0; k < nh; ++k) {
    put(hr[(size_t)k], hc[(size_t)k], k);
    if (hr[(size_t)k] != hc[(size_t)k]) put(hc[(size_t)k], hr[(size_t)k], k);
  }
  for (int64_t k = 0; k < nj; ++k) { put(nvar + jr[(size_t)k], jc[(size_t)k], nh + k); put(jc[(size_t)k], nvar + jr[(size_t)k], nh + k); }
  for (int64_t i = 0; i < nvar; ++i) put(i, i, nh + nj + i);
  for (int64_t r = 0; r < ncon; ++r) put(nvar + r, nvar + r, nh + nj + nvar + r);
  // the padding's unit diagonal: the places of a block no unknown occupies, the border's tail
  const uint32_t one = (uint32_t)(nh + nj + nvar + ncon);
  {
    std::vector<char> used((size_t)(L.S * L.nb), 0);
    for (int64_t u = 0; u < nvar + ncon; ++u) if (L.blk[(size_t)u] >= 0) used[(size_t)(L.blk[(size_t)u] * L.nb + L.loc[(size_t)u])] = 1;
    for (int64_t s = 0; s < L.S * L.nb; ++s) if (!used[(size_t)s]) { const int64_t k = s / L.nb, l = s % L.nb; pairs.emplace_back(L.oD() + (k * L.nb + l) * L.nb + l, one); }
    for (int64_t l = L.n_border; l < L.ne; ++l) pairs.emplace_back(L.oG() + l * L.ne + l, one);
    if (L.hubs) {               // hubs nobody owns (time blocks with fewer than hw of them, the tail up to Hp)
      std::vector<char> owned((size_t)L.Hp, 0);
      for (int64_t h : L.hub_of) if (h >= 0) owned[(size_t)h] = 1;
      for (int64_t h = 0; h < L.Hp; ++h) if (!owned[(size_t)h]) pairs.emplace_back(L.oG() + h * L.Hp + h, one);
    }
  }
  // by destination, sources of one destination in the order above: the regions D | Bt | E are block-major, so a counting
  // sort over (region, block) followed by a sort inside each bucket (a few hundred entries) is the global order
  {
    const int64_t S = L.S, wD = (int64_t)L.nb * L.nb, wB = (int64_t)L.nc * L.nc, wE = std::max<int64_t>(L.hubs ? (int64_t)L.nq * L.hw : (int64_t)L.nb * L.ne, 1);
    auto bucket = [&](int64_t d) -> int64_t {
      if (d < L.oB()) return (d - L.oD()) / wD;
      if (d < L.oE()) return S + (d - L.oB()) / wB;
      if (d < L.oG()) return 2 * S + (d - L.oE()) / wE;
      return 3 * S;
    };
    std::vector<int64_t> start((size_t)(3 * S + 2), 0);
    for (const auto &pr : pairs) ++start[(size_t)(bucket(pr.first) + 1)];
    for (size_t b = 1; b < start.size(); ++b) start[b] += start[b - 1];
    std::vector<std::pair<int64_t, uint32_t>> sorted(pairs.size());
    {
      std::vector<int64_t> at(start.begin(), start.end() - 1);
      for (const auto &pr : pairs) sorted[(size_t)at[(size_t)bucket(pr.first)]++] = pr;
    }
    for (int64_t b = 0; b + 1 < (int64_t)start.size(); ++b)
      std::stable_sort(sorted.begin() + start[(size_t)b], sorted.begin() + start[(size_t)(b + 1)],
                       [](const std::pair<int64_t, uint32_t> &a, const std::pair<int64_t, uint32_t> &c) { return a.first < c.first; });
    pairs.swap(sorted);
  }
  P.perm.reserve(pairs.size());
  for (size_t k = 0; k < pairs.size(); ++k) {
    if (k == 0 || pairs[k].first != pairs[k - 1].first) { P.dest.push_back(pairs[k].first); P.seg.push_back((uint32_t)k); }
    P.perm.push_back(pairs[k].second);
  }
  P.seg.push_back((uint32_t)pairs.size());
  return P;
}

// eigenvalues of a small symmetric matrix (cyclic Jacobi): the inertia of the border's Schur complement
inline void sym_eigenvalues(std::vector<double> a, int n, std::vector<double> &ev) {
  for (int sweep = 0; sweep < 60; ++sweep) {
    double offd = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += a[(size_t)(i * n + i)] * a[(size_t)(i * n + i)];
      for (int j = i + 1; j < n; ++j) offd += a[(size_t)(i * n + j)] * a[(size_t)(i * n + j)];
    }
    if (offd <= 1e-30 * diag || offd < 1e-300) break;      // (relative: a converged sweep leaves rounding, not zero)
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)(p * n + q)];
        if (apq == 0.0) continue;
        const double th = (a[(size_t)(q * n + q)] - a[(size_t)(p * n + p)]) / (2.0 * apq);
        const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0)), c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) { const double akp = a[(size_t)(k * n + p)], akq = a[(size_t)(k * n + q)]; a[(size_t)(k * n + p)] = c * akp - s * akq; a[(size_t)(k * n + q)] = s * akp + c * akq; }
        for (int k = 0; k < n; ++k) { const double apk = a[(size_t)(p * n + k)], aqk = a[(size_t)(q * n + k)]; a[(size_t)(p * n + k)] = c * apk - s * aqk; a[(size_t)(q * n + k)] = s * apk + c * aqk; }
      }
  }
  ev.resize((size_t)n);
  for (int i = 0; i < n; ++i) ev[(size_t)i] = a[(size_t)(i * n + i)];
}

// inertia of a symmetric matrix by LDL' with symmetric (diagonal) pivoting — O(n^3 / 3), for borders beyond what the Jacobi sweeps
// above handle cheaply.  A pivot below `rel` times the largest diagonal seen is DOUBTFUL (counted, then skipped).
inline void sym_inertia_ldl(std::vector<double> a, int n, int64_t &neg, int64_t &doubtful, double rel = 1e-14) {
  neg = doubtful = 0;
  std::vector<char> done((size_t)n, 0);
  double scale = 0.0;
  for (int i = 0; i < n; ++i) scale = std::max(scale, std::fabs(a[(size_t)(i * n + i)]));
  for (int step = 0; step < n; ++step) {
    int p = -1;
    for (int i = 0; i < n; ++i) if (!done[(size_t)i] && (p < 0 || std::fabs(a[(size_t)(i * n + i)]) > std::fabs(a[(size_t)(p * n + p)]))) p = i;
    const double d = a[(size_t)(p * n + p)];
    done[(size_t)p] = 1;
    if (std::fabs(d) <= rel * scale || d == 0.0) {
      // (a zero diagonal with off-diagonal mass would need a 2 x 2 pivot: one positive and one negative eigenvalue — reported as
      // doubtful here, which makes the caller shift; the regularised systems this is used on are quasi-definite)
      ++doubtful;
      continue;
    }
    if (d < 0.0) ++neg;
    for (int i = 0; i < n; ++i) {
      if (done[(size_t)i]) continue;
      const double f = a[(size_t)(i * n + p)] / d;
      if (f == 0.0) continue;
      for (int j = 0; j < n; ++j) if (!done[(size_t)j]) a[(size_t)(i * n + j)] -= f * a[(size_t)(p * n + j)];
    }
  }
}

// A x = b for a small dense matrix, partial pivoting (in place on copies); false when singular
inline bool dense_solve(std::vector<double> a, int n, std::vector<double> &b) {
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int i = k + 1; i < n; ++i) if (std::fabs(a[(size_t)(i * n + k)]) > std::fabs(a[(size_t)(p * n + k)])) p = i;
    if (a[(size_t)(p * n + k)] == 0.0) return false;
    if (p != k) { for (int j = 0; j < n; ++j) std::swap(a[(size_t)(p * n + j)], a[(size_t)(k * n + j)]); std::swap(b[(size_t)p], b[(size_t)k]); }
    for (int i = k + 1; i < n; ++i) {
      const double f = a[(size_t)(i * n + k)] / a[(size_t)(k * n + k)];
      if (f == 0.0) continue;
      for (int j = k; j < n; ++j) a[(size_t)(i * n + j)] -= f * a[(size_t)(k * n + j)];
      b[(size_t)i] -= f * b[(size_t)k];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[(size_t)i];
    for (int j = i + 1; j < n; ++j) s -= a[(size_t)(i * n + j)] * b[(size_t)j];
    b[(size_t)i] = s / a[(size_t)(i * n + i)];
  }
  return true;
}

}  // namespace iem
