// iem_codegen.hpp — model → fused HIP kernels (source + launch descriptors).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "iem_model.hpp"

namespace iem {

enum KernelKind { KK_CONS = 0, KK_JAC = 1, KK_HESS = 2, KK_OBJ = 3, KK_GRAD = 4, KK_JPROD = 5, KK_JTPROD = 6, KK_HPROD = 7, KK_COUNT = 8,
                  // jac_coord! + hess_coord! in ONE launch (iem_jac_hess_coord): the bodies of both kinds behind one workgroup-id
                  // dispatcher; `out` = the Jacobian values, `aux` = the Hessian values.  Not a per-kind table index (KK_COUNT stays 8).
                  KK_PAIR = 8,
                  // one launch per solver phase (iem_eval_trial: obj + cons!; iem_eval_accepted: grad! + jac_coord! + hess_coord!): the
                  // member kinds' bodies behind one workgroup-id dispatcher.  Pointers: trial out = c, aux = objective scalar, p2 =
                  // partials; accepted out = jac values, aux = hess values, p2 = g, p3 = grad!'s reduction buffer
                  KK_TRIAL = 9, KK_ACCEPTED = 10,
                  // ... and all five evaluations of one point in one launch (iem_eval_all): p4 = c, p5 = the objective's partials, p6 = its scalar
                  KK_ALL = 11, KK_LAST = 11 };

struct KernelDesc {
  std::string name;
  int kind = 0;
  int64_t grid[3] = {1, 1, 1};  // workgroups
  // argument block, in this order after the fixed head {x, th, y, v, out, w, aux}:
  std::vector<int64_t> ip;  // long long ip[]
  std::vector<double> dp;   // double dp[]
  std::vector<int> fa;      // const double* fa[]  (model array ids, uploaded as f64)
  std::vector<int> ia;      // const long long* ia[] (model array ids, uploaded as i64)
  int lds_bytes = 0;
  bool tables_in_memory = false;  // ip/dp/fa/ia passed as pointers to device tables (argument block too large)
  int block = 256;          // threads per workgroup of this kernel
  int64_t partial_off = 0;  // KK_OBJ: first partial slot written by this kernel
  int64_t n_blocks = 1;
  // bookkeeping for the roofline line
  int64_t alg_bytes_read = 0, alg_bytes_written = 0;
  // 0-based index ranges [lo, hi] of x (and of v, for the kinds whose v lives in variable space) the kernel's LIVE loads
  // can touch over its launch domain — a sharded handle derives from them which kinds read a halo entry, i.e. which
  // calls must wait for an asynchronous halo exchange (iem_halo_exchange_async) and which may overlap it
  std::vector<std::pair<int64_t, int64_t>> x_ranges, v_ranges;
  int lds_slots = 0;        // the staging batch this kernel was generated with (Options::lds_slots or its large-grid override)
  bool carries = false;     // the kernel has the halo-carrier prologue (Options::carrier): a launch may bring one extra leading workgroup
  int inter = -1;           // >= 0: bodies of one launch with the same value (and the same grid) have their workgroups interleaved
};

struct Options {
  int store_mode = 2;  // 0 direct strided stores, 1 wave-level LDS-transposed stores, 2 block-cooperative 128-B-aligned stores
  int block = 0;       // workgroup size of the fused kernels (multiple of 64); 0 = chosen per model: 512 (halves the partial cache
                       // lines at block seams) unless 256-lane tiles waste > 2 % fewer lanes on the rows of a 2-D/3-D grid
  int lds_slots = 24;  // store_mode 2: values per lane staged per barrier pair (LDS = 2 KB x this per workgroup)
  int reorder = 1;     // 1: emit cheap templates first so the store stream starts early
  int no_fuse = 0;     // 1: one kernel per template (the reference design's launch structure; baseline/ablation only)
  int hess_merge = 0;  // 1: opt-in merged Hessian layout (duplicate (row,col) slots of one support summed in registers)
  int ablate = 0;      // timing experiments only (WRONG results): 1 no LDS transpose, 2 no transcendentals
  int min_waves = 0;   // >0: __launch_bounds__(256, min_waves) on the fused kernels
  int nt_stores = 1;   // 1: non-temporal stores for the streamed COO outputs
  int wide_stores = 1; // 1: 16 bytes per lane in the block store (in-process A/B r04: jac_coord! -0.5 .. -1 %, hess_coord! -0.3 %; profiles/r04_ab_jac_split.txt)
  int overlap = 1;     // 1: block-store kernels overlap their tiles by 16 lanes so that every 128-byte line is written whole
  int xcd_remap = 0;   // 1: consecutive logical workgroups share an XCD (experiment: partial lines at block seams did NOT merge in its L2; -1.5 %)
  int split_small = 64; // support grids of at most this many workgroups run their templates side by side (0: never)
  int fuse_groups = 1; // 1: one launch per call even when its templates live on several support grids (workgroup-id dispatch)
  int fuse_zero = 1;   // 1: scatter kernels zero the untouched output entries themselves when nothing accumulates (no memset launch)
  int fp_contract = 0; // 0: -ffp-contract=off (bit-comparable with the oracle's arithmetic), 1: fast (FMA)
  int obj_wgs = 1024;  // obj: at most this many workgroups walk the tiles (one partial each; fixed, so the summation order is)
  int flat2d = 0;      // 1: every 2-D support grid is walked by one linear lane index (no partly filled workgroup per row; sub-box
                       // templates stored by item ordinal).  Opt-in: measured equal on jac/hess and slower on cons! for pandemic
                       // 5000 x 100 (profiles/r02_ab_flush_flat.txt) — the 256-lane tiles of Options::block = 0 already remove the waste
  int flush32 = 2;     // block-store loops: 2 = 32-bit offsets relative to the block start, per-lane predicate on every round (default);
                       // 1 = whole rounds decided by scalar compares + branch (1-5 % SLOWER: the branches cost more than the predicates
                       // they save); 0 = round-1 form with 64-bit predicates (equal to 2) — profiles/r02_ab_flush_flat.txt
  int pull_scatter = 1;   // 1: scatter kinds (grad!/jtprod!/hprod!): an addend that lands on a neighbour lane's entry (x[i-1] of a
                          // difference row) is computed by that lane through a shifted clone of the template — exclusive stores, no
                          // zero fill — instead of an atomic (0: A/B)
  int det_scatter = 1;    // 1: what would still be a float atomic AND can meet more than one other addend in its entry is parked and
                          // summed in a fixed order from a host-built plan (Program::Gather) — every kind bitwise reproducible on
                          // every model; 2: also when every entry gets at most two addends; 0: f64 atomics
  int64_t det_scatter_max = 1LL << 28;   // ... unless a kind has more addends than this (plan + scratch: 16 bytes each)
  int det_axis = 1;       // 1: sums over a non-lane axis are reduced in a fixed order by a follow-up kernel (Program::axis); 0: atomics
  int lazy_loads = 2;     // product and scatter kinds with >= lazy_min_loads loads: 1 load the rows of v / y where first used, 2 every load (0: all loads at the head)
  int lazy_min_loads = 48;
  int name_tag = 0;         // > 0: kernel names end in _b<tag> (set by the runtime for a handle's second code object)
  int lazy_all_kinds = 0;   // experiment: also cons!/jac_coord!/hess_coord!/obj
  int autotune = 0;    // opt-in (measured gains depend on the process).  1: jac_coord!/hess_coord! of large grids keep a second code object (lds_slots = 48) and pick, per output
                       // buffer, the faster of the two from their first twenty calls (runtime only; the generator ignores it)
  int autotune_min_blocks = 400;   // ... grids of at least this many workgroups (about 2e5 supports)
  int obj_unroll = 1;  // 2: the objective's tile walk takes two tiles per trip (single-body kernels)
  int det_shared = 1;  // 1: scatter entries shared by many items are reduced deterministically (iem_shared_*), 0: one f64 atomic per wave
  // LARGE grids — outputs far beyond the 256-MiB Infinity Cache, everything goes to DRAM — get another kernel shape for
  // jac_coord! / hess_coord!, chosen from the GRID SIZE per kind (never from a timer): when a kind has a grid of at least
  // `big_batch_jac` / `big_batch_hess` workgroups (counted at the model's own tile; 4000 = about 2e6 quadrotor supports; 0:
  // never), ALL kernels of that kind run `big_tile`-lane workgroups (1024: twice the contiguous chunk per store stream;
  // every other kind keeps the model's tile — cons! is 27 % slower at 1024 lanes), stage `big_batch_slots` (quoted for
  // 256 lanes: 12 values per lane at 1024, one 96-KB workgroup per CU) and (`big_xcd`) walk their tiles XCD-aware:
  // consecutive tiles on ONE XCD, so that neighbouring chunks of a stream leave through one L2.  In-process A/Bs on two
  // boxes (profiles/r03_ab_kernel_shapes.txt): 2e6 supports fused pair -3..-5 %, 4e6 -5..-8 % (0.69 -> 0.76..0.79 of
  // peak), 8e6 -4..-8 %; below ~1e6 no shape wins reliably (the buffers' placement moves the same kernel by more).
  int big_batch_slots = 48;
  int64_t big_batch_jac = 4000, big_batch_hess = 4000;
  int big_xcd = 1;
  int big_tile = 1024;     // 0: keep the model's tile (the 512-lane, 48-slot form of the first measurements)
  // Orthogonal collocation: the node x element boxes of the derivative rows ride on the lanes of the support grid itself.
  // 1 = for the scatter kinds (grad! / jtprod! / hprod!), together with the element lists of constant_over_collocation:
  // every addend is computed by the lane that owns its entry (KernelBuilder::pull_folded) — exclusive stores instead of
  // the plan-driven gather (quadrotor OC3, 5e5 public supports: jtprod! 417 -> 103 us); 2 (default) = the full boxes also
  // for every other kind: a row and the dynamics at its node share their loads (cons! -10 %, jprod! -18 %, jac_coord!
  // equal; profiles/r03_collocation_fold.json); 0 = off
  int fold_colloc = 2;
  int fold_max_n = 6;      // ... for at most this many rows per element (the clones of a derivative row grow with its square)
  int pair_kernel = 1;     // 1: also emit the fused jac_coord! + hess_coord! launch (KK_PAIR, iem_jac_hess_coord)
  int store_wait = 0;      // experiment: s_waitcnt vmcnt(0) behind every flushed batch (paces a wave's outstanding stores)
  // 1: the kernels of the carrier kinds (cons!, jac_coord!, hess_coord!, jprod!, obj, the pair) can carry a deferred halo
  // exchange as one extra leading workgroup (iem_halo_wg).  Set by iem_create_sharded; an unsharded handle's kernels have no
  // such prologue (and no LDS word for it) at all.
  int carrier = 0;
  int phase_kernels = 1;   // 1: also emit the one-launch-per-solver-phase kernels (KK_TRIAL, KK_ACCEPTED); their member kinds then
                           // always take the bodies-behind-a-dispatcher form, which the phase kernels share
  // jac_coord! of a lane-fused support grid runs as TWO bodies behind the dispatcher, their
  // workgroups interleaved so that both are resident: body a = the templates whose partials are item data or constants
  // (linear rows: difference rows h, -1, +1 of src/transform.jl:511-562, affine dynamics) — no x load, no arithmetic, a
  // fill-shaped body; body b = the rest (loads, trigonometry).  Halves the store fronts a workgroup keeps open; same bytes.
  // 1: bodies alternate workgroup by workgroup; 2: in runs of 8 workgroups (one per XCD), so every XCD sees both; 0: off
  int jac_split = 1;
  int64_t jac_split_min = 0;    // ... only grids of more workgroups than this (0: every lane-fused grid — the source stays size-independent)
  int cons_direct_2d = 1;  // cons! of a model whose largest grid is 2-D: plain coalesced stores instead of the LDS re-cut (kind_options)
  int split_shift = 0;     // jac_split's bodies: 1 = the FIRST body of an interleaved run (the data rows) takes its tiles half a grid away from the
                           // others' — its store fronts then sit in another part of the output while the computed rows' fronts pass
  int pair_inter = 0;      // the fused pair: 1 = bodies of equal grids (jac_coord!'s halves, hess_coord!) interleaved the same way; 0 (default,
                           // measured faster) = jac_coord! as ONE body, its workgroups first, hess_coord!'s behind them
  // runtime only (the generator ignores them)
  int comm_timeout_ms = 5000;   // bound of every mailbox wait (halo exchange / fold / all-reduce kernels)
};

// one block of the merged Hessian layout: `nslots` values per item, position o + nslots*k + s;
// slot s is the pair (idx_i[s], idx_j[s]) of index-expression ids of template `tpl`
struct HessClass {
  int tpl = 0;
  int64_t n_items = 0, o = 0;
  std::vector<int> idx_i, idx_j;
};

struct Program {
  std::string source;
  int block = 512;       // the workgroup size the kernels were generated for (Options::block resolved)
  uint64_t key = 0;
  std::vector<KernelDesc> kernels;
  int64_t n_partials = 0;
  std::vector<HessClass> hess_classes;  // non-empty iff Options::hess_merge
  int64_t nnzh_merged = 0;
  // 0-based index ranges [lo, hi] of g that the gradient kernels overwrite completely
  std::vector<std::pair<int64_t, int64_t>> grad_covered;      // KK_GRAD
  std::vector<std::pair<int64_t, int64_t>> covered[KK_COUNT];  // per scatter kind (grad, jtprod, hprod)
  std::vector<std::pair<int64_t, int64_t>> zero_ranges[KK_COUNT];  // [lo, hi) the runtime must memset before launching the kind (empty when fused into a kernel)
  // deterministic shared-entry reduction of a scatter kind (grad, jtprod, hprod): `red_values` per-lane
  // values are parked per workgroup of the call (`red_wgs` of them) in a buffer of red_values*red_wgs
  // doubles + ticket words (zeroed once), passed as the kind's `aux` argument
  int64_t red_values[KK_COUNT] = {}, red_wgs[KK_COUNT] = {};
  // Sums over a NON-LANE axis of a scatter kind (pandemic: column u(t) of jtprod! gets one addend per scenario):
  // the kernel parks each lane's addend at aux[off + row*n0 + lane] (row = position in dims 1, 2 of the template's
  // box) and iem_axis_sum_kernel, launched behind the kind's kernels, writes out[c + k0*lane] = sum over the rows in
  // row order — deterministic, no atomics.  `off` counts doubles from the start of the kind's aux buffer.
  struct AxisSum { int64_t c, k0, n0, rows, off; };
  std::vector<AxisSum> axis[KK_COUNT];
  // What would still be a float atomic (Options::det_scatter): every addend is parked at aux[aux_off + position] and
  // iem_gather_sum_kernel writes out[dest[i]] = sum of aux[aux_off + perm[k]], k in [seg[i], seg[i+1]) — in that order.
  struct Gather { std::vector<int64_t> dest, seg, perm; int64_t aux_off = 0, park_doubles = 0; };
  Gather gather[KK_COUNT];
  int64_t aux_doubles[KK_COUNT] = {};   // whole aux buffer of the kind: shared-entry part (values*wgs + ticket words), then the axis rows
};

Program generate(const Model &m, const Options &opt);
uint64_t fnv1a64(const std::string &s);

}  // namespace iem
