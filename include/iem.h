/* iem.h — C-ABI of libiem_hip.so: the MI355X evaluation backend for
 * InfiniteOpt → ExaModels transcriptions.
 *
 * The reference reaches its evaluator through Julia method calls on an
 * ExaModels.ExaModel built at /root/reference/src/infiniteopt_backend.jl:155-156
 * (`ExaModels.ExaCore(model, data; backend)` → `ExaModels.ExaModel(core)`); the NLP
 * solvers then call the NLPModels API on it every iteration
 * (/root/reference/ext/InfiniteExaModelsIpopt.jl:48-49,59-60,
 *  /root/reference/ext/InfiniteExaModelsMadNLP.jl:49-50,64).  Each entry point
 * below replaces one of those calls; a Julia `ccall` / Python `ctypes` shim binds
 * them one to one.
 *
 * Conventions
 *   - every function returns 0 on success, a negative IEM_E_* code otherwise; the
 *     message is in iem_last_error() (thread-local).  Nothing throws across the ABI.
 *   - `d_` arguments are DEVICE pointers (ROCArray / torch data_ptr), `h_` are HOST
 *     pointers.  The caller owns every array; outputs are fully overwritten.
 *   - evaluation calls are enqueued on the handle's stream and return immediately,
 *     except iem_obj (returns a host scalar) and the *_structure calls.
 *   - a handle is not thread-safe; use one per solver.
 */
#ifndef IEM_H
#define IEM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct iem_model iem_model;

enum {
  IEM_OK = 0,
  IEM_E_BLOB = -1,    /* malformed / unsupported blob                     */
  IEM_E_HIP = -2,     /* HIP runtime error                                */
  IEM_E_COMPILE = -3, /* kernel generation / hiprtc failure               */
  IEM_E_ARG = -4,     /* bad argument                                     */
  IEM_E_NODEVICE = -5,/* no usable gfx950 device                          */
  IEM_E_COMM = -6     /* a mailbox wait (halo exchange / fold / all-reduce) timed out since the last check: a peer did not take
                         part.  The kernels have set what they should have delivered to NaN; reported once, by the next host
                         synchronisation point (iem_obj, iem_obj_end, iem_synchronize), then cleared                         */
};

/* mirrors NLPModels' `meta` fields the reference reads
 * (infiniteopt_backend.jl:600-601 get_x0/get_y0; ext/InfiniteExaModels{Ipopt,MadNLP}.jl solver construction) */
typedef struct iem_meta_t {
  int64_t nvar, ncon, npar;
  int64_t nnzj, nnzh;
  int64_t n_templates;
  int32_t minimize;
  int32_t n_kernels; /* fused kernels generated for this model */
} iem_meta_t;

/* which host mirror iem_get_host returns */
enum { IEM_X0 = 0, IEM_LVAR = 1, IEM_UVAR = 2, IEM_LCON = 3, IEM_UCON = 4, IEM_Y0 = 5, IEM_THETA = 6 };

/* per-template layout (ExaModels SIMDFunction offsets o0/o1/o2 and steps) */
typedef struct iem_template_info_t {
  int64_t kind; /* 0 objective, 1 constraint */
  int64_t n_items, o0, o1, o2, o1step, o2step;
} iem_template_info_t;

/* one fused kernel of the model: launch shape and ALGORITHMIC traffic (every distinct
 * input element read once, every output element written once) — what the roofline
 * line of bench.py is computed from */
typedef struct iem_kernel_info_t {
  char name[64];
  int32_t kind; /* 0 cons, 1 jac, 2 hess, 3 obj, 4 grad, 5 jprod, 6 jtprod, 7 hprod, 8 jac + hess in one launch (iem_jac_hess_coord) */
  int32_t jit;  /* 1 if this model's code object was compiled by hiprtc (cache miss) */
  int64_t grid[3];
  int64_t lds_bytes;
  int64_t alg_bytes_read, alg_bytes_written;
} iem_kernel_info_t;

/* ---- lifecycle -----------------------------------------------------------
 * iem_create   replaces ExaModels.ExaModel(core) with backend = MI355XBackend()
 *              (infiniteopt_backend.jl:155-156; README.md:41 `backend = CUDABackend()`).
 *              `blob` is the transcribed model (include/iem_blob.h). */
int iem_create(const void *blob, size_t nbytes, int device, iem_model **out);
/* The same with per-HANDLE generator options: the process-wide defaults (iem_set_option) with
 * `opts[0 .. n_opts)` applied on top, for this model only — two models with different layouts
 * (e.g. a merged-Hessian one next to a default one) coexist in one process.  Option names as for
 * iem_set_option. */
typedef struct iem_option_t {
  const char *name;
  int64_t value;
} iem_option_t;
int iem_create_opts(const void *blob, size_t nbytes, int device, const iem_option_t *opts, int n_opts, iem_model **out);

/* ---- multi-GPU: one process (or one handle) per GPU ------------------------------------------
 * The reference is single-device; its templates are embarrassingly parallel over the supports of
 * an infinite parameter, so the path shards by support: rank r of `world` owns a
 * contiguous block of the supports of parameter group `group` (the 1-based group id the blob's grid
 * hints and slab table use: 1 = the first infinite parameter).  iem_create_sharded takes the GLOBAL
 * blob — exactly what iem_create takes — and cuts the rank's window in C++ (csrc/iem_shard.hpp): local
 * x = window slices of the sharded slabs (with the stencil's halo in front: 1 support for backward
 * differences, the element's first support for orthogonal collocation) + replicated slabs; templates over
 * the group are cut to the owned supports (whole elements for collocation boxes, filtered lists for domain
 * restrictions); cons!/jac_coord!/hess_coord! then need no communication at all.  Two exchanges remain:
 *   iem_halo_exchange        before cons!/jac!/hess!: the stencil neighbours  x_k[a_r - 1]
 *                            (transform.jl:535-557) from the left rank into the halo entries of x
 *   iem_allreduce_obj_grad   after obj / grad!: the scalar objective and the gradient entries of
 *                            replicated variables, summed over ranks in rank order (bitwise equal
 *                            on every rank)
 *   (iem_halo_fold           the transpose of the first, for J'v / Hv on a rank's rows)
 * Each is one small kernel that writes straight into the peers' mailboxes (HIP IPC; xGMI between
 * GPUs) and waits on its own — asynchronous on the handle's stream, graph-replayable, every wait
 * bounded (iem_comm_status reports a time-out).  Wiring: every rank calls iem_comm_export, the host
 * all-gathers the IEM_COMM_HANDLE_BYTES-byte handles (MPI / torch.distributed / a pipe — like an
 * ncclUniqueId), every rank calls iem_comm_connect with all of them in rank order. */
typedef struct iem_shard_t {
  int32_t group, rank, world;
  int32_t mailbox_kind;        /* 0 no mailbox yet; 1 uncached, 3 fine-grained device memory (coherent across GPUs); 2 plain hipMalloc (one-GPU rehearsal only) */
  int64_t n_global;            /* supports of the sharded group                       */
  int64_t own_lo, own_n, halo; /* first owned support (global, 0-based), count, halo supports in front */
  int64_t halo_reach, halo_doubles;
  int64_t nvar_global, ncon_global, nnzj_global, nnzh_global;
  int64_t nvar, ncon, nnzj, nnzh, n_templates; /* this rank's shard */
  int64_t n_shared;            /* replicated variables = gradient entries the all-reduce sums */
} iem_shard_t;
/* a local template and where its items sit in the global model: local item coordinate k_d = global
 * coordinate klo[d] + k_d of the box global_dims; global COO position of local slot s of item k:
 * global_o1 + o1step * (global ordinal of k) + s  (likewise o2; rows: global_o0 + ordinal) */
typedef struct iem_shard_template_t {
  int64_t global_index, kind, n_items;
  int64_t klo[3], dims[3], global_dims[3];
  int64_t o0, o1, o2, global_o0, global_o1, global_o2;
  int64_t o1step, o2step;
  /* -1: the local items are the box klo .. klo + dims of global_dims; >= 0: the template is an explicit item
   * list (a domain restriction filtered its iterator, transform.jl:448-451) and local item j is global item
   * items[items_offset + j] of the array iem_shard_template_items / iem_shard_blob hand out */
  int64_t items_offset;
} iem_shard_template_t;
#define IEM_COMM_HANDLE_BYTES 128
int iem_create_sharded(const void *blob, size_t nbytes, int device, int group, int rank, int world,
                       const iem_option_t *opts, int n_opts, iem_model **out);
int iem_shard_info(const iem_model *m, iem_shard_t *out);
/* local variable -> global variable (0-based), and per local variable: bit 0 owned by this rank,
 * bit 1 replicated on every rank, bit 2 halo copy of the left neighbour's variable (either may be NULL) */
int iem_shard_var_map(const iem_model *m, int64_t *h_map, uint8_t *h_flag);
int iem_shard_template_info(const iem_model *m, int64_t i, iem_shard_template_t *out);
/* global item ordinals of every explicit-list template, concatenated (see items_offset); h_items may be NULL to
 * query the length */
int iem_shard_template_items(const iem_model *m, int64_t *h_items, int64_t *out_n);
/* the cut without a device (tooling / tests): the rank's shard re-serialised as a blob of its own, plus
 * the maps; every out array is malloc'ed (iem_free), any of the last five may be NULL */
int iem_shard_blob(const void *blob, size_t nbytes, int group, int rank, int world, void **out_blob, size_t *out_nbytes,
                   iem_shard_t *out_info, int64_t **out_var_map, uint8_t **out_var_flag, iem_shard_template_t **out_tpl,
                   int64_t **out_items);
int iem_comm_export(iem_model *m, void *out_handle /* IEM_COMM_HANDLE_BYTES */);
int iem_comm_connect(iem_model *m, const void *all_handles /* world x IEM_COMM_HANDLE_BYTES, rank order */);
int iem_halo_exchange(iem_model *m, double *d_x);
/* The same exchange OFF the critical path — nothing is launched for it.  The call only DEFERS the exchange; it then rides on
 * the first evaluation launch that takes the same d_x and whose kernels cannot touch a halo entry of it: ONE EXTRA LEADING
 * WORKGROUP of that kernel sends my boundary supports to the right neighbour, waits (bounded) for the left neighbour's,
 * writes them into the halo entries of d_x and acknowledges, while the kernel's other workgroups evaluate — no launch, no
 * stream, no event of its own, and every kernel launched BEHIND that one sees the halo entries as after iem_halo_exchange.
 * Which calls can carry it follows from what their generated kernels load (iem_halo_reads): obj, and jac_coord! /
 * hess_coord! / iem_jac_hess_coord whenever the stencil rows are linear (their partials are item data — the reference's
 * derivative approximations, transform.jl:511-562); cons! of such a model reads x_k[a_r - 1] (transform.jl:535-557): if it
 * comes first, it gets the stand-alone exchange kernel in front of it — exactly iem_halo_exchange.  In solver order (obj,
 * grad!, cons!, jac_coord!, hess_coord! at a new point: ext/InfiniteExaModelsIpopt.jl:48-49) the exchange rides on obj and is
 * complete before cons! starts.  A call that neither touches nor can carry (grad!, the products) leaves it deferred;
 * iem_halo_wait, iem_synchronize, iem_halo_fold, iem_allreduce_obj_grad, iem_comm_status and a further exchange flush it
 * (stand-alone kernel).  ORDERING RULE: every rank issues its mailbox kernels in the same order — whether an evaluation
 * call carries or flushes a deferred exchange depends on the rank's own halo (rank 0 has none), so no collective
 * (fold, all-reduce) ever overtakes a deferred exchange: each flushes it first.
 * Contract: between this call and the evaluation call that carries it (or iem_halo_wait) the caller enqueues nothing that
 * writes d_x or reads its halo entries.  Graph-capturable (the decision is taken at capture time; a replay repeats it).
 * iem_halo_reads: for kernel kind `kind` (iem_kernel_info_t.kind): can it touch a halo entry through x / through a
 * variable-space v, and can it carry a deferred exchange of x. */
int iem_halo_exchange_async(iem_model *m, double *d_x);
int iem_halo_wait(iem_model *m);
int iem_halo_reads(const iem_model *m, int kind, int *out_x, int *out_v, int *out_carrier);
/* The transposed exchange, for a vector in VARIABLE space produced by a transposed operator on this rank's rows
 * (iem_jtprod): the entries of the halo copies hold what this rank's rows owe to variables the LEFT neighbour owns
 * (the x_k[a_r - 1] column of the first difference row, src/transform.jl:535-557).  They are sent to the left
 * neighbour, which ADDS them to its owned entries (one addend per entry: order-independent), and zeroed here.
 * Entries of replicated variables are summed with iem_allreduce_obj_grad (d_obj may be NULL).  Asynchronous on the
 * handle's stream, graph-capturable, bounded waits like iem_halo_exchange. */
int iem_halo_fold(iem_model *m, double *d_vec);
int iem_allreduce_obj_grad(iem_model *m, double *d_obj /* device scalar, may be NULL */, double *d_g);
/* synchronises the handle's stream; 0 = every exchange so far completed, else a bit mask of time-outs (1 / 2 halo ack / data,
 * 4 all-reduce, 8 / 16 fold ack / data).  A time-out never hangs and never goes unnoticed: the kernel that ran into it writes
 * NaN instead of the data that did not arrive, and the next iem_obj / iem_obj_end / iem_synchronize returns IEM_E_COMM (and
 * clears the mask).  The bound is the per-handle option "comm_timeout_ms" (default 5000). */
int iem_comm_status(iem_model *m, int64_t *out_status);

int iem_destroy(iem_model *m);
int iem_meta(const iem_model *m, iem_meta_t *out);
int iem_template_info(const iem_model *m, int64_t i, iem_template_info_t *out);
int iem_kernel_info(const iem_model *m, int k, iem_kernel_info_t *out);
int iem_get_host(const iem_model *m, int which, double *h_out);
int iem_set_stream(iem_model *m, void *hip_stream);
int iem_synchronize(iem_model *m);

/* ExaModels.set_parameter!(core, param, vals)  (infiniteopt_backend.jl:522-526,546):
 * overwrite θ[off .. off+len) (0-based offset) from a host array. */
int iem_set_parameter(iem_model *m, int64_t off, int64_t len, const double *h_vals);

/* ---- NLPModels evaluation API ------------------------------------------------
 * obj / grad! / cons! / jac_coord! / hess_coord!(m, x, y, vals; obj_weight) as called by
 * the solvers (ext/InfiniteExaModelsIpopt.jl:49, ext/InfiniteExaModelsMadNLP.jl:50). */
int iem_obj(iem_model *m, const double *d_x, double *h_out);
int iem_obj_device(iem_model *m, const double *d_x, double *d_out); /* async variant */
/* iem_obj in two halves: begin enqueues the objective kernel (its last workgroup writes the scalar into mapped host memory) and
 * returns at once, end waits for the value.  A solver that evaluates obj, grad!, cons!, jac_coord!, hess_coord! at one point
 * (ext/InfiniteExaModelsIpopt.jl:48-49) calls begin first and end after its last launch: the host round trip of the scalar
 * (~8 us of the ~14 us iem_obj takes on a small model) overlaps the other four calls.  One begin outstanding per handle. */
int iem_obj_begin(iem_model *m, const double *d_x);
int iem_obj_end(iem_model *m, double *h_out);
int iem_grad(iem_model *m, const double *d_x, double *d_g);
int iem_cons(iem_model *m, const double *d_x, double *d_c);
int iem_jac_coord(iem_model *m, const double *d_x, double *d_vals);
int iem_hess_coord(iem_model *m, const double *d_x, const double *d_y, double obj_weight, double *d_vals);
/* jac_coord!(m, x, jac) + hess_coord!(m, x, y, hess; obj_weight) in ONE launch: the two are independent given x and y
 * (MadNLP evaluates both at every accepted point, ext/InfiniteExaModelsMadNLP.jl:49-50,64).  Identical bytes to the two
 * calls; one launch ramp and drain instead of two, and on a shard-sized grid (about one workgroup per CU and kind) both
 * kinds are resident together. */
int iem_jac_hess_coord(iem_model *m, const double *d_x, const double *d_y, double obj_weight, double *d_jac, double *d_hess);

/* One launch per SOLVER PHASE (extensions; identical bytes to the separate calls, which remain).  An interior-point solver
 * evaluates obj + cons! at every trial point of its line search and grad! + jac_coord! + hess_coord! once per accepted
 * point (ext/InfiniteExaModelsMadNLP.jl:49-50,64, ext/InfiniteExaModelsIpopt.jl:48-49 of the reference); on the grids
 * the reference benchmarks (ESCAPE34/run_cases_gpu.jl:89-102: 1 000 - 16 000 supports) every call is one 5-7 us launch,
 * so the phase costs what its launches cost.
 *   iem_eval_trial     c = cons(x) into d_c and f = obj(x): returned in *h_obj (the call then waits for the scalar, like
 *                      iem_obj), or — h_obj == NULL — collected later by iem_obj_end (the call arms it like iem_obj_begin)
 *   iem_eval_accepted  g = grad(x), jac values, Lagrangian Hessian values (obj_weight, y) — asynchronous on the stream
 * Outputs are fully overwritten.  Handles without the fused kernels make the separate calls themselves. */
int iem_eval_trial(iem_model *m, const double *d_x, double *d_c, double *h_obj /* may be NULL */);
int iem_eval_accepted(iem_model *m, const double *d_x, const double *d_y, double obj_weight, double *d_g, double *d_jac, double *d_hess);
/* ... and all five of ONE point in one launch — the solver's first trial point is usually the accepted one; h_obj as above */
int iem_eval_all(iem_model *m, const double *d_x, const double *d_y, double obj_weight, double *d_c, double *d_g, double *d_jac, double *d_hess,
                 double *h_obj /* may be NULL */);

/* matrix-free products (NLPModels jprod! / jtprod! / hprod!; ExaModels' `prod = true` path —
 * not used by the reference's solvers): Jv (ncon), J'v (nvar), Hv (nvar) with
 * H the Hessian of obj_weight*f + y'c. */
int iem_jprod(iem_model *m, const double *d_x, const double *d_v, double *d_Jv);
int iem_jtprod(iem_model *m, const double *d_x, const double *d_v, double *d_Jtv);
int iem_hprod(iem_model *m, const double *d_x, const double *d_y, const double *d_v, double obj_weight, double *d_Hv);

/* jac_structure! / hess_structure! — one-off; `base` = 1 for Julia, 0 for C/Python.
 * Hessian pairs are lower-triangular (row >= col); COO may repeat positions. */
int iem_jac_structure(iem_model *m, int64_t *h_rows, int64_t *h_cols, int base);
int iem_hess_structure(iem_model *m, int64_t *h_rows, int64_t *h_cols, int base);
int iem_jac_structure_device(iem_model *m, int64_t *d_rows, int64_t *d_cols, int base);
int iem_hess_structure_device(iem_model *m, int64_t *d_rows, int64_t *d_cols, int base);

/* COO -> CSR value assembly with duplicate summation (the step that follows
 * jac_coord!/hess_coord! in every solver iteration).  `d_perm` lists the COO positions sorted
 * by (row, col); `d_seg[i] .. d_seg[i+1]` delimits the duplicates of CSR nonzero i (n_csr + 1
 * entries).  The plan is built once from the structure (csr.py); this call is per iteration,
 * deterministic (fixed summation order, no atomics). */
int iem_csr_values(iem_model *m, int64_t n_csr, const int64_t *d_seg, const int64_t *d_perm, const double *d_coo,
                   double *d_csr);
/* the same with 32-bit plan words (all COO positions < 2^32): half the plan traffic */
int iem_csr_values32(iem_model *m, int64_t n_csr, const uint32_t *d_seg, const uint32_t *d_perm, const double *d_coo,
                     double *d_csr);

/* y = A x for a CSR matrix with 32-bit indices (the KKT matrix csr.py / kkt.py assemble: residuals for iterative refinement).
 * One thread per row; `d_long_rows` (n_long of them, may be NULL / 0) lists the rows with more than IEM_SPMV_LONG_ROW entries
 * (a first-stage variable's column of J' has one per scenario): those get a workgroup each, summed in a fixed order. */
#define IEM_SPMV_LONG_ROW 256
int iem_csr_spmv(iem_model *m, int64_t n, const int32_t *d_rowptr, const int32_t *d_colind, const double *d_vals, const double *d_x, double *d_y,
                 int64_t n_long, const int64_t *d_long_rows);

/* ---- chain KKT solver (the linear solve that follows jac_coord!/hess_coord! in every interior-point
 * iteration; the reference hands it to MadNLPGPU + CUDSS, README.md:36-37) --------------------------------------------
 * The augmented system of a transcription whose supports couple through a derivative stencil only (transform.jl:535-557)
 * is block tridiagonal once its unknowns are grouped by support, plus a small dense border for finite / first-stage
 * variables — and its coupling blocks are NARROW: K[block k, block k-1] has entries only on a few rows R of block k (the
 * derivative-approximation rows) and a few columns C of block k-1 (the differentiated states), a property block cyclic
 * reduction preserves.  The caller (kkt_chain.py; a Julia host likewise) owns the block arrays on the device and fills
 * them every iteration: D (S x nb x nb, the diagonal blocks, symmetric), Bt (S x nc x nc, Bt[k] = K[block k, block k-1]
 * restricted to rows d_rows[0..nc) of block k and columns d_cols[0..nc) of block k-1; local indices, -1 = padding),
 * E (S x nb x ne, the coupling to the border), row-major.  iem_kkt_chain_factor runs block cyclic reduction in place
 * (hand-written kernels, csrc/iem_kkt_device.h; the inverses on the FP64 matrix cores): D becomes the inverses of the pivot
 * blocks, Bt the couplings of every level, BR (S x nc x nc: the coupling of block i + s to i at the level that eliminated i),
 * Z (as E) and Gp (S x ne x ne, per-block border Schur terms: G_schur = G - sum_k Gp[k]) are outputs; d_info[0] =
 * negative pivots (the inertia), d_info[1] = pivots below `tiny` (replaced by +-tiny: do not trust the factors).
 * iem_kkt_chain_solve: phase 0 reduces the right-hand side r (S x nb, in place; z: S x nb of scratch that must survive until
 * phase 1) and writes rBp (S x ne: r_border_schur = r_border - sum_k rBp[k]); the caller solves the ne x ne border system;
 * phase 1 substitutes back (r becomes the solution).
 * d_Bt == NULL (and d_BR == d_rows == d_cols == d_z == NULL): the blocks couple to the border only (scenario blocks of a
 * two-stage problem) — one launch instead of the levels (ne = -1: no border and a kernel shape for ONE block per launch — the pivot
 * blocks of a dense block factorisation, S = 1).  nb: multiple of 4 in 4..96, ne: multiple of 4 in 0..128, nc: multiple
 * of 4 in 4..48, nc <= nb.  Asynchronous on the handle's stream. */
int iem_kkt_chain_factor(iem_model *m, int64_t S, int nb, int ne, int nc, double *d_D, double *d_Bt, double *d_BR, const int32_t *d_rows,
                         const int32_t *d_cols, double *d_E, double *d_Z, double *d_Gp, int64_t *d_info, double tiny);
/* ONE step of that reduction (no border), for a caller that interleaves work of its own between the levels — the span-sparse
 * border of a laned 2-D grid, kkt_chain.HubChainKKT.  The chain is cut into LANES of lane_len blocks each (S = lanes x lane_len;
 * 0: one chain), every lane reduced by the same levels (block t of a lane leaves at the level s with t mod 2s == s):
 * what = 0 eliminate the blocks t = (2k+1)s of every lane, 1 fold them into the survivors t = 2ks, 2 the last remaining block
 * of every lane (t = 0), 3 clear the pivot counters (d_info).  Levels: s = 1, 2, 4, ... < lane_len. */
int iem_kkt_chain_level(iem_model *m, int64_t S, int64_t lane_len, int nb, int nc, double *d_D, double *d_Bt, double *d_BR, const int32_t *d_rows,
                        const int32_t *d_cols, int64_t *d_info, double tiny, int64_t s, int what);
int iem_kkt_chain_solve(iem_model *m, int64_t S, int nb, int ne, int nc, const double *d_Dinv, const double *d_Bt, const double *d_BR,
                        const int32_t *d_rows, const int32_t *d_cols, const double *d_Z, double *d_r, double *d_z, double *d_rBp,
                        const double *d_xB, int phase);
/* Between two levels of that reduction: the SPAN-SPARSE border columns of the blocks still alive (kkt_chain.HubChainKKT).  A block
 * alive at level s (time block t = a s, a its index among the alive) carries columns for the hubs t - (s - 1) .. t + (s - 1):
 * d_E is [alive][lanes][nq][W], W = (2 s - 1) hw, on the nq local rows d_q that ever hold a border entry (d_qr / d_qc: the
 * positions of the coupling rows / columns inside d_q).  After iem_kkt_chain_level(what = 0) and BEFORE (what = 1):
 * d_Z [eliminated][lanes][nq][W] = D_i^-1[Q, Q] E_i of the blocks just eliminated, d_En [survivors][lanes][nq][(4 s - 1) hw] = the
 * survivors' columns widened by their neighbours' terms.  last != 0: only d_Z for block 0 of every lane (d_E [1][lanes][nq][W]). */
int iem_kkt_hub_level(iem_model *m, int64_t S, int64_t lane_len, int nb, int nc, const double *d_Dinv, const double *d_Bt, const int32_t *d_q, int nq,
                      const int32_t *d_qr, int nr, const int32_t *d_qc, int ncq, int hw, int64_t s, const double *d_E, double *d_Z, double *d_En,
                      int last);
/* ... for factors made lane by lane (iem_kkt_chain_level with lane_len; ne must be 0 when lane_len != S) */
int iem_kkt_chain_solve_lanes(iem_model *m, int64_t S, int64_t lane_len, int nb, int ne, int nc, const double *d_Dinv, const double *d_Bt,
                              const double *d_BR, const int32_t *d_rows, const int32_t *d_cols, const double *d_Z, double *d_r, double *d_z,
                              double *d_rBp, const double *d_xB, int phase);
/* The same solver as ONE object — what a host without the Python layer (a Julia MadNLP linear-solver wrapper) binds.
 * iem_kkt_create analyses the model once on the host: grouping of the unknowns (variable u, then the multiplier of row
 * u - nvar) into chain blocks + border from the slab table and the Jacobian / Hessian structure, the narrow coupling, and a
 * gather plan from the positions of hess_coord! / jac_coord! values to the block entries (duplicates of the COO layout are
 * summed in a fixed order); it owns the device buffers.  Per iteration:
 *   iem_kkt_assemble(k, d_hess, d_jac, d_sigma, delta_w, delta_c)   K = [H + diag(sigma) + delta_w I, J'; J, -delta_c I]
 *                                                                  (d_hess / d_jac: what the handle's iem_hess_coord / iem_jac_coord
 *                                                                   wrote; d_sigma: nvar doubles or NULL)
 *   iem_kkt_factor(k, inertia)        block cyclic reduction + the border's Schur complement; inertia[3] = {positive, negative,
 *                                     doubtful} pivots of K (a correctly regularised system has ncon negative ones); synchronises
 *   iem_kkt_solve(k, d_rhs, d_sol)    K sol = rhs (nvar + ncon doubles each; no refinement — K x for a residual is
 *                                     iem_hprod + iem_jtprod / iem_jprod + the diagonal terms)
 * d_rhs and d_sol may be the same array.  The object borrows the model handle (its stream, device and kernel cache): destroy
 * it before iem_destroy(m).  Models whose blocks / border / coupling exceed the solver's limits (96 / 64 / 48) are refused by
 * iem_kkt_create. */
typedef struct iem_kkt iem_kkt;
typedef struct {
  int64_t S, n, n_border, block_doubles;   /* blocks, unknowns, border unknowns, doubles of the block buffer D | Bt | E | G */
  int32_t nb, ne, nc, reach, group, phase;
  /* 2-D support grids: lanes > 1 = one chain per point of the other parameter; hubs != 0 = the border (n_border unknowns, ne = 0
   * for the kernels) is kept as span-sparse hub columns: hubs_per_block of them per time block, hub_rows local rows of a block
   * ever hold a border entry, the hubs' Schur complement is a dense matrix with rows of hub_ld doubles */
  int64_t lanes, hub_ld;
  int32_t hubs, hub_rows, hubs_per_block, reserved_;
} iem_kkt_info_t;
int iem_kkt_create(iem_model *m, int group /* 0: the parameter group the stencil runs along */, iem_kkt **out);
int iem_kkt_destroy(iem_kkt *k);
int iem_kkt_info(const iem_kkt *k, iem_kkt_info_t *out);
/* the grouping itself (host arrays; any of them may be NULL): block (-1: border) and place of every unknown, the coupling's rows / columns (nc each, -1 padded) */
int iem_kkt_layout(const iem_kkt *k, int64_t *h_blk, int64_t *h_loc, int32_t *h_rows, int32_t *h_cols);
/* the host analysis alone, from a blob (no device needed; every array malloc'ed — iem_free): the grouping as above and the
 * gather plan  flat[dest[i]] = sum over k in [seg[i], seg[i + 1]) of source(perm[k]),  sources indexing the virtual array
 * hess values | jac values | sigma + delta_w per variable | -delta_c per row | 1.0 (the padding's unit diagonal) */
int iem_kkt_analyse_blob(const void *blob, size_t nbytes, int group, iem_kkt_info_t *info, int64_t **out_blk, int64_t **out_loc, int32_t **out_rows,
                         int32_t **out_cols, int64_t **out_dest, uint32_t **out_seg, uint32_t **out_perm, int64_t *out_n_dest, int64_t *out_n_perm);
int iem_kkt_assemble(iem_kkt *k, const double *d_hess, const double *d_jac, const double *d_sigma, double delta_w, double delta_c);
int iem_kkt_factor(iem_kkt *k, int64_t *out_inertia);
int iem_kkt_solve(iem_kkt *k, const double *d_rhs, double *d_sol);

/* HIP source of the solver's kernels for one (nb, ne, nc) and its cache key — for offline builds (no device needed; malloc'ed) */
int iem_kkt_source(int nb, int ne, int nc, char **out_src, uint64_t *out_key);

/* ---- kernel generation (no device needed) ---------------------------------------
 * The evaluator of a model is specialised HIP source generated from its templates
 * and compiled for gfx950 (offline into a code-object cache, or by hiprtc on a cache
 * miss).  These two calls expose the generator so a build step can pre-compile. */
int iem_emit_source(const void *blob, size_t nbytes, char **out_src, uint64_t *out_key);
/* text description of the launches (kernel name, grid, argument tables) — for tooling/tests */
int iem_emit_launch_plan(const void *blob, size_t nbytes, char **out_txt);
/* Hessian structure of a blob under the current options, computed on the host (tooling/tests;
 * arrays are malloc'ed, release with iem_free). */
int iem_blob_hess_structure(const void *blob, size_t nbytes, int base, int64_t **out_rows, int64_t **out_cols, int64_t *out_nnz);
/* values of model array `id` as the library sees it after parsing — including the float columns it
 * synthesises when it recovers a product lattice from a flat iterator (tooling/tests; malloc'ed) */
int iem_blob_array(const void *blob, size_t nbytes, int id, double **out_vals, int64_t *out_n);
void iem_free(void *p);

/* knobs: iem_set_option changes the PROCESS DEFAULTS that iem_create and iem_emit_* read; per-handle
 * values go through iem_create_opts (defaults in csrc/iem_codegen.hpp):
 *   "store_mode"   0 direct strided stores, 1 wave-level LDS-transposed stores, 2 (default)
 *                  workgroup-staged stores re-cut at 128-byte lines
 *   "overlap"      1 (default): block-store kernels overlap their tiles by 16 lanes so that every
 *                  128-byte line is written whole by one workgroup
 *   "nt_stores", "lds_slots", "reorder", "min_waves"
 *   "block"        workgroup size; 0 (default) = per model: 512, or 256 when that wastes > 2 % fewer
 *                  lanes on the rows of a 2-D / 3-D support grid (5 000 x 100: 11 tiles of 496 vs 21 of 240)
 *   "fp_contract"  0 (default): no FMA contraction — bit-comparable with the CPU oracle
 *   "split_small"  support grids of at most this many workgroups (default 64) run their templates
 *                  side by side in one launch instead of fused lane-wise (0: never)
 *   "fuse_groups"  1 (default): one launch per call even across several support grids
 *   "fuse_zero"    1 (default): scatter kernels zero untouched output entries themselves
 *   "poll_obj"     1 (default): iem_obj polls the mapped host scalar instead of a stream sync
 *   "xcd_remap", "wide_stores" (16-byte block stores): 0 (default), measured no gain;
 *   "no_fuse", "ablate": experiments / baselines only
 *   "hess_merge"   1 selects the opt-in MERGED Hessian layout (duplicate (row,col) slots of one
 *                  support summed in registers: fewer nnzh, not ExaModels' COO layout —
 *                  hess_structure!/hess_coord! stay mutually consistent).
 *   "obj_wgs"      at most this many workgroups walk the objective's tiles (default 1024; fixed, so the
 *                  summation order is — obj is bitwise reproducible)
 *   "det_shared"   1 (default): gradient / J'v / Hv entries shared by many items are reduced in a fixed
 *                  order (no float atomics); 0: one f64 atomic per wave (A/B runs)
 *   "pull_scatter" 1 (default): grad!/jtprod!/hprod! compute a stencil neighbour's addend (x[i-1] of a difference
 *                  row) on the neighbour's lane — exclusive stores, no zero fill; 0: atomics (A/B runs)
 *   "fold_colloc"  orthogonal-collocation models: the node x element boxes of the derivative rows ride on the lanes of the
 *                  support grid.  1: for grad!/jtprod!/hprod! (with the element lists of constant_over_collocation) — every
 *                  addend on the lane that owns its entry, exclusive stores, no gather plan; 2 (default): also for every
 *                  other kind (shared loads); 0: off.  "fold_max_n" (6): at most this many rows per element
 *   "det_scatter"  1 (default): scatter addends that would still be float atomics AND can meet more than one other addend in
 *                  their entry (collocation stencils, gathered indices) are parked per item and summed per entry in the
 *                  order of a plan built at create time (12 bytes of plan + 8 of scratch per addend, at most
 *                  "det_scatter_max" = 2^28 addends per kind); 2: every remaining atomic; 0: f64 atomics
 *   "det_axis"     1 (default): grad!/jtprod!/hprod! sums over a non-lane axis (an entry that depends on t only, summed over
 *                  scenarios) are parked per item and reduced in row order by a follow-up kernel; 0: f64 atomics
 *   "lazy_loads"   2 (default): product / scatter kernels with >= "lazy_min_loads" (48) loads emit a load where its value
 *                  is first used instead of at the head of the kernel (register pressure); 1: only rows of v / y; 0: never
 *   "big_batch_jac", "big_batch_hess", "big_tile", "big_batch_slots", "big_xcd"   the LARGE-GRID shape of jac_coord! /
 *                  hess_coord!: when a kind has a support grid of at least this many workgroups (defaults 4000 / 4000 — about
 *                  2e6 quadrotor supports, outputs far beyond the Infinity Cache; 0: never), all kernels of that kind run
 *                  "big_tile"-lane workgroups (1024; 0 = the model's own; the other kinds keep theirs), stage
 *                  "big_batch_slots" (48) values per barrier pair — one 96-KB workgroup per CU — and ("big_xcd" = 1) walk their
 *                  tiles XCD-aware.  A function of kind and grid sizes only, never of a timer
 *   "pair_kernel"  1 (default): the handle also carries the fused jac + hess launch behind iem_jac_hess_coord
 *   "comm_timeout_ms"  bound of every mailbox wait of this handle's exchange kernels (default 5000)
 *   "autotune"     0 (default) / 1 (opt-in): handles whose jac/hess grid has >= "autotune_min_blocks" (400) workgroups
 *                  keep a second code object with a 48-slot LDS store batch and choose per output buffer,
 *                  from the first twenty calls into it (HIP events, every call a valid evaluation), which of
 *                  the two writes that buffer faster (the buffer's physical placement decides,
 *                  by up to 10 %).  Both variants write identical bytes.  0: default code object only. */
int iem_set_option(const char *name, int64_t value);

/* per-kernel timing of the last jac/hess call pair, measured with HIP events on the
 * handle's stream (used by bench.py for the roofline line) */
int iem_time_kernels(iem_model *m, const double *d_x, const double *d_y, double *d_jac, double *d_hess,
                     int iters, double *h_ms_jac, double *h_ms_hess);

/* Store-batch tuner (options "autotune", "autotune_min_blocks"): which variant jac_coord! (kind 0) /
 * hess_coord! (kind 1) uses for output buffer d_vals — -1 still measuring (or tuner off / buffer not seen),
 * 0 the default code object, 1 the large-batch one.  Introspection only; results never depend on it. */
int iem_tuner_choice(iem_model *m, int kind, const double *d_vals, int *out_choice);
/* Decide NOW for these output buffers (either may be NULL): per kind, twelve launches of each of the handle's two
 * code objects — complete evaluations of jac_coord!(x) into d_jac / hess_coord!(x, y; obj_weight) into d_hess, ten of
 * them timed between one pair of events — then a stream synchronise; the next call uses the faster object.  A host
 * calls it once after allocating its COO value buffers (solver set-up); without it the first twenty calls of the solve
 * into a buffer measure.  No-op (IEM_OK) on handles without a second code object. */
int iem_tune(iem_model *m, const double *d_x, const double *d_y, double obj_weight, double *d_jac, double *d_hess);

const char *iem_last_error(void);
const char *iem_version(void);

#ifdef __cplusplus
}
#endif
#endif /* IEM_H */
