/* iem_oracle.c — CPU restatement of the reference's evaluation algorithm.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY UNPINNED at the kernel boundary: the arithmetic of this path lives in the
 * third-party package ExaModels.jl (pinned "0.11.2", /root/reference/Project.toml:7,23),
 * which is not vendored under /root/reference and cannot be executed here (no Julia).
 * No reference test asserts a Jacobian/Hessian value or index.
 * This file restates ExaModels' published algorithm:
 *
 *   - a template = one expression tree f(item, x, θ) shared by all items of an
 *     iterator (add_con/add_obj call sites /root/reference/src/transform.jl:458,559,
 *     597,614,700,741);
 *   - constants, item data and θ entries evaluate to plain reals, so a binary node
 *     with one real operand behaves as a unary node ("FirstFixed"/"SecondFixed");
 *   - first order: forward sweep storing local partials, then a depth-first,
 *     left-to-right reverse sweep that adds each leaf adjoint into slot
 *     comp1(visit#) of the item's block; slots are the first-occurrence-unique
 *     variable-index *expressions* of that visit sequence (o1step of them);
 *   - second order: the hrpass0 / hrpass / hdrpass recursion (top-level "+", "-"
 *     and real scaling do not create Hessian entries; below the first nonlinear
 *     node every variable visit owns a diagonal slot and every binary node crosses
 *     its two subtrees); slots are first-occurrence-unique ordered index pairs
 *     (o2step of them); rows/cols are emitted lower-triangular (row >= col);
 *     a Var×Var cross term with equal indices carries the factor 2;
 *   - offsets o0/o1/o2 are running counters in add_con/add_obj CALL order;
 *   - evaluation = serial loops, zero-fill then "+=".
 *
 * It is pinned indirectly by the reference's solver-level known answers
 * (/root/reference/test/solve.jl:146,154,187,206; test/ipopt.jl:181-186) in
 * tests/test_known_answers.py and by independent derivatives (torch float64
 * autograd on a separate Python tree evaluator) in tests/test_oracle_autodiff.py.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/iem_blob.h"

#ifdef _OPENMP
#include <omp.h>
#endif

enum { K_REAL = 0, K_VAR = 1, K_N1 = 2, K_N2 = 3 };
enum { FX_NONE = 0, FX_FIRST = 1, FX_SECOND = 2 }; /* which operand of a binary is the real */

typedef struct {
  int op, a, b;
  double imm;
  int kind;  /* K_* after constness analysis */
  int fixed; /* FX_* */
  int inner; /* the non-real child of a K_N1 */
} onode;

typedef struct {
  int mode;
  int64_t base, step[3];
  int arr;
} ofield;

typedef struct {
  int64_t c0;
  int nterms;
  int field[IEM_MAX_IDX_TERMS];
  int64_t coef[IEM_MAX_IDX_TERMS];
} oidx;

typedef struct {
  int kind, nd;
  int64_t n_items, dims[3];
  int n_if, n_ff, n_idx, n_nodes, root;
  ofield *ifields, *ffields;
  oidx *idx;
  onode *nodes;
  int lmode, umode, larr, uarr;
  double lval, uval;
  int64_t o0, o1, o2;
  int o1step, o2step;
  int n1, n2;       /* visit counts */
  int *comp1;       /* visit -> slot */
  int *comp2;
  int *slot1_idx;   /* slot -> idx id */
  int *slot2_i, *slot2_j;
} otpl;

typedef struct {
  int kind;
  int64_t n;
  const void *data; /* into blob copy */
  double fill;
  int64_t r0, rstep;
} oarr;

typedef struct orc_model {
  int64_t *blob;
  int64_t nvar, npar, ncon, nnzj, nnzh, n_tpl, n_arr;
  int minimize;
  oarr *arrs;
  otpl *tpl;
  double *theta;
  int arr_x0, arr_lvar, arr_uvar, arr_theta;
  int max_nodes;
} orc_model;

static double w2d(int64_t w) {
  double d;
  memcpy(&d, &w, 8);
  return d;
}

static double arr_f(const oarr *a, int64_t j) {
  switch (a->kind) {
    case IEM_A_F64_DATA: return ((const double *)a->data)[j];
    case IEM_A_I64_DATA: return (double)((const int64_t *)a->data)[j];
    case IEM_A_F64_FILL: return a->fill;
    default: return (double)(a->r0 + a->rstep * j);
  }
}
static int64_t arr_i(const oarr *a, int64_t j) {
  switch (a->kind) {
    case IEM_A_I64_DATA: return ((const int64_t *)a->data)[j];
    case IEM_A_I64_RANGE: return a->r0 + a->rstep * j;
    case IEM_A_F64_DATA: return (int64_t)((const double *)a->data)[j];
    default: return (int64_t)a->fill;
  }
}

/* ------------------------------------------------------------------------- */
/* operator table: value, first and second derivative                          */
/* (operator set: /root/reference/src/operators.jl:3-44)                        */
/* ------------------------------------------------------------------------- */
#define D2R 0.017453292519943295
#define R2D 57.29577951308232
#define LN2 0.6931471805599453
#define LN10 2.302585092994046

static void un_eval(int op, double x, double *f, double *d, double *h) {
  double s, c, t, u;
  switch (op) {
    case IEM_OP_NEG: *f = -x; *d = -1.0; *h = 0.0; break;
    case IEM_OP_POS: *f = x; *d = 1.0; *h = 0.0; break;
    case IEM_OP_INV: u = 1.0 / x; *f = u; *d = -(u * u); *h = 2.0 * (u * u) * u; break;
    case IEM_OP_SQRT: s = sqrt(x); *f = s; *d = 0.5 / s; *h = -0.25 / (x * s); break;
    case IEM_OP_CBRT: c = cbrt(x); *f = c; *d = 1.0 / (3.0 * c * c); *h = -2.0 / (9.0 * x * c * c); break;
    case IEM_OP_ABS: *f = fabs(x); *d = (x >= 0.0) ? 1.0 : -1.0; *h = 0.0; break;
    case IEM_OP_ABS2: *f = x * x; *d = 2.0 * x; *h = 2.0; break;
    case IEM_OP_EXP: u = exp(x); *f = u; *d = u; *h = u; break;
    case IEM_OP_EXP2: u = exp2(x); *f = u; *d = u * LN2; *h = u * LN2 * LN2; break;
    case IEM_OP_LOG: u = 1.0 / x; *f = log(x); *d = u; *h = -(u * u); break;
    case IEM_OP_LOG2: u = 1.0 / x; *f = log2(x); *d = u / LN2; *h = -(u * u) / LN2; break;
    case IEM_OP_LOG10: u = 1.0 / x; *f = log10(x); *d = u / LN10; *h = -(u * u) / LN10; break;
    case IEM_OP_LOG1P: u = 1.0 / (1.0 + x); *f = log1p(x); *d = u; *h = -(u * u); break;
    case IEM_OP_SIN: s = sin(x); c = cos(x); *f = s; *d = c; *h = -s; break;
    case IEM_OP_COS: s = sin(x); c = cos(x); *f = c; *d = -s; *h = -c; break;
    case IEM_OP_TAN: t = tan(x); u = 1.0 + t * t; *f = t; *d = u; *h = 2.0 * t * u; break;
    case IEM_OP_ASIN: u = 1.0 - x * x; s = sqrt(u); *f = asin(x); *d = 1.0 / s; *h = x / (u * s); break;
    case IEM_OP_ACOS: u = 1.0 - x * x; s = sqrt(u); *f = acos(x); *d = -1.0 / s; *h = -x / (u * s); break;
    case IEM_OP_CSC: s = 1.0 / sin(x); t = cos(x) * s; *f = s; *d = -s * t; *h = s * (t * t + s * s); break;
    case IEM_OP_SEC: c = 1.0 / cos(x); t = sin(x) * c; *f = c; *d = c * t; *h = c * (t * t + c * c); break;
    case IEM_OP_COT: t = 1.0 / tan(x); u = 1.0 + t * t; *f = t; *d = -u; *h = 2.0 * t * u; break;
    case IEM_OP_ATAN: u = 1.0 / (1.0 + x * x); *f = atan(x); *d = u; *h = -2.0 * x * u * u; break;
    case IEM_OP_ACOT: u = 1.0 / (1.0 + x * x); *f = atan(1.0 / x); *d = -u; *h = 2.0 * x * u * u; break;
    case IEM_OP_SIND: s = sin(x * D2R); c = cos(x * D2R); *f = s; *d = D2R * c; *h = -(D2R * D2R) * s; break;
    case IEM_OP_COSD: s = sin(x * D2R); c = cos(x * D2R); *f = c; *d = -D2R * s; *h = -(D2R * D2R) * c; break;
    case IEM_OP_TAND: t = tan(x * D2R); u = 1.0 + t * t; *f = t; *d = D2R * u; *h = (D2R * D2R) * 2.0 * t * u; break;
    case IEM_OP_CSCD: s = 1.0 / sin(x * D2R); t = cos(x * D2R) * s; *f = s; *d = -D2R * s * t; *h = (D2R * D2R) * s * (t * t + s * s); break;
    case IEM_OP_SECD: c = 1.0 / cos(x * D2R); t = sin(x * D2R) * c; *f = c; *d = D2R * c * t; *h = (D2R * D2R) * c * (t * t + c * c); break;
    case IEM_OP_COTD: t = 1.0 / tan(x * D2R); u = 1.0 + t * t; *f = t; *d = -D2R * u; *h = (D2R * D2R) * 2.0 * t * u; break;
    case IEM_OP_ATAND: u = 1.0 / (1.0 + x * x); *f = R2D * atan(x); *d = R2D * u; *h = -R2D * 2.0 * x * u * u; break;
    case IEM_OP_ACOTD: u = 1.0 / (1.0 + x * x); *f = R2D * atan(1.0 / x); *d = -R2D * u; *h = R2D * 2.0 * x * u * u; break;
    case IEM_OP_SINH: s = sinh(x); c = cosh(x); *f = s; *d = c; *h = s; break;
    case IEM_OP_COSH: s = sinh(x); c = cosh(x); *f = c; *d = s; *h = c; break;
    case IEM_OP_TANH: t = tanh(x); u = 1.0 - t * t; *f = t; *d = u; *h = -2.0 * t * u; break;
    case IEM_OP_CSCH: s = 1.0 / sinh(x); t = cosh(x) * s; *f = s; *d = -s * t; *h = s * (t * t + s * s); break;
    case IEM_OP_SECH: c = 1.0 / cosh(x); t = tanh(x); *f = c; *d = -c * t; *h = c * (t * t - c * c); break;
    case IEM_OP_COTH: t = 1.0 / tanh(x); u = 1.0 - t * t; *f = t; *d = u; *h = -2.0 * t * u; break;
    case IEM_OP_ATANH: u = 1.0 / (1.0 - x * x); *f = atanh(x); *d = u; *h = 2.0 * x * u * u; break;
    case IEM_OP_ACOTH: u = 1.0 / (1.0 - x * x); *f = atanh(1.0 / x); *d = u; *h = 2.0 * x * u * u; break;
    default: *f = *d = *h = NAN;
  }
}

static double bin_val(int op, double a, double b) {
  switch (op) {
    case IEM_OP_ADD: return a + b;
    case IEM_OP_SUB: return a - b;
    case IEM_OP_MUL: return a * b;
    case IEM_OP_DIV: return a / b;
    case IEM_OP_POW: return pow(a, b);
    default: return NAN;
  }
}

/* partials of a binary op; `need` = FX_NONE (all), FX_FIRST (wrt b only), FX_SECOND (wrt a only) */
static void bin_partials(int op, double a, double b, int need, double *y1, double *y2, double *h11,
                         double *h12, double *h22) {
  *y1 = *y2 = *h11 = *h12 = *h22 = 0.0;
  switch (op) {
    case IEM_OP_ADD: *y1 = 1.0; *y2 = 1.0; break;
    case IEM_OP_SUB: *y1 = 1.0; *y2 = -1.0; break;
    case IEM_OP_MUL: *y1 = b; *y2 = a; *h12 = 1.0; break;
    case IEM_OP_DIV: {
      double ib = 1.0 / b;
      *y1 = ib;
      *y2 = -a * ib * ib;
      *h12 = -(ib * ib);
      *h22 = 2.0 * a * ib * ib * ib;
      break;
    }
    case IEM_OP_POW: {
      if (need != FX_FIRST) {
        *y1 = b * pow(a, b - 1.0);
        *h11 = b * (b - 1.0) * pow(a, b - 2.0);
      }
      if (need != FX_SECOND) {
        double la = log(a), p = pow(a, b);
        *y2 = p * la;
        *h22 = p * la * la;
        if (need == FX_NONE) *h12 = pow(a, b - 1.0) * (1.0 + b * la);
      }
      break;
    }
    default: break;
  }
}

/* ------------------------------------------------------------------------- */
/* blob parsing                                                              */
/* ------------------------------------------------------------------------- */
typedef struct {
  int *v;
  int n, cap;
} ivec;
static void iv_push(ivec *a, int x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? 2 * a->cap : 16;
    a->v = (int *)realloc(a->v, sizeof(int) * a->cap);
  }
  a->v[a->n++] = x;
}

/* symbolic traversals: record the visit sequences --------------------------- */
static void sym_grpass(const otpl *t, int n, ivec *out) {
  const onode *nd = &t->nodes[n];
  switch (nd->kind) {
    case K_VAR: iv_push(out, nd->a); break;
    case K_N1: sym_grpass(t, nd->inner, out); break;
    case K_N2: sym_grpass(t, nd->a, out); sym_grpass(t, nd->b, out); break;
    default: break;
  }
}
static void sym_hdrpass(const otpl *t, int n1, int n2, ivec *oi, ivec *oj) {
  const onode *a = &t->nodes[n1], *b = &t->nodes[n2];
  if (a->kind == K_REAL || b->kind == K_REAL) return;
  if (a->kind == K_VAR && b->kind == K_VAR) {
    iv_push(oi, a->a);
    iv_push(oj, b->a);
  } else if (a->kind == K_N1 && b->kind == K_N1) {
    sym_hdrpass(t, a->inner, b->inner, oi, oj);
  } else if (a->kind == K_VAR && b->kind == K_N1) {
    sym_hdrpass(t, n1, b->inner, oi, oj);
  } else if (a->kind == K_N1 && b->kind == K_VAR) {
    sym_hdrpass(t, a->inner, n2, oi, oj);
  } else if (a->kind == K_N2 && b->kind == K_N2) {
    sym_hdrpass(t, a->a, b->a, oi, oj);
    sym_hdrpass(t, a->a, b->b, oi, oj);
    sym_hdrpass(t, a->b, b->a, oi, oj);
    sym_hdrpass(t, a->b, b->b, oi, oj);
  } else if (a->kind == K_N1 && b->kind == K_N2) {
    sym_hdrpass(t, a->inner, b->a, oi, oj);
    sym_hdrpass(t, a->inner, b->b, oi, oj);
  } else if (a->kind == K_N2 && b->kind == K_N1) {
    sym_hdrpass(t, a->a, b->inner, oi, oj);
    sym_hdrpass(t, a->b, b->inner, oi, oj);
  } else if (a->kind == K_VAR && b->kind == K_N2) {
    sym_hdrpass(t, n1, b->a, oi, oj);
    sym_hdrpass(t, n1, b->b, oi, oj);
  } else if (a->kind == K_N2 && b->kind == K_VAR) {
    sym_hdrpass(t, a->a, n2, oi, oj);
    sym_hdrpass(t, a->b, n2, oi, oj);
  }
}
static void sym_hrpass(const otpl *t, int n, ivec *oi, ivec *oj) {
  const onode *nd = &t->nodes[n];
  switch (nd->kind) {
    case K_VAR: iv_push(oi, nd->a); iv_push(oj, nd->a); break;
    case K_N1: sym_hrpass(t, nd->inner, oi, oj); break;
    case K_N2:
      sym_hrpass(t, nd->a, oi, oj);
      sym_hrpass(t, nd->b, oi, oj);
      sym_hdrpass(t, nd->a, nd->b, oi, oj);
      break;
    default: break;
  }
}
/* is this node one of the "linear at top level" forms hrpass0 passes through? */
static int is_linear_n1(const onode *nd) {
  if (nd->kind != K_N1) return 0;
  if (nd->fixed != FX_NONE) return nd->op == IEM_OP_MUL || nd->op == IEM_OP_ADD || nd->op == IEM_OP_SUB;
  return nd->op == IEM_OP_NEG || nd->op == IEM_OP_POS;
}
static void sym_hrpass0(const otpl *t, int n, ivec *oi, ivec *oj) {
  const onode *nd = &t->nodes[n];
  if (nd->kind == K_VAR || nd->kind == K_REAL) return;
  if (is_linear_n1(nd)) {
    sym_hrpass0(t, nd->inner, oi, oj);
  } else if (nd->kind == K_N2 && (nd->op == IEM_OP_ADD || nd->op == IEM_OP_SUB)) {
    sym_hrpass0(t, nd->a, oi, oj);
    sym_hrpass0(t, nd->b, oi, oj);
  } else {
    sym_hrpass(t, n, oi, oj);
  }
}

static int parse_template(orc_model *m, const int64_t *w, otpl *t) {
  int p = 0;
  t->kind = (int)w[p++];
  t->n_items = w[p++];
  t->nd = (int)w[p++];
  for (int d = 0; d < 3; ++d) t->dims[d] = w[p++];
  p += 4; /* grid hint: unused by the oracle */
  t->n_if = (int)w[p++];
  t->n_ff = (int)w[p++];
  t->n_idx = (int)w[p++];
  t->n_nodes = (int)w[p++];
  t->root = (int)w[p++];
  t->lmode = (int)w[p++]; t->lval = w2d(w[p++]); t->larr = (int)w[p++];
  t->umode = (int)w[p++]; t->uval = w2d(w[p++]); t->uarr = (int)w[p++];
  t->ifields = (ofield *)calloc(t->n_if ? t->n_if : 1, sizeof(ofield));
  t->ffields = (ofield *)calloc(t->n_ff ? t->n_ff : 1, sizeof(ofield));
  for (int f = 0; f < t->n_if + t->n_ff; ++f) {
    ofield *fl = f < t->n_if ? &t->ifields[f] : &t->ffields[f - t->n_if];
    fl->mode = (int)w[p++];
    fl->base = w[p++];
    for (int d = 0; d < 3; ++d) fl->step[d] = w[p++];
    fl->arr = (int)w[p++];
  }
  t->idx = (oidx *)calloc(t->n_idx ? t->n_idx : 1, sizeof(oidx));
  for (int i = 0; i < t->n_idx; ++i) {
    t->idx[i].c0 = w[p++];
    t->idx[i].nterms = (int)w[p++];
    for (int j = 0; j < IEM_MAX_IDX_TERMS; ++j) {
      t->idx[i].field[j] = (int)w[p++];
      t->idx[i].coef[j] = w[p++];
    }
  }
  t->nodes = (onode *)calloc(t->n_nodes, sizeof(onode));
  for (int n = 0; n < t->n_nodes; ++n) {
    onode *nd = &t->nodes[n];
    nd->op = (int)w[p++];
    nd->a = (int)w[p++];
    nd->b = (int)w[p++];
    nd->imm = w2d(w[p++]);
    nd->fixed = FX_NONE;
    nd->inner = -1;
    if (nd->op == IEM_OP_VAR) nd->kind = K_VAR;
    else if (nd->op <= IEM_OP_PAR) nd->kind = K_REAL;
    else if (IEM_OP_IS_UNARY(nd->op)) {
      nd->kind = t->nodes[nd->a].kind == K_REAL ? K_REAL : K_N1;
      nd->inner = nd->a;
    } else {
      int ka = t->nodes[nd->a].kind, kb = t->nodes[nd->b].kind;
      if (ka == K_REAL && kb == K_REAL) nd->kind = K_REAL;
      else if (ka == K_REAL) { nd->kind = K_N1; nd->fixed = FX_FIRST; nd->inner = nd->b; }
      else if (kb == K_REAL) { nd->kind = K_N1; nd->fixed = FX_SECOND; nd->inner = nd->a; }
      else nd->kind = K_N2;
    }
  }
  if (t->n_nodes > m->max_nodes) m->max_nodes = t->n_nodes;

  /* slot compressors (ExaModels' Compressor): first-occurrence unique */
  ivec v1 = {0}, vi = {0}, vj = {0};
  sym_grpass(t, t->root, &v1);
  sym_hrpass0(t, t->root, &vi, &vj);
  t->n1 = v1.n;
  t->comp1 = (int *)malloc(sizeof(int) * (v1.n ? v1.n : 1));
  t->slot1_idx = (int *)malloc(sizeof(int) * (v1.n ? v1.n : 1));
  t->o1step = 0;
  for (int i = 0; i < v1.n; ++i) {
    int s = -1;
    for (int j = 0; j < t->o1step; ++j)
      if (t->slot1_idx[j] == v1.v[i]) { s = j; break; }
    if (s < 0) { s = t->o1step; t->slot1_idx[t->o1step++] = v1.v[i]; }
    t->comp1[i] = s;
  }
  t->n2 = vi.n;
  t->comp2 = (int *)malloc(sizeof(int) * (vi.n ? vi.n : 1));
  t->slot2_i = (int *)malloc(sizeof(int) * (vi.n ? vi.n : 1));
  t->slot2_j = (int *)malloc(sizeof(int) * (vi.n ? vi.n : 1));
  t->o2step = 0;
  for (int i = 0; i < vi.n; ++i) {
    int s = -1;
    for (int j = 0; j < t->o2step; ++j)
      if (t->slot2_i[j] == vi.v[i] && t->slot2_j[j] == vj.v[i]) { s = j; break; }
    if (s < 0) { s = t->o2step; t->slot2_i[s] = vi.v[i]; t->slot2_j[s] = vj.v[i]; t->o2step++; }
    t->comp2[i] = s;
  }
  free(v1.v); free(vi.v); free(vj.v);
  return p;
}

orc_model *orc_create(const void *blob, size_t nbytes) {
  if (nbytes < 8 * IEM_HDR_WORDS) return NULL;
  const int64_t *w0 = (const int64_t *)blob;
  if (w0[0] != IEM_BLOB_MAGIC || w0[1] != IEM_BLOB_VERSION) return NULL;
  if ((size_t)w0[8] * 8 != nbytes) return NULL;
  orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
  m->blob = (int64_t *)malloc(nbytes);
  memcpy(m->blob, blob, nbytes);
  const int64_t *w = m->blob;
  m->nvar = w[2]; m->npar = w[3]; m->ncon = w[4]; m->n_tpl = w[5]; m->n_arr = w[6];
  m->minimize = (int)w[7];
  m->arr_x0 = (int)w[10]; m->arr_lvar = (int)w[11]; m->arr_uvar = (int)w[12]; m->arr_theta = (int)w[13];
  m->arrs = (oarr *)calloc(m->n_arr ? m->n_arr : 1, sizeof(oarr));
  const int64_t *aw = w + IEM_HDR_WORDS;
  for (int64_t i = 0; i < m->n_arr; ++i, aw += IEM_ARR_WORDS) {
    oarr *a = &m->arrs[i];
    a->kind = (int)aw[0];
    a->n = aw[1];
    a->data = w + aw[2];
    a->fill = w2d(aw[3]);
    a->r0 = aw[3];
    a->rstep = aw[4];
  }
  m->theta = (double *)malloc(sizeof(double) * (m->npar ? m->npar : 1));
  for (int64_t i = 0; i < m->npar; ++i) m->theta[i] = arr_f(&m->arrs[m->arr_theta], i);
  m->tpl = (otpl *)calloc(m->n_tpl ? m->n_tpl : 1, sizeof(otpl));
  const int64_t *tw = w + IEM_HDR_WORDS + IEM_ARR_WORDS * m->n_arr;
  int64_t o0 = 0, o1 = 0, o2 = 0;
  for (int64_t i = 0; i < m->n_tpl; ++i) {
    otpl *t = &m->tpl[i];
    parse_template(m, w + tw[i], t);
    t->o2 = o2;
    o2 += t->n_items * t->o2step;
    if (t->kind == IEM_T_CON) {
      t->o0 = o0; o0 += t->n_items;
      t->o1 = o1; o1 += t->n_items * t->o1step;
    }
  }
  m->nnzj = o1;
  m->nnzh = o2;
  if (o0 != m->ncon) { /* inconsistent blob */ }
  return m;
}

void orc_destroy(orc_model *m) {
  if (!m) return;
  for (int64_t i = 0; i < m->n_tpl; ++i) {
    otpl *t = &m->tpl[i];
    free(t->ifields); free(t->ffields); free(t->idx); free(t->nodes);
    free(t->comp1); free(t->comp2); free(t->slot1_idx); free(t->slot2_i); free(t->slot2_j);
  }
  free(m->tpl); free(m->arrs); free(m->theta); free(m->blob); free(m);
}

/* meta: nvar, ncon, npar, nnzj, nnzh, minimize, n_templates */
void orc_meta(const orc_model *m, int64_t *out) {
  out[0] = m->nvar; out[1] = m->ncon; out[2] = m->npar; out[3] = m->nnzj; out[4] = m->nnzh;
  out[5] = m->minimize; out[6] = m->n_tpl;
}

/* per-template layout: kind, n_items, o0, o1, o2, o1step, o2step */
void orc_template_info(const orc_model *m, int64_t i, int64_t *out) {
  const otpl *t = &m->tpl[i];
  out[0] = t->kind; out[1] = t->n_items; out[2] = t->o0; out[3] = t->o1; out[4] = t->o2;
  out[5] = t->o1step; out[6] = t->o2step;
}

/* which: 0 x0, 1 lvar, 2 uvar, 3 lcon, 4 ucon, 5 y0, 6 theta */
void orc_get_array(const orc_model *m, int which, double *out) {
  if (which <= 2) {
    const oarr *a = &m->arrs[which == 0 ? m->arr_x0 : which == 1 ? m->arr_lvar : m->arr_uvar];
    for (int64_t i = 0; i < m->nvar; ++i) out[i] = arr_f(a, i);
  } else if (which == 3 || which == 4) {
    for (int64_t i = 0; i < m->n_tpl; ++i) {
      const otpl *t = &m->tpl[i];
      if (t->kind != IEM_T_CON) continue;
      int mode = which == 3 ? t->lmode : t->umode;
      for (int64_t k = 0; k < t->n_items; ++k)
        out[t->o0 + k] = mode ? arr_f(&m->arrs[which == 3 ? t->larr : t->uarr], k)
                              : (which == 3 ? t->lval : t->uval);
    }
  } else if (which == 5) {
    for (int64_t i = 0; i < m->ncon; ++i) out[i] = 0.0;
  } else {
    memcpy(out, m->theta, sizeof(double) * m->npar);
  }
}

void orc_set_parameter(orc_model *m, int64_t off, int64_t len, const double *vals) {
  memcpy(m->theta + off, vals, sizeof(double) * len);
}

/* ------------------------------------------------------------------------- */
/* per-item evaluation                                                       */
/* ------------------------------------------------------------------------- */
typedef struct {
  double *val, *y1, *y2, *h11, *h12, *h22; /* per node */
  int64_t *idx;                             /* per index expr: 1-based variable index */
  int64_t ifv[16];
} scratch;

static scratch *scratch_new(const orc_model *m) {
  scratch *s = (scratch *)calloc(1, sizeof(scratch));
  int n = m->max_nodes ? m->max_nodes : 1;
  s->val = (double *)malloc(sizeof(double) * n * 6);
  s->y1 = s->val + n; s->y2 = s->y1 + n; s->h11 = s->y2 + n; s->h12 = s->h11 + n; s->h22 = s->h12 + n;
  s->idx = (int64_t *)malloc(sizeof(int64_t) * n);
  return s;
}
static void scratch_free(scratch *s) { free(s->val); free(s->idx); free(s); }

static inline int64_t field_pos(const ofield *f, const int64_t *kc) {
  return f->base + f->step[0] * kc[0] + f->step[1] * kc[1] + f->step[2] * kc[2];
}

static void item_indices(const orc_model *m, const otpl *t, int64_t k, scratch *s, int64_t *kc) {
  kc[0] = k % t->dims[0];
  kc[1] = (k / t->dims[0]) % t->dims[1];
  kc[2] = k / (t->dims[0] * t->dims[1]);
  for (int f = 0; f < t->n_if; ++f) {
    const ofield *fl = &t->ifields[f];
    int64_t p = field_pos(fl, kc);
    s->ifv[f] = fl->mode == IEM_F_AFFINE ? p : arr_i(&m->arrs[fl->arr], p);
  }
  for (int i = 0; i < t->n_idx; ++i) {
    const oidx *ix = &t->idx[i];
    int64_t v = ix->c0;
    for (int j = 0; j < ix->nterms; ++j) v += ix->coef[j] * s->ifv[ix->field[j]];
    s->idx[i] = v;
  }
}

/* order: 0 = values only, 1 = + first partials, 2 = + second partials */
static void forward(const orc_model *m, const otpl *t, int64_t k, const double *x, scratch *s, int order) {
  int64_t kc[3];
  item_indices(m, t, k, s, kc);
  for (int n = 0; n < t->n_nodes; ++n) {
    const onode *nd = &t->nodes[n];
    switch (nd->op) {
      case IEM_OP_CONST: s->val[n] = nd->imm; break;
      case IEM_OP_DATA: {
        const ofield *fl = &t->ffields[nd->a];
        s->val[n] = arr_f(&m->arrs[fl->arr], field_pos(fl, kc));
        break;
      }
      case IEM_OP_PAR: s->val[n] = m->theta[s->idx[nd->a] - 1]; break;
      case IEM_OP_VAR: s->val[n] = x[s->idx[nd->a] - 1]; break;
      default:
        if (IEM_OP_IS_UNARY(nd->op)) {
          double f, d, h;
          un_eval(nd->op, s->val[nd->a], &f, &d, &h);
          s->val[n] = f;
          if (order >= 1) { s->y1[n] = d; s->h11[n] = h; }
        } else {
          double a = s->val[nd->a], b = s->val[nd->b];
          s->val[n] = bin_val(nd->op, a, b);
          if (order >= 1 && nd->kind != K_REAL) {
            double y1, y2, h11, h12, h22;
            bin_partials(nd->op, a, b, nd->fixed, &y1, &y2, &h11, &h12, &h22);
            if (nd->fixed == FX_FIRST) { s->y1[n] = y2; s->h11[n] = h22; }      /* unary in b */
            else if (nd->fixed == FX_SECOND) { s->y1[n] = y1; s->h11[n] = h11; } /* unary in a */
            else { s->y1[n] = y1; s->y2[n] = y2; s->h11[n] = h11; s->h12[n] = h12; s->h22[n] = h22; }
          }
        }
    }
  }
}

/* first-order reverse sweep ------------------------------------------------ */
typedef struct {
  const otpl *t;
  const scratch *s;
  double *dense;      /* grad: dense[idx-1] += adj */
  double *coo;        /* jac values block of this item */
} rctx;

static int grpass(const rctx *c, int n, int cnt, double adj) {
  const onode *nd = &c->t->nodes[n];
  switch (nd->kind) {
    case K_VAR:
      if (c->dense) c->dense[c->s->idx[nd->a] - 1] += adj;
      else c->coo[c->t->comp1[cnt]] += adj;
      return cnt + 1;
    case K_N1: return grpass(c, nd->inner, cnt, adj * c->s->y1[n]);
    case K_N2:
      cnt = grpass(c, nd->a, cnt, adj * c->s->y1[n]);
      return grpass(c, nd->b, cnt, adj * c->s->y2[n]);
    default: return cnt;
  }
}

/* second-order reverse sweep ----------------------------------------------- */
typedef struct {
  const otpl *t;
  const scratch *s;
  double *coo; /* hess values block of this item */
} hctx;

static int hdrpass(const hctx *c, int n1, int n2, int cnt, double adj) {
  const onode *a = &c->t->nodes[n1], *b = &c->t->nodes[n2];
  const scratch *s = c->s;
  if (a->kind == K_REAL || b->kind == K_REAL) return cnt;
  if (a->kind == K_VAR && b->kind == K_VAR) {
    if (s->idx[a->a] == s->idx[b->a]) c->coo[c->t->comp2[cnt]] += 2.0 * adj;
    else c->coo[c->t->comp2[cnt]] += adj;
    return cnt + 1;
  } else if (a->kind == K_N1 && b->kind == K_N1) {
    return hdrpass(c, a->inner, b->inner, cnt, adj * s->y1[n1] * s->y1[n2]);
  } else if (a->kind == K_VAR && b->kind == K_N1) {
    return hdrpass(c, n1, b->inner, cnt, adj * s->y1[n2]);
  } else if (a->kind == K_N1 && b->kind == K_VAR) {
    return hdrpass(c, a->inner, n2, cnt, adj * s->y1[n1]);
  } else if (a->kind == K_N2 && b->kind == K_N2) {
    cnt = hdrpass(c, a->a, b->a, cnt, adj * s->y1[n1] * s->y1[n2]);
    cnt = hdrpass(c, a->a, b->b, cnt, adj * s->y1[n1] * s->y2[n2]);
    cnt = hdrpass(c, a->b, b->a, cnt, adj * s->y2[n1] * s->y1[n2]);
    return hdrpass(c, a->b, b->b, cnt, adj * s->y2[n1] * s->y2[n2]);
  } else if (a->kind == K_N1 && b->kind == K_N2) {
    cnt = hdrpass(c, a->inner, b->a, cnt, adj * s->y1[n1] * s->y1[n2]);
    return hdrpass(c, a->inner, b->b, cnt, adj * s->y1[n1] * s->y2[n2]);
  } else if (a->kind == K_N2 && b->kind == K_N1) {
    cnt = hdrpass(c, a->a, b->inner, cnt, adj * s->y1[n1] * s->y1[n2]);
    return hdrpass(c, a->b, b->inner, cnt, adj * s->y2[n1] * s->y1[n2]);
  } else if (a->kind == K_VAR && b->kind == K_N2) {
    cnt = hdrpass(c, n1, b->a, cnt, adj * s->y1[n2]);
    return hdrpass(c, n1, b->b, cnt, adj * s->y2[n2]);
  } else { /* N2 x VAR */
    cnt = hdrpass(c, a->a, n2, cnt, adj * s->y1[n1]);
    return hdrpass(c, a->b, n2, cnt, adj * s->y2[n1]);
  }
}

static int hrpass(const hctx *c, int n, int cnt, double adj, double adj2) {
  const onode *nd = &c->t->nodes[n];
  const scratch *s = c->s;
  switch (nd->kind) {
    case K_VAR:
      c->coo[c->t->comp2[cnt]] += adj2;
      return cnt + 1;
    case K_N1: {
      double y = s->y1[n];
      return hrpass(c, nd->inner, cnt, adj * y, adj2 * (y * y) + adj * s->h11[n]);
    }
    case K_N2: {
      double y1 = s->y1[n], y2 = s->y2[n];
      double adj2y1y2 = adj2 * y1 * y2;
      double adjh12 = adj * s->h12[n];
      cnt = hrpass(c, nd->a, cnt, adj * y1, adj2 * (y1 * y1) + adj * s->h11[n]);
      cnt = hrpass(c, nd->b, cnt, adj * y2, adj2 * (y2 * y2) + adj * s->h22[n]);
      return hdrpass(c, nd->a, nd->b, cnt, adj2y1y2 + adjh12);
    }
    default: return cnt;
  }
}

static int hrpass0(const hctx *c, int n, int cnt, double adj, double adj2) {
  const onode *nd = &c->t->nodes[n];
  const scratch *s = c->s;
  if (nd->kind == K_VAR || nd->kind == K_REAL) return cnt;
  if (is_linear_n1(nd)) {
    double y = s->y1[n];
    return hrpass0(c, nd->inner, cnt, adj * y, adj2 * (y * y));
  }
  if (nd->kind == K_N2 && (nd->op == IEM_OP_ADD || nd->op == IEM_OP_SUB)) {
    cnt = hrpass0(c, nd->a, cnt, adj * s->y1[n], adj2);
    return hrpass0(c, nd->b, cnt, adj * s->y2[n], adj2);
  }
  return hrpass(c, n, cnt, adj, adj2);
}

/* ------------------------------------------------------------------------- */
/* NLPModels-style entry points                                              */
/* ------------------------------------------------------------------------- */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

double orc_obj(const orc_model *m, const double *x) {
  double total = 0.0;
  for (int64_t i = 0; i < m->n_tpl; ++i) {
    const otpl *t = &m->tpl[i];
    if (t->kind != IEM_T_OBJ) continue;
    double acc = 0.0;
#pragma omp parallel num_threads(g_threads) reduction(+ : acc)
    {
      scratch *s = scratch_new(m);
#pragma omp for schedule(static)
      for (int64_t k = 0; k < t->n_items; ++k) {
        forward(m, t, k, x, s, 0);
        acc += s->val[t->root];
      }
      scratch_free(s);
    }
    total += acc;
  }
  return total;
}

void orc_cons(const orc_model *m, const double *x, double *cvals) {
  for (int64_t i = 0; i < m->n_tpl; ++i) {
    const otpl *t = &m->tpl[i];
    if (t->kind != IEM_T_CON) continue;
#pragma omp parallel num_threads(g_threads)
    {
      scratch *s = scratch_new(m);
#pragma omp for schedule(static)
      for (int64_t k = 0; k < t->n_items; ++k) {
        forward(m, t, k, x, s, 0);
        cvals[t->o0 + k] = s->val[t->root];
      }
      scratch_free(s);
    }
  }
}

/* dense gradient; serial (scatter-add into shared entries) */
void orc_grad(const orc_model *m, const double *x, double *g) {
  memset(g, 0, sizeof(double) * m->nvar);
  scratch *s = scratch_new(m);
  for (int64_t i = 0; i < m->n_tpl; ++i) {
    const otpl *t = &m->tpl[i];
    if (t->kind != IEM_T_OBJ) continue;
    for (int64_t k = 0; k < t->n_items; ++k) {
      forward(m, t, k, x, s, 1);
      rctx c = {t, s, g, NULL};
      grpass(&c, t->root, 0, 1.0);
    }
  }
  scratch_free(s);
}

void orc_jac_structure(const orc_model *m, int64_t *rows, int64_t *cols, int base) {
  scratch *s = scratch_new(m);
  int64_t kc[3];
  for (int64_t i = 0; i < m->n_tpl; ++i) {
    const otpl *t = &m->tpl[i];
    if (t->kind != IEM_T_CON) continue;
    for (int64_t k = 0; k < t->n_items; ++k) {
      item_indices(m, t, k, s, kc);
      int64_t o = t->o1 + (int64_t)t->o1step * k;
      for (int sl = 0; sl < t->o1step; ++sl) {
        rows[o + sl] = t->o0 + k + base;
        cols[o + sl] = s->idx[t->slot1_idx[sl]] - 1 + base;
      }
    }
  }
  scratch_free(s);
}

void orc_jac_coord(const orc_model *m, const double *x, double *vals) {
  /* ExaModels: fill!(jac, 0), then one serial "+=" loop per template.  The OpenMP variant
   * (cpu_baseline "all cores") keeps one parallel region per call; items are independent. */
#pragma omp parallel num_threads(g_threads)
  {
    scratch *s = scratch_new(m);
#pragma omp for schedule(static)
    for (int64_t j = 0; j < m->nnzj; ++j) vals[j] = 0.0;
    for (int64_t i = 0; i < m->n_tpl; ++i) {
      const otpl *t = &m->tpl[i];
      if (t->kind != IEM_T_CON) continue;
#pragma omp for schedule(static) nowait
      for (int64_t k = 0; k < t->n_items; ++k) {
        forward(m, t, k, x, s, 1);
        rctx c = {t, s, NULL, vals + t->o1 + (int64_t)t->o1step * k};
        grpass(&c, t->root, 0, 1.0);
      }
    }
    scratch_free(s);
  }
}

void orc_hess_structure(const orc_model *m, int64_t *rows, int64_t *cols, int base) {
  scratch *s = scratch_new(m);
  int64_t kc[3];
  for (int64_t i = 0; i < m->n_tpl; ++i) {
    const otpl *t = &m->tpl[i];
    for (int64_t k = 0; k < t->n_items; ++k) {
      item_indices(m, t, k, s, kc);
      int64_t o = t->o2 + (int64_t)t->o2step * k;
      for (int sl = 0; sl < t->o2step; ++sl) {
        int64_t a = s->idx[t->slot2_i[sl]], b = s->idx[t->slot2_j[sl]];
        rows[o + sl] = (a >= b ? a : b) - 1 + base; /* lower triangle: row >= col */
        cols[o + sl] = (a >= b ? b : a) - 1 + base;
      }
    }
  }
  scratch_free(s);
}

/* Hessian of  obj_weight*f(x) + sum_k y_k c_k(x)  (NLPModels hess_coord!(m, x, y, vals; obj_weight)) */
void orc_hess_coord(const orc_model *m, const double *x, const double *y, double obj_weight, double *vals) {
#pragma omp parallel num_threads(g_threads)
  {
    scratch *s = scratch_new(m);
#pragma omp for schedule(static)
    for (int64_t j = 0; j < m->nnzh; ++j) vals[j] = 0.0;
    for (int64_t i = 0; i < m->n_tpl; ++i) {
      const otpl *t = &m->tpl[i];
      if (t->o2step == 0) continue;
#pragma omp for schedule(static) nowait
      for (int64_t k = 0; k < t->n_items; ++k) {
        forward(m, t, k, x, s, 2);
        hctx c = {t, s, vals + t->o2 + (int64_t)t->o2step * k};
        double adj = t->kind == IEM_T_OBJ ? obj_weight : y[t->o0 + k];
        hrpass0(&c, t->root, 0, adj, 0.0);
      }
    }
    scratch_free(s);
  }
}

/* matrix-free products, restated through the COO output (checker for jprod!/jtprod!/hprod!) */
void orc_jprod(const orc_model *m, const double *x, const double *v, double *Jv) {
  double *vals = (double *)malloc(sizeof(double) * (m->nnzj ? m->nnzj : 1));
  int64_t *r = (int64_t *)malloc(sizeof(int64_t) * (m->nnzj ? m->nnzj : 1)), *c = (int64_t *)malloc(sizeof(int64_t) * (m->nnzj ? m->nnzj : 1));
  orc_jac_coord(m, x, vals);
  orc_jac_structure(m, r, c, 0);
  for (int64_t i = 0; i < m->ncon; ++i) Jv[i] = 0.0;
  for (int64_t e = 0; e < m->nnzj; ++e) Jv[r[e]] += vals[e] * v[c[e]];
  free(vals); free(r); free(c);
}
void orc_jtprod(const orc_model *m, const double *x, const double *v, double *Jtv) {
  double *vals = (double *)malloc(sizeof(double) * (m->nnzj ? m->nnzj : 1));
  int64_t *r = (int64_t *)malloc(sizeof(int64_t) * (m->nnzj ? m->nnzj : 1)), *c = (int64_t *)malloc(sizeof(int64_t) * (m->nnzj ? m->nnzj : 1));
  orc_jac_coord(m, x, vals);
  orc_jac_structure(m, r, c, 0);
  for (int64_t i = 0; i < m->nvar; ++i) Jtv[i] = 0.0;
  for (int64_t e = 0; e < m->nnzj; ++e) Jtv[c[e]] += vals[e] * v[r[e]];
  free(vals); free(r); free(c);
}
void orc_hprod(const orc_model *m, const double *x, const double *y, const double *v, double obj_weight, double *Hv) {
  double *vals = (double *)malloc(sizeof(double) * (m->nnzh ? m->nnzh : 1));
  int64_t *r = (int64_t *)malloc(sizeof(int64_t) * (m->nnzh ? m->nnzh : 1)), *c = (int64_t *)malloc(sizeof(int64_t) * (m->nnzh ? m->nnzh : 1));
  orc_hess_coord(m, x, y, obj_weight, vals);
  orc_hess_structure(m, r, c, 0);
  for (int64_t i = 0; i < m->nvar; ++i) Hv[i] = 0.0;
  for (int64_t e = 0; e < m->nnzh; ++e) {
    Hv[r[e]] += vals[e] * v[c[e]];
    if (r[e] != c[e]) Hv[c[e]] += vals[e] * v[r[e]];
  }
  free(vals); free(r); free(c);
}
