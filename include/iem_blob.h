/* iem_blob.h — wire format of a transcribed model ("blob").
 *
 * The blob is what crosses the C-ABI in iem_create(): the list of SIMD templates
 * (expression tree + item iterator + bounds) and the core buffers that
 * /root/reference/src/transform.jl builds through ExaModels.add_var / add_par /
 * add_con / add_obj (transform.jl:113,127,154,179,458,559,597,614,700,741).
 *
 * Everything is a little-endian stream of 8-byte words (int64 or IEEE double,
 * bit-cast), so a parser is `const int64_t *w`.  Indices inside trees are
 * 1-based exactly as ExaModels.Var(i) / ParameterNode(i) hold them.
 *
 *   word 0            IEM_BLOB_MAGIC
 *   word 1            IEM_BLOB_VERSION
 *   word 2..9         nvar, npar, ncon, n_templates, n_arrays, minimize(0/1),
 *                     total_words, slab_table_word_offset (0: none)
 *   word 10..13       array ids of x0, lvar, uvar, theta
 *   array table       n_arrays x 6 words {kind, n, data_word_offset, a, b, reserved}
 *                       kind F64_DATA : n doubles at data_word_offset
 *                       kind I64_DATA : n int64   at data_word_offset
 *                       kind F64_FILL : n copies of (double)a
 *                       kind I64_RANGE: a + b*j, j = 0..n-1
 *   template table    n_templates x 1 word: word offset of each template record
 *   template records  see below
 *   slab table        optional (header word 9): n_slabs, then n_slabs x 8 words
 *                       {offset (0-based), nd, dims[3], group[3]} — the add_var slabs of x in order
 *                       (transform.jl:113,154), tiling 0..nvar; group[a] = infinite-parameter group
 *                       (1-based, the same ids the grid hints use) axis a runs over, 0 = none.  Only
 *                       sharding reads it (iem_create_sharded cuts a rank's support window from it).
 *   array payloads
 *
 * Template record (all words):
 *   kind (IEM_T_OBJ / IEM_T_CON), n_items, nd, dims[3],
 *   grid_id, origin[3]            fusion hint: grid coordinate of item coord 0 per dim
 *                                 (grid_id < 0: not on a support grid)
 *   n_ifields, n_ffields, n_idx, n_nodes, root,
 *   lcon_mode, lcon_val, lcon_arr, ucon_mode, ucon_val, ucon_arr   (mode 0 = scalar value, 1 = array id)
 *   ifields  n_ifields x 6 {mode, base, step[3], arr}
 *              mode IEM_F_AFFINE: value = base + sum_d step[d]*k_d
 *              mode IEM_F_GATHER: value = arr[base + sum_d step[d]*k_d]   (0-based into arr)
 *   ffields  n_ffields x 6 {mode(=GATHER), base, step[3], arr}
 *   idx      n_idx x (2 + 2*IEM_MAX_IDX_TERMS) {c0, nterms, (ifield, coef) x IEM_MAX_IDX_TERMS}
 *              value = c0 + sum coef*ifield      (1-based index into x or theta)
 *   nodes    n_nodes x 4 {op, a, b, imm(double)}, children precede parents
 *
 * Item coordinates: item ordinal k = k_0 + dims[0]*(k_1 + dims[1]*k_2), first
 * coordinate fastest — the order Iterators.product gives at transform.jl:445.
 */
#ifndef IEM_BLOB_H
#define IEM_BLOB_H

#include <stdint.h>

#define IEM_BLOB_MAGIC 0x31424f4c424d4549LL /* "IEMBLOB1" */
#define IEM_BLOB_VERSION 1
#define IEM_HDR_WORDS 14
#define IEM_ARR_WORDS 6
#define IEM_MAX_DIMS 3
#define IEM_MAX_IDX_TERMS 3
#define IEM_IDX_WORDS (2 + 2 * IEM_MAX_IDX_TERMS)
#define IEM_FIELD_WORDS 6
#define IEM_NODE_WORDS 4
#define IEM_TPL_FIXED_WORDS 21
#define IEM_SLAB_WORDS 8

enum { IEM_A_F64_DATA = 0, IEM_A_I64_DATA = 1, IEM_A_F64_FILL = 2, IEM_A_I64_RANGE = 3 };
enum { IEM_T_OBJ = 0, IEM_T_CON = 1 };
enum { IEM_F_AFFINE = 0, IEM_F_GATHER = 1 };

/* Node vocabulary: leaves as emitted by transform.jl:290-334 (_map_variable),
 * operators as listed in /root/reference/src/operators.jl:3-44. */
enum {
  IEM_OP_CONST = 0, /* imm                                    */
  IEM_OP_DATA = 1,  /* a = ffield id      (item data as Float64 leaf) */
  IEM_OP_PAR = 2,   /* a = idx id         theta[idx]          */
  IEM_OP_VAR = 3,   /* a = idx id         x[idx]              */
  /* binary: children a, b */
  IEM_OP_ADD = 10,
  IEM_OP_SUB = 11,
  IEM_OP_MUL = 12,
  IEM_OP_DIV = 13,
  IEM_OP_POW = 14,
  /* unary: child a */
  IEM_OP_NEG = 20,
  IEM_OP_POS = 21,
  IEM_OP_INV = 22,
  IEM_OP_SQRT = 23,
  IEM_OP_CBRT = 24,
  IEM_OP_ABS = 25,
  IEM_OP_ABS2 = 26,
  IEM_OP_EXP = 27,
  IEM_OP_EXP2 = 28,
  IEM_OP_LOG = 29,
  IEM_OP_LOG2 = 30,
  IEM_OP_LOG10 = 31,
  IEM_OP_LOG1P = 32,
  IEM_OP_SIN = 33,
  IEM_OP_COS = 34,
  IEM_OP_TAN = 35,
  IEM_OP_ASIN = 36,
  IEM_OP_ACOS = 37,
  IEM_OP_CSC = 38,
  IEM_OP_SEC = 39,
  IEM_OP_COT = 40,
  IEM_OP_ATAN = 41,
  IEM_OP_ACOT = 42,
  IEM_OP_SIND = 43,
  IEM_OP_COSD = 44,
  IEM_OP_TAND = 45,
  IEM_OP_CSCD = 46,
  IEM_OP_SECD = 47,
  IEM_OP_COTD = 48,
  IEM_OP_ATAND = 49,
  IEM_OP_ACOTD = 50,
  IEM_OP_SINH = 51,
  IEM_OP_COSH = 52,
  IEM_OP_TANH = 53,
  IEM_OP_CSCH = 54,
  IEM_OP_SECH = 55,
  IEM_OP_COTH = 56,
  IEM_OP_ATANH = 57,
  IEM_OP_ACOTH = 58,
  IEM_OP_UNARY_END = 59
};

#define IEM_OP_IS_BINARY(op) ((op) >= IEM_OP_ADD && (op) <= IEM_OP_POW)
#define IEM_OP_IS_UNARY(op) ((op) >= IEM_OP_NEG && (op) < IEM_OP_UNARY_END)

#endif /* IEM_BLOB_H */
