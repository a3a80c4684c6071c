/* c_abi_demo.c — libiem_hip.so from plain C: no Python, no torch, no Julia.
 *
 *   gcc -O2 -std=c11 -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/c_abi_demo.c \
 *       -Linfiniteexamodels.jl_amd -liem_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/infiniteexamodels.jl_amd -Wl,-rpath,/opt/rocm/lib -o c_abi_demo
 *   ./c_abi_demo model.blob x.f64 y.f64
 *
 * Reads a blob (include/iem_blob.h — e.g. written by ExaCore.to_blob() or by the Julia writer),
 * an evaluation point x and multipliers y (raw little-endian doubles), runs the five NLPModels
 * calls the reference's solvers make every iteration (ext/InfiniteExaModelsIpopt.jl:48-60) on
 * device arrays it owns, then one KKT solve with the chain solver (iem_kkt_*), and prints one line per call: name, length,
 * sum, sum of squares — tests/test_c_abi_demo.py compares them with the oracle (the solve: with SciPy's sparse LU).   */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include "iem.h"

#define IEM(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, iem_last_error()); return 1; } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

static void *slurp(const char *path, size_t *n) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(1); }
  fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
  void *p = malloc(*n ? *n : 1);
  if (fread(p, 1, *n, f) != *n) { perror(path); exit(1); }
  fclose(f);
  return p;
}

static int report(const char *name, const double *d, long long n) {
  double *h = (double *)malloc((size_t)(n ? n : 1) * 8), s = 0.0, q = 0.0;
  HIP(hipMemcpy(h, d, (size_t)n * 8, hipMemcpyDeviceToHost));
  for (long long i = 0; i < n; ++i) { s += h[i]; q += h[i] * h[i]; }
  printf("%s %lld %.17g %.17g\n", name, n, s, q);
  free(h);
  return 0;
}

int main(int argc, char **argv) {
  if (argc != 4) { fprintf(stderr, "usage: %s model.blob x.f64 y.f64\n", argv[0]); return 2; }
  size_t nb, nx, ny;
  void *blob = slurp(argv[1], &nb);
  double *hx = (double *)slurp(argv[2], &nx), *hy = (double *)slurp(argv[3], &ny);
  iem_model *m = NULL;
  IEM(iem_create(blob, nb, 0, &m));
  iem_meta_t meta;
  IEM(iem_meta(m, &meta));
  if (nx != (size_t)meta.nvar * 8 || ny != (size_t)meta.ncon * 8) { fprintf(stderr, "x/y length does not match the model\n"); return 2; }
  printf("meta %lld %lld %lld %lld %d\n", (long long)meta.nvar, (long long)meta.ncon, (long long)meta.nnzj, (long long)meta.nnzh, meta.n_kernels);
  double *x, *y, *g, *c, *jv, *hv;
  HIP(hipMalloc((void **)&x, nx + 8)); HIP(hipMalloc((void **)&y, ny + 8));
  HIP(hipMalloc((void **)&g, nx + 8)); HIP(hipMalloc((void **)&c, ny + 8));
  HIP(hipMalloc((void **)&jv, (size_t)meta.nnzj * 8 + 8)); HIP(hipMalloc((void **)&hv, (size_t)meta.nnzh * 8 + 8));
  HIP(hipMemcpy(x, hx, nx, hipMemcpyHostToDevice)); HIP(hipMemcpy(y, hy, ny, hipMemcpyHostToDevice));
  double f = 0.0;
  IEM(iem_obj(m, x, &f));                       /* NLPModels.obj         */
  IEM(iem_grad(m, x, g));                       /* NLPModels.grad!       */
  IEM(iem_cons(m, x, c));                       /* NLPModels.cons!       */
  IEM(iem_jac_coord(m, x, jv));                 /* NLPModels.jac_coord!  */
  IEM(iem_hess_coord(m, x, y, 1.0, hv));        /* NLPModels.hess_coord! */
  IEM(iem_synchronize(m));
  printf("obj 1 %.17g %.17g\n", f, f * f);
  if (report("grad", g, meta.nvar) || report("cons", c, meta.ncon) || report("jac", jv, meta.nnzj) || report("hess", hv, meta.nnzh)) return 1;
  int64_t *rows = (int64_t *)malloc((size_t)(meta.nnzj ? meta.nnzj : 1) * 8), *cols = (int64_t *)malloc((size_t)(meta.nnzj ? meta.nnzj : 1) * 8);
  IEM(iem_jac_structure(m, rows, cols, 1));     /* NLPModels.jac_structure!, Julia-style 1-based */
  long long sr = 0, sc = 0;
  for (int64_t i = 0; i < meta.nnzj; ++i) { sr += rows[i]; sc += cols[i]; }
  printf("jac_structure %lld %lld %lld\n", (long long)meta.nnzj, sr, sc);
  /* the solver-facing forms of the same evaluations: objective launched first and collected last (iem_obj_begin / _end),
   * jac_coord! + hess_coord! in one launch (iem_jac_hess_coord) — identical results, into fresh buffers */
  double *jv2, *hv2, f2 = 0.0;
  HIP(hipMalloc((void **)&jv2, (size_t)meta.nnzj * 8 + 8)); HIP(hipMalloc((void **)&hv2, (size_t)meta.nnzh * 8 + 8));
  IEM(iem_obj_begin(m, x));
  IEM(iem_grad(m, x, g));
  IEM(iem_cons(m, x, c));
  IEM(iem_jac_hess_coord(m, x, y, 1.0, jv2, hv2));
  IEM(iem_obj_end(m, &f2));
  IEM(iem_synchronize(m));
  printf("obj2 1 %.17g %.17g\n", f2, f2 * f2);
  if (report("jac2", jv2, meta.nnzj) || report("hess2", hv2, meta.nnzh)) return 1;
  /* one launch per solver phase: obj + cons! at a trial point (iem_eval_trial), grad! + jac_coord! + hess_coord! at the
   * accepted point (iem_eval_accepted) — identical results again */
  double *c3, *g3, f3 = 0.0;
  HIP(hipMalloc((void **)&c3, (size_t)meta.ncon * 8 + 8)); HIP(hipMalloc((void **)&g3, (size_t)meta.nvar * 8 + 8));
  IEM(iem_eval_trial(m, x, c3, &f3));
  IEM(iem_eval_accepted(m, x, y, 1.0, g3, jv2, hv2));
  IEM(iem_synchronize(m));
  printf("obj3 1 %.17g %.17g\n", f3, f3 * f3);
  if (report("cons3", c3, meta.ncon) || report("grad3", g3, meta.nvar) || report("jac3", jv2, meta.nnzj) || report("hess3", hv2, meta.nnzh)) return 1;
  /* the linear solve of a solver iteration (where the reference plugs CUDSS, README.md:36-37): the KKT matrix of this
   * point, K = [H + 0.01 I, J'; J, -1e-6 I], assembled from the two value buffers above, factorised, and K s = (grad; cons)
   * solved — skipped (printed as such) when the model's blocks are beyond the chain solver */
  iem_kkt *k = NULL;
  if (iem_kkt_create(m, 0, &k) == 0) {
    int64_t inertia[3];
    double *rhs;
    HIP(hipMalloc((void **)&rhs, nx + ny + 8));
    HIP(hipMemcpy(rhs, g, nx, hipMemcpyDeviceToDevice));
    HIP(hipMemcpy((char *)rhs + nx, c, ny, hipMemcpyDeviceToDevice));
    IEM(iem_kkt_assemble(k, hv, jv, NULL, 1e-2, 1e-6));
    IEM(iem_kkt_factor(k, inertia));
    IEM(iem_kkt_solve(k, rhs, rhs));
    IEM(iem_synchronize(m));
    printf("kkt_inertia %lld %lld %lld\n", (long long)inertia[0], (long long)inertia[1], (long long)inertia[2]);
    if (report("kkt_solution", rhs, meta.nvar + meta.ncon)) return 1;
    IEM(iem_kkt_destroy(k));
  } else {
    printf("kkt_refused %s\n", iem_last_error());
  }
  IEM(iem_destroy(m));
  return 0;
}
