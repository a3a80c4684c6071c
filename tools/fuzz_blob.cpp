// fuzz_blob.cpp — sanitizer harness for the blob parser / validator / generator (host only).
// The C-ABI accepts blobs from foreign producers (the Julia writer), so a malformed blob must
// come back as an error code, never as a crash or an out-of-bounds read.  Build with
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 -w -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
//       -Iinclude tools/fuzz_blob.cpp infiniteexamodels.jl_amd/csrc/iem_api.cpp \
//       infiniteexamodels.jl_amd/csrc/iem_codegen.cpp -L/opt/rocm/lib -lamdhip64 -lhiprtc -ldl -o /tmp/fuzz_blob
// and feed it the files written by `python tools/fuzz_blob_gen.py SEED N DIR` (tools/fuzz_blob.sh
// does both).  tests/test_blob_fuzz.py runs the same mutations against the shipped library.
#include "iem.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <fstream>
#include <chrono>
int main(int argc, char **argv) {
  int ok = 0, rej = 0;
  for (int i = 1; i < argc; ++i) {
    std::ifstream f(argv[i], std::ios::binary);
    std::vector<char> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    fprintf(stderr, "%s\n", argv[i]);
    auto t0 = std::chrono::steady_clock::now();
    char *src = nullptr, *plan = nullptr; uint64_t key; int64_t *r = nullptr, *c = nullptr, nnz = 0;
    int rc1 = iem_emit_source(b.data(), b.size(), &src, &key);
    int rc2 = iem_emit_launch_plan(b.data(), b.size(), &plan);
    int rc3 = iem_blob_hess_structure(b.data(), b.size(), 1, &r, &c, &nnz);
    if (src) iem_free(src); if (plan) iem_free(plan); if (r) iem_free(r); if (c) iem_free(c);
    for (int rw = 0; rw < 3; ++rw) {   // the window cut of iem_create_sharded: slab table, re-based indices, serialiser, re-parse
      static const int ranks[3] = {0, 1, 2}, worlds[3] = {2, 2, 3};
      void *lb = nullptr; size_t ln = 0; iem_shard_t info; int64_t *vm = nullptr; uint8_t *vf = nullptr; iem_shard_template_t *tp = nullptr;
      int64_t *its = nullptr;
      if (iem_shard_blob(b.data(), b.size(), 1, ranks[rw], worlds[rw], &lb, &ln, &info, &vm, &vf, &tp, &its) == 0) {
        char *s2 = nullptr;
        iem_emit_source(lb, ln, &s2, &key);
        if (s2) iem_free(s2);
        iem_free(lb); iem_free(vm); iem_free(vf); iem_free(tp); iem_free(its);
      }
    }
    (rc1 || rc2 || rc3) ? ++rej : ++ok;
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (dt > 1) fprintf(stderr, "  SLOW %.1fs\n", dt);
  }
  printf("ok %d rejected %d\n", ok, rej);
}
