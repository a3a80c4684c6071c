// iem_api.cpp — libiem_hip.so: C-ABI (include/iem.h) over the generated gfx950 kernels.
//
// Host side of the drop-in boundary: parses the transcribed model, generates and loads
// its fused kernels (code-object cache → hiprtc on a miss), keeps θ / item-data arrays
// resident in HBM, and enqueues one launch per support grid for each NLPModels call.
// There is NO CPU evaluation path in this library: without a HIP device iem_create
// fails with IEM_E_NODEVICE.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/iem.h"
#include "iem_codegen.hpp"
#include "iem_kkt_host.hpp"
#include <memory>
#include "iem_model.hpp"
#include "iem_shard.hpp"

static const char *kDeviceHeader =
#include "iem_device_h.inc"
    ;
static const char *kKktSource =
#include "iem_kkt_device_h.inc"
    ;

namespace {

thread_local std::string g_err;
iem::Options g_opt;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(IEM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                   \
  } while (0)

// Entry points run on the handle's device whatever the calling thread's current device is.
struct DevGuard {
  int prev = -1;
  bool switched = false;
  explicit DevGuard(int d) {
    if (hipGetDevice(&prev) == hipSuccess && prev != d) switched = hipSetDevice(d) == hipSuccess;
  }
  ~DevGuard() { if (switched) hipSetDevice(prev); }
};

int g_opt_poll_obj = 1;   // iem_set_option("poll_obj", 0): iem_obj always synchronises the stream

std::string contract_flag(const iem::Options &o) { return o.fp_contract ? "-ffp-contract=fast" : "-ffp-contract=off"; }

// first line carries the compile flags so that an offline build (lib.precompile) and the
// hiprtc path compile the same key with the same options
std::string full_source(const iem::Program &p, const iem::Options &o) {
  std::string s;
  s += "// iem-flags: -O3 " + contract_flag(o) + " -std=c++17\n";
  s += "#ifndef __HIPCC_RTC__\n#include <hip/hip_runtime.h>\n#endif\n";
  s += std::string("#define IEM_NT ") + (o.nt_stores ? "1" : "0") + "\n";
  s += "#define IEM_TILE " + std::to_string(p.block) + "\n";
  s += std::string("#define IEM_WIDE_STORES ") + (o.wide_stores ? "1" : "0") + "\n";
  if (o.flush32 != 2) s += "#define IEM_FLUSH32 " + std::to_string(o.flush32) + "\n";
  if (o.ablate) s += "#define IEM_ABLATE " + std::to_string(o.ablate & 1) + "  // timing experiment, results are wrong\n";
  // A program whose kernels all use one workgroup size sees the device header as it is.  Otherwise (jac_coord! /
  // hess_coord! of a large grid run wider tiles than the rest) the header's tile-dependent region is repeated per size,
  // each copy in a namespace iem_t<size> — the generator wraps the kernels of that size in the same namespace.
  std::vector<int> tiles;
  for (const iem::KernelDesc &kd : p.kernels)
    if (std::find(tiles.begin(), tiles.end(), kd.block) == tiles.end()) tiles.push_back(kd.block);
  std::sort(tiles.begin(), tiles.end());
  if (tiles.size() <= 1) {
    s += kDeviceHeader;
  } else {
    const std::string hdr(kDeviceHeader);
    const size_t a = hdr.find("// IEM-TILE-REGION-BEGIN"), b = hdr.find("// IEM-TILE-REGION-END");
    if (a == std::string::npos || b == std::string::npos || b < a) throw std::runtime_error("internal: device header without tile-region markers");
    s += hdr.substr(0, a);
    for (int t : tiles)
      s += "#undef IEM_TILE\n#define IEM_TILE " + std::to_string(t) + "\nnamespace iem_t" + std::to_string(t) + " {\n" + hdr.substr(a, b - a) + "}  // namespace iem_t" +
           std::to_string(t) + "\n";
    s += "#undef IEM_TILE\n#define IEM_TILE " + std::to_string(p.block) + "\n";
    s += hdr.substr(b);
  }
  s += "\n";
  s += p.source;
  return s;
}

std::string lib_dir() {
  Dl_info info;
  if (dladdr((void *)&iem_version, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    size_t k = p.find_last_of('/');
    if (k != std::string::npos) return p.substr(0, k);
  }
  return ".";
}

std::string cache_dir() {
  const char *e = std::getenv("IEM_KERNEL_CACHE");
  if (e && *e) return e;
  return lib_dir() + "/kernels";   // next to libiem_hip.so: where lib.precompile / build() put the offline-built objects
}

std::string key_hex(uint64_t k) {
  char b[32];
  std::snprintf(b, sizeof b, "%016llx", (unsigned long long)k);
  return b;
}

bool read_file(const std::string &path, std::vector<char> &out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  f.seekg(0, std::ios::end);
  std::streamsize n = f.tellg();
  if (n <= 0) return false;
  f.seekg(0);
  out.resize((size_t)n);
  return (bool)f.read(out.data(), n);
}

}  // namespace

struct iem_model {
  iem::Model model;
  iem::Program prog;
  iem::Options opt;     // this handle's generator options (process defaults + iem_create_opts overrides)
  int poll_obj = 1;
  int device = 0;
  hipStream_t stream = nullptr;
  hipModule_t mod = nullptr;
  std::vector<hipFunction_t> fns;
  hipFunction_t fn_struct = nullptr, fn_csr = nullptr, fn_csr32 = nullptr, fn_axis = nullptr, fn_spmv = nullptr, fn_spmv_long = nullptr;
  long long *d_axis[iem::KK_COUNT] = {};
  long long *d_gather[iem::KK_COUNT] = {};   // per scatter kind: dest | seg | perm of its plan-driven gather (iem_gather_sum_kernel)
  hipFunction_t fn_gather = nullptr;   // per scatter kind: table of its axis sums (iem_axis_sum_kernel)
  double *d_theta = nullptr, *d_partials = nullptr, *d_obj = nullptr;
  double *d_red[iem::KK_COUNT] = {};   // per scatter kind: parked shared-entry values + tickets (iem_shared_*)
  // Second code object of jac_coord!/hess_coord! with a larger LDS staging batch (lds_slots = 48: one 96-KB
  // workgroup per CU, a third of the concurrently open store streams) and the tuner that picks, per kind and
  // per OUTPUT BUFFER, whichever of the two is faster there — how a COO buffer's physical pages fall onto the HBM
  // channels decides that, and both variants write identical bytes, so the first twenty calls into a
  // buffer run ten with one, ten with the other, under HIP events, and every call is a valid evaluation.
#define IEM_TUNE_CALLS 20   // measured calls per output buffer: a block of ten per variant
  struct Alt {
    bool on = false;
    iem::Program prog;
    hipModule_t mod = nullptr;
    std::vector<hipFunction_t> fns;
    std::vector<std::vector<uint64_t>> argbuf;
    std::vector<void *> d_tables;
  } alt;
  struct Tune {
    const void *out = nullptr;   // the buffer the decision belongs to
    int calls = 0, choice = -1;  // choice: -1 undecided, 0 default, 1 alt
    hipEvent_t ev[IEM_TUNE_CALLS][2] = {};
    bool have_events = false;
  };
  struct TuneSet {               // decisions for the last four output buffers of a kind (a solver alternates between few)
    Tune slot[4];
    int next = 0;
  } tune[2];   // [0] jac_coord!, [1] hess_coord!
  // multi-GPU (iem_create_sharded): what was cut, and the mailbox the peers push into
  bool sharded = false;
  iem::ShardInfo shard;
  hipFunction_t fn_halo = nullptr, fn_reduce = nullptr, fn_fold = nullptr;
  unsigned long long *mailbox = nullptr;   // device memory, exported through HIP IPC
  size_t mailbox_words = 0;
  int mailbox_kind = 0;   // 0 none yet, 1 uncached (fine-grained) device memory, 2 plain hipMalloc
  bool connected = false;
  std::vector<unsigned long long *> peers; // peers[r]: rank r's mailbox as mapped here (peers[rank] = mailbox)
  std::vector<void *> ipc_opened;          // what hipIpcCloseMemHandle must release
  unsigned long long **d_peers = nullptr;
  long long *d_halo_src = nullptr, *d_halo_dst = nullptr, *d_shared = nullptr;
  int64_t n_shared = 0;
  double *h_obj = nullptr;   // pinned + mapped host scalar
  double *d_hobj = nullptr;  // its device address
  bool obj_armed = false;    // iem_obj_begin launched, iem_obj_end not yet called
  // a mailbox wait that timed out (iem_device.h: IemCommErr) also lands here — mapped pinned host memory the host
  // synchronisation points read without a copy (h_obj + 1)
  volatile unsigned long long *h_status = nullptr;
  unsigned long long *d_hstatus = nullptr;
  // Asynchronous halo exchange (iem_halo_exchange_async): nothing is launched for it.  The exchange is DEFERRED and rides
  // on the first evaluation launch that takes the same x and cannot touch a halo entry of it: one extra leading workgroup
  // of that kernel (iem_halo_wg, through the kernel argument `comm` = d_comm) sends, receives and writes the halo entries
  // while the kernel's other workgroups evaluate; a launch that CAN touch a halo entry first gets the stand-alone
  // exchange kernel in front of it.  (A comm stream + events was measured: two cross-stream dependencies cost more —
  // ~8 us per step — than the 3-4 us stand-alone kernel they were meant to hide; tools/halo_overlap_probe.py.)
  bool halo_deferred = false;
  double *halo_vec = nullptr;
  void *d_comm = nullptr;        // IemHaloArgs in device memory (x unused: the carrier kernel passes its own)
  bool reads_halo_x[iem::KK_LAST + 1] = {}, reads_halo_v[iem::KK_LAST + 1] = {}, carrier[iem::KK_LAST + 1] = {};
  uint64_t nonce = 0;
  // chain KKT solver (iem_kkt_chain_*): one code object per (block size, border size)
  struct KktMod { unsigned elim_wg = 64; int elim_bpw = 1; hipModule_t mod = nullptr; hipFunction_t elim = nullptr, upd = nullptr, fwd = nullptr, bwd = nullptr, gather = nullptr, move = nullptr, colsum = nullptr, hub_z = nullptr, hub_widen = nullptr, hub_mask = nullptr, hub_ety = nullptr, hub_ex = nullptr, hub_leaf = nullptr, hub_diagmax = nullptr, fz = nullptr, fs = nullptr, bw = nullptr; int solve_bpw = 0; };
  std::map<std::pair<int, int>, KktMod> kkt_mods;
  std::map<int, void *> d_arrays;  // model array id -> device copy
  std::vector<std::vector<uint64_t>> argbuf;  // per kernel: launch argument block; only the six head words change per call
  std::vector<void *> d_tables;    // per kernel: device copy of {ip, dp, fa, ia} when they do not fit the argument block
  std::vector<double> theta_host;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool jit = false;
  std::vector<std::pair<int64_t, int64_t>> grad_zero;  // [lo, hi) ranges of g the kernels do not overwrite
  std::vector<std::pair<int64_t, int64_t>> zero_ranges[iem::KK_COUNT];
};

namespace {

int upload_array(iem_model *m, int id, bool as_int) {
  if (m->d_arrays.count(id)) return IEM_OK;
  const iem::ArrayDesc &a = m->model.arrs[id];
  void *d = nullptr;
  size_t bytes = (size_t)std::max<int64_t>(a.n, 1) * 8;
  HIP_TRY(hipMalloc(&d, bytes));
  if (a.kind == IEM_A_F64_DATA || a.kind == IEM_A_I64_DATA) {
    bool src_int = a.kind == IEM_A_I64_DATA;
    if (src_int == as_int) {
      HIP_TRY(hipMemcpy(d, a.data, (size_t)a.n * 8, hipMemcpyHostToDevice));
    } else if (as_int) {
      std::vector<int64_t> tmp(a.n);
      for (int64_t j = 0; j < a.n; ++j) tmp[j] = a.i(j);
      HIP_TRY(hipMemcpy(d, tmp.data(), (size_t)a.n * 8, hipMemcpyHostToDevice));
    } else {
      std::vector<double> tmp(a.n);
      for (int64_t j = 0; j < a.n; ++j) tmp[j] = a.f(j);
      HIP_TRY(hipMemcpy(d, tmp.data(), (size_t)a.n * 8, hipMemcpyHostToDevice));
    }
  } else if (as_int) {
    std::vector<int64_t> tmp(a.n);
    for (int64_t j = 0; j < a.n; ++j) tmp[j] = a.i(j);
    HIP_TRY(hipMemcpy(d, tmp.data(), (size_t)a.n * 8, hipMemcpyHostToDevice));
  } else {
    std::vector<double> tmp(a.n);
    for (int64_t j = 0; j < a.n; ++j) tmp[j] = a.f(j);
    HIP_TRY(hipMemcpy(d, tmp.data(), (size_t)a.n * 8, hipMemcpyHostToDevice));
  }
  m->d_arrays[id] = d;
  return IEM_OK;
}

// hiprtc build of the model's kernels for the handle's device; stores the code object in the cache
int jit_compile(iem_model *m, const std::string &src, const std::string &dir, const std::string &path, std::vector<char> &code) {
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, m->device));
  std::string arch = prop.gcnArchName;
  size_t colon = arch.find(':');
  if (colon != std::string::npos) arch = arch.substr(0, colon);
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "iem_kernels.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    return fail(IEM_E_COMPILE, "hiprtcCreateProgram failed");
  std::string archopt = "--offload-arch=" + arch;
  std::string cflag = contract_flag(m->opt);
  if (src.rfind("// iem-flags:", 0) == 0) {      // the source's own flag line decides (the offline build reads the same line)
    const size_t eol = src.find('\n'), p = src.find("-ffp-contract=");
    if (p != std::string::npos && p < eol) cflag = src.substr(p, src.find_first_of(" \n", p) - p);
  }
  const char *opts[] = {archopt.c_str(), "-O3", cflag.c_str(), "-std=c++17"};
  hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    return fail(IEM_E_COMPILE, "hiprtc: " + log);
  }
  size_t n = 0;
  hiprtcGetCodeSize(prog, &n);
  code.resize(n);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  m->jit = true;
  mkdir(dir.c_str(), 0755);
  // several ranks may compile the same key at once: private temp name, atomic rename
  const std::string tmp = path + ".tmp." + std::to_string((long long)getpid());
  std::ofstream f(tmp, std::ios::binary);
  if (f) {
    f.write(code.data(), (std::streamsize)code.size());
    f.close();
    if (std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
  }
  return IEM_OK;
}

// code object of a complete HIP source (code-object cache next to the library -> hiprtc on a miss)
int load_source(iem_model *m, const std::string &src, hipModule_t *mod) {
  const uint64_t key = iem::fnv1a64(src);
  const std::string dir = cache_dir();
  const std::string path = dir + "/iem_" + key_hex(key) + ".hsaco";
  std::vector<char> code;
  bool loaded = false;
  if (read_file(path, code)) {
    loaded = hipModuleLoadData(mod, code.data()) == hipSuccess;
    if (!loaded) { *mod = nullptr; (void)hipGetLastError(); }
  }
  if (!loaded) {
    int rc = jit_compile(m, src, dir, path, code);
    if (rc) return rc;
    HIP_TRY(hipModuleLoadData(mod, code.data()));
  }
  return IEM_OK;
}

// code object of `prog` (cache -> hiprtc on a miss) and its kernel functions
int load_program(iem_model *m, const iem::Program &prog, const iem::Options &opt, hipModule_t *mod, std::vector<hipFunction_t> *fns) {
  const std::string src = full_source(prog, opt);
  const uint64_t key = iem::fnv1a64(src);
  const std::string dir = cache_dir();
  const std::string path = dir + "/iem_" + key_hex(key) + ".hsaco";
  std::vector<char> code;
  bool loaded = false;
  if (read_file(path, code)) {
    // a cached object that does not load (truncated file, built for another architecture) is rebuilt
    loaded = hipModuleLoadData(mod, code.data()) == hipSuccess;
    if (!loaded) { *mod = nullptr; (void)hipGetLastError(); }
  }
  if (!loaded) {
    int rc = jit_compile(m, src, dir, path, code);
    if (rc) return rc;
    HIP_TRY(hipModuleLoadData(mod, code.data()));
  }
  fns->resize(prog.kernels.size());
  for (size_t k = 0; k < prog.kernels.size(); ++k)
    HIP_TRY(hipModuleGetFunction(&(*fns)[k], *mod, prog.kernels[k].name.c_str()));
  return IEM_OK;
}

int compile_or_load(iem_model *m) {
  int rc = load_program(m, m->prog, m->opt, &m->mod, &m->fns);
  if (rc) return rc;
  HIP_TRY(hipModuleGetFunction(&m->fn_struct, m->mod, "iem_structure_kernel"));
  HIP_TRY(hipModuleGetFunction(&m->fn_csr, m->mod, "iem_csr_gather_sum"));
  HIP_TRY(hipModuleGetFunction(&m->fn_csr32, m->mod, "iem_csr_gather_sum32"));
  HIP_TRY(hipModuleGetFunction(&m->fn_spmv, m->mod, "iem_csr_spmv_kernel"));
  HIP_TRY(hipModuleGetFunction(&m->fn_spmv_long, m->mod, "iem_csr_spmv_long_kernel"));
  HIP_TRY(hipModuleGetFunction(&m->fn_axis, m->mod, "iem_axis_sum_kernel"));
  HIP_TRY(hipModuleGetFunction(&m->fn_gather, m->mod, "iem_gather_sum_kernel"));
  HIP_TRY(hipModuleGetFunction(&m->fn_halo, m->mod, "iem_halo_kernel"));
  HIP_TRY(hipModuleGetFunction(&m->fn_fold, m->mod, "iem_halo_fold_kernel"));
  HIP_TRY(hipModuleGetFunction(&m->fn_reduce, m->mod, "iem_allreduce_kernel"));
  return IEM_OK;
}

// Builds the static part of a kernel's argument block once (iem_create); launching only rewrites
// the head {x, theta, y, v, out, w, aux, comm}.
void build_argbuf(iem_model *m, const iem::KernelDesc &kd, const void *d_table, std::vector<uint64_t> &buf) {
  buf.assign(13, 0);   // head: x, theta, y, v, out, w, aux, comm, p2 .. p6
  auto push_ptr = [&](const void *p) { buf.push_back((uint64_t)(uintptr_t)p); };
  if (kd.tables_in_memory) {
    const uint64_t *tb = (const uint64_t *)d_table;
    size_t nip = std::max<size_t>(1, kd.ip.size()), ndp = std::max<size_t>(1, kd.dp.size()), nfa = std::max<size_t>(1, kd.fa.size());
    push_ptr(tb); push_ptr(tb + nip); push_ptr(tb + nip + ndp); push_ptr(tb + nip + ndp + nfa);
  } else {
    for (int64_t v : kd.ip) buf.push_back((uint64_t)v);
    if (kd.ip.empty()) buf.push_back(0);
    for (double v : kd.dp) { uint64_t b; std::memcpy(&b, &v, 8); buf.push_back(b); }
    if (kd.dp.empty()) buf.push_back(0);
    for (int id : kd.fa) push_ptr(m->d_arrays[id]);
    if (kd.fa.empty()) buf.push_back(0);
    for (int id : kd.ia) push_ptr(m->d_arrays[id]);
    if (kd.ia.empty()) buf.push_back(0);
  }
}

// uploads what the program's kernels read and builds their argument blocks
int prepare_program(iem_model *m, const iem::Program &prog, std::vector<void *> &d_tables, std::vector<std::vector<uint64_t>> &argbuf) {
  int rc;
  for (const iem::KernelDesc &kd : prog.kernels) {
    for (int id : kd.fa) if ((rc = upload_array(m, id, false)) != IEM_OK) return rc;
    for (int id : kd.ia) if ((rc = upload_array(m, id, true)) != IEM_OK) return rc;
  }
  d_tables.assign(prog.kernels.size(), nullptr);
  for (size_t k = 0; k < prog.kernels.size(); ++k) {
    const iem::KernelDesc &kd = prog.kernels[k];
    if (!kd.tables_in_memory) continue;
    std::vector<uint64_t> tb;
    for (int64_t v : kd.ip) tb.push_back((uint64_t)v);
    if (kd.ip.empty()) tb.push_back(0);
    for (double v : kd.dp) { uint64_t b; std::memcpy(&b, &v, 8); tb.push_back(b); }
    if (kd.dp.empty()) tb.push_back(0);
    for (int id : kd.fa) tb.push_back((uint64_t)(uintptr_t)m->d_arrays[id]);
    if (kd.fa.empty()) tb.push_back(0);
    for (int id : kd.ia) tb.push_back((uint64_t)(uintptr_t)m->d_arrays[id]);
    if (kd.ia.empty()) tb.push_back(0);
    HIP_TRY(hipMalloc(&d_tables[k], tb.size() * 8));
    HIP_TRY(hipMemcpy(d_tables[k], tb.data(), tb.size() * 8, hipMemcpyHostToDevice));
  }
  argbuf.resize(prog.kernels.size());
  for (size_t k = 0; k < prog.kernels.size(); ++k) build_argbuf(m, prog.kernels[k], d_tables[k], argbuf[k]);
  return IEM_OK;
}

// `carry`: the launch also carries the deferred halo exchange of x — one extra leading workgroup column (iem_halo_wg)
int launch_one(iem_model *m, const iem::KernelDesc &kd, hipFunction_t fn, std::vector<uint64_t> &buf, const double *x, const double *y,
               double *out, double w, const double *v, double *aux, bool carry = false, double *p2 = nullptr, double *p3 = nullptr,
               double *p4 = nullptr, double *p5 = nullptr, double *p6 = nullptr) {
  if (kd.n_blocks <= 0) return IEM_OK;   // a support grid none of whose templates has an item
  buf[0] = (uint64_t)(uintptr_t)x; buf[1] = (uint64_t)(uintptr_t)m->d_theta; buf[2] = (uint64_t)(uintptr_t)y;
  buf[3] = (uint64_t)(uintptr_t)v; buf[4] = (uint64_t)(uintptr_t)out;
  std::memcpy(&buf[5], &w, 8);
  buf[6] = (uint64_t)(uintptr_t)aux;
  buf[7] = carry ? (uint64_t)(uintptr_t)m->d_comm : 0;
  buf[8] = (uint64_t)(uintptr_t)p2; buf[9] = (uint64_t)(uintptr_t)p3;
  buf[10] = (uint64_t)(uintptr_t)p4; buf[11] = (uint64_t)(uintptr_t)p5; buf[12] = (uint64_t)(uintptr_t)p6;
  size_t sz = buf.size() * 8;
  void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, buf.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  HIP_TRY(hipModuleLaunchKernel(fn, (unsigned)kd.grid[0] + (carry ? 1u : 0u), (unsigned)kd.grid[1], (unsigned)kd.grid[2], (unsigned)kd.block, 1, 1, 0,
                                m->stream, nullptr, cfg));
  return IEM_OK;
}

int launch(iem_model *m, size_t k, const double *x, const double *y, double *out, double w, const double *v = nullptr, double *aux = nullptr) {
  return launch_one(m, m->prog.kernels[k], m->fns[k], m->argbuf[k], x, y, out, w, v, aux);
}

// A mailbox wait timed out since the last check: the kernels poisoned what they delivered (NaN) and recorded it in
// mapped host memory.  Host synchronisation points (iem_obj / iem_obj_end / iem_synchronize) report it once and clear it.
int comm_check(iem_model *m) {
  if (!m->h_status || *m->h_status == 0) return IEM_OK;
  const unsigned long long bits = *m->h_status;
  *m->h_status = 0;
  if (m->mailbox) { const unsigned long long z = 0; (void)hipMemcpy(m->mailbox, &z, 8, hipMemcpyHostToDevice); (void)hipGetLastError(); }
  return fail(IEM_E_COMM, "a mailbox wait timed out (status bits " + std::to_string(bits) + ": 1/2 halo ack/data, 4 all-reduce, 8/16 fold ack/data): "
                          "a peer did not take part in the exchange; the halo entries / reduced values it should have delivered were set to NaN");
}

// the all-reduce runs on G workgroups, each on its own chunk of the NR doubles (one per 1 024, at most 64)
int64_t reduce_chunks(int64_t NR) { return std::min<int64_t>(64, std::max<int64_t>(1, (NR + 1023) / 1024)); }
struct HaloArgsH { double *x; unsigned long long *mine, *left, *right; const long long *src, *dst; long long NH, W, G; unsigned long long *hstatus; long long ticks; };
HaloArgsH halo_args(iem_model *m, double *d_x) {
  const iem::ShardInfo &si = m->shard;
  return HaloArgsH{d_x, m->mailbox, si.rank > 0 ? m->peers[si.rank - 1] : nullptr, si.rank + 1 < si.world ? m->peers[si.rank + 1] : nullptr,
                   m->d_halo_src, m->d_halo_dst, (long long)si.halo_doubles, (long long)si.world, (long long)reduce_chunks(1 + m->n_shared),
                   m->d_hstatus, (long long)m->opt.comm_timeout_ms * 100000LL};
}
int halo_launch(iem_model *m, double *d_x, hipStream_t stream) {
  HaloArgsH A = halo_args(m, d_x);
  size_t sz = sizeof A;
  void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  HIP_TRY(hipModuleLaunchKernel(m->fn_halo, 1, 1, 1, 256, 1, 1, 0, stream, nullptr, cfg));
  return IEM_OK;
}
int halo_flush(iem_model *m) {
  if (!m->halo_deferred) return IEM_OK;
  m->halo_deferred = false;
  return halo_launch(m, m->halo_vec, m->stream);
}
// Ordering of an evaluation call against a DEFERRED halo exchange (iem_halo_exchange_async), decided for the call about to
// be launched: a kind that can touch a halo entry of the exchanged vector (through x, or through a variable-space v) gets
// the stand-alone exchange kernel in front of it; otherwise, if it takes the same x and is a carrier kind, its first kernel
// CARRIES the exchange (*carry: one extra leading workgroup); otherwise nothing happens and the exchange stays deferred.
int halo_plan(iem_model *m, int kind, const void *x, const void *v, bool *carry) {
  *carry = false;
  if (!m->halo_deferred) return IEM_OK;
  const bool touches = (x == m->halo_vec && m->reads_halo_x[kind]) || (v && v == m->halo_vec && m->reads_halo_v[kind]);
  if (touches) return halo_flush(m);
  if (x == m->halo_vec && m->carrier[kind] && m->d_comm) { *carry = true; m->halo_deferred = false; }
  return IEM_OK;
}

int launch_kind_raw(iem_model *m, int kind, const double *x, const double *y, double *out, double w, const double *v, double *aux);

// what runs BEHIND the kernels of a scatter kind (also behind the one-launch accepted-point kernel, for grad!)
int kind_followups(iem_model *m, int kind, double *out, double *aux) {
  if (!m->prog.axis[kind].empty()) {   // sums over a non-lane axis: the rows the kernels parked -> one write per entry (iem_axis_sum_kernel)
    int64_t n0 = 1;
    for (auto &a : m->prog.axis[kind]) n0 = std::max(n0, a.n0);
    void *args[] = {(void *)&out, (void *)&aux, (void *)&m->d_axis[kind]};
    HIP_TRY(hipModuleLaunchKernel(m->fn_axis, (unsigned)((n0 + 63) / 64), (unsigned)m->prog.axis[kind].size(), 1, 256, 1, 1, 0, m->stream, args, nullptr));   // 64 lanes x 4 row groups per workgroup
  }
  if (!m->prog.gather[kind].dest.empty()) {   // what would have been float atomics: parked addends summed per entry in plan order
    const iem::Program::Gather &G = m->prog.gather[kind];
    long long n = (long long)G.dest.size();
    const double *parked = aux + G.aux_off;
    const long long *dest = m->d_gather[kind], *seg = dest + n;
    const void *perm = seg + n + 1;
    int wide = G.park_doubles >= (1LL << 32) ? 1 : 0;
    void *args[] = {(void *)&out, (void *)&parked, (void *)&dest, (void *)&seg, (void *)&perm, (void *)&n, (void *)&wide};
    HIP_TRY(hipModuleLaunchKernel(m->fn_gather, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, m->stream, args, nullptr));
  }
  return IEM_OK;
}

int launch_kind(iem_model *m, int kind, const double *x, const double *y, double *out, double w, const double *v = nullptr, double *aux = nullptr) {
  return launch_kind_raw(m, kind, x, y, out, w, v, aux);
}

int launch_kind_raw(iem_model *m, int kind, const double *x, const double *y, double *out, double w, const double *v, double *aux) {
  bool carry = false;
  int rc0 = halo_plan(m, kind, x, v, &carry);
  if (rc0) return rc0;
  for (size_t k = 0; k < m->prog.kernels.size(); ++k)
    if (m->prog.kernels[k].kind == kind) {
      if (m->prog.kernels[k].n_blocks <= 0) continue;
      int rc = launch_one(m, m->prog.kernels[k], m->fns[k], m->argbuf[k], x, y, out, w, v, aux, carry);
      if (rc) return rc;
      carry = false;   // the first kernel of the call carries it
    }
  if (carry) { m->halo_deferred = true; }   // (no kernel of the kind was launched: still pending)
  return kind_followups(m, kind, out, aux);
}

int launch_kind_alt(iem_model *m, int kind, const double *x, const double *y, double *out, double w) {
  bool carry = false;
  int rc = halo_plan(m, kind, x, nullptr, &carry);
  if (rc) return rc;
  for (size_t k = 0; k < m->alt.prog.kernels.size(); ++k)
    if (m->alt.prog.kernels[k].kind == kind) {
      if (m->alt.prog.kernels[k].n_blocks <= 0) continue;
      rc = launch_one(m, m->alt.prog.kernels[k], m->alt.fns[k], m->alt.argbuf[k], x, y, out, w, nullptr, nullptr, carry);
      if (rc) return rc;
      carry = false;
    }
  if (carry) m->halo_deferred = true;
  return IEM_OK;
}

// jac_coord! / hess_coord! through the tuner (struct Alt): the first IEM_TUNE_CALLS calls into an output buffer
// run ten with the default, then ten with the alt object, under events; then the faster variant (medians of the last
// eight of each block; alt only if > 2 % faster) is kept.
int launch_tuned(iem_model *m, int which, int kind, const double *x, const double *y, double *out, double w) {
  if (!m->alt.on) return launch_kind(m, kind, x, y, out, w);
  iem_model::TuneSet &S = m->tune[which];
  iem_model::Tune *hit = nullptr;
  for (auto &t : S.slot)
    if (t.out == out) hit = &t;
  if (!hit) {                    // a buffer not seen lately: its own measurement, in the oldest slot
    hit = &S.slot[S.next];
    S.next = (S.next + 1) % 4;
    hit->out = out; hit->calls = 0; hit->choice = -1;
  }
  iem_model::Tune &T = *hit;
  if (T.choice >= 0) return T.choice ? launch_kind_alt(m, kind, x, y, out, w) : launch_kind(m, kind, x, y, out, w);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(m->stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  if (cap != hipStreamCaptureStatusNone) return launch_kind(m, kind, x, y, out, w);   // no timing inside a graph capture
  if (T.calls < IEM_TUNE_CALLS) {
    if (!T.have_events) {
      for (auto &e : T.ev) for (auto &q : e) HIP_TRY(hipEventCreate(&q));
      T.have_events = true;
    }
    const int v = T.calls >= IEM_TUNE_CALLS / 2 ? 1 : 0;   // a block of each: alternating single launches measures the switch, not the variant
    HIP_TRY(hipEventRecord(T.ev[T.calls][0], m->stream));
    int rc = v ? launch_kind_alt(m, kind, x, y, out, w) : launch_kind(m, kind, x, y, out, w);
    HIP_TRY(hipEventRecord(T.ev[T.calls][1], m->stream));
    ++T.calls;
    return rc;
  }
  if (hipEventQuery(T.ev[IEM_TUNE_CALLS - 1][1]) == hipSuccess) {   // all have run: decide (never waits)
    // single launches time to +-5 % and the first ones of a process run cold: median of each variant's last
    // eight samples; the large batch must win by 2 %
    std::vector<float> s[2];
    for (int c = 0; c < IEM_TUNE_CALLS; ++c) {
      const int v = c >= IEM_TUNE_CALLS / 2 ? 1 : 0;
      if (c - v * (IEM_TUNE_CALLS / 2) < 2) continue;   // the first two launches of a block: cold instruction cache, the other variant's tail
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, T.ev[c][0], T.ev[c][1]) == hipSuccess) s[v].push_back(ms);
    }
    float med[2] = {0.f, 1e30f};
    for (int v = 0; v < 2; ++v)
      if (!s[v].empty()) { std::sort(s[v].begin(), s[v].end()); med[v] = s[v][s[v].size() / 2]; }
    T.choice = med[1] < 0.98f * med[0] ? 1 : 0;
    if (getenv("IEM_TUNER_LOG")) {
      fprintf(stderr, "iem tuner: kind %d buffer %p:", kind, (const void *)out);
      for (int c = 0; c < IEM_TUNE_CALLS; ++c) { float ms = -1.f; (void)hipEventElapsedTime(&ms, T.ev[c][0], T.ev[c][1]); fprintf(stderr, " %s%.4f", c >= IEM_TUNE_CALLS / 2 ? "a" : "d", ms); }
      fprintf(stderr, " medians %.4f / %.4f -> %s\n", med[0], med[1], T.choice ? "alt" : "default");
    }
    return T.choice ? launch_kind_alt(m, kind, x, y, out, w) : launch_kind(m, kind, x, y, out, w);
  }
  (void)hipGetLastError();
  return launch_kind(m, kind, x, y, out, w);
}

// host-side item index evaluation for the structure calls
struct ItemIdx {
  const iem::Model &m;
  const iem::Template &t;
  std::vector<int64_t> ifv, idx;
  ItemIdx(const iem::Model &mm, const iem::Template &tt) : m(mm), t(tt), ifv(tt.ifields.size()), idx(tt.idx.size()) {}
  void eval(int64_t k) {
    int64_t kc[3] = {k % t.dims[0], (k / t.dims[0]) % t.dims[1], k / (t.dims[0] * t.dims[1])};
    for (size_t f = 0; f < t.ifields.size(); ++f) {
      const iem::FieldDesc &fl = t.ifields[f];
      int64_t p = fl.base + fl.step[0] * kc[0] + fl.step[1] * kc[1] + fl.step[2] * kc[2];
      ifv[f] = fl.mode == IEM_F_AFFINE ? p : m.arrs[fl.arr].i(p);
    }
    for (size_t i = 0; i < t.idx.size(); ++i) {
      int64_t v = t.idx[i].c0;
      for (int j = 0; j < t.idx[i].nterms; ++j) v += t.idx[i].coef[j] * ifv[t.idx[i].field[j]];
      idx[i] = v;
    }
  }
};

// host mirrors of the device structs in iem_device.h
struct IdxDescH {
  long long c, k[3];
  int ng, garr[3];
  long long gcoef[3], gbase[3], gstep[3][3];
};
struct StructArgsH {
  long long *rows, *cols;
  const IdxDescH *ia, *ib;
  const long long *const *iarrs;
  long long dims0, dims1, n_items, o, o0, base;
  int nslots;
};

IdxDescH make_idx_desc(const iem::Template &t, int idx_id, const std::map<int, int> &iarr_slot) {
  IdxDescH d;
  std::memset(&d, 0, sizeof d);
  const iem::IdxExpr &ix = t.idx[idx_id];
  d.c = ix.c0;
  for (int j = 0; j < ix.nterms; ++j) {
    const iem::FieldDesc &f = t.ifields[ix.field[j]];
    if (f.mode == IEM_F_AFFINE) {
      d.c += ix.coef[j] * f.base;
      for (int dd = 0; dd < 3; ++dd) d.k[dd] += ix.coef[j] * f.step[dd];
    } else {
      int g = d.ng++;
      d.garr[g] = iarr_slot.at(f.arr);
      d.gcoef[g] = ix.coef[j];
      d.gbase[g] = f.base;
      for (int dd = 0; dd < 3; ++dd) d.gstep[g][dd] = f.step[dd];
    }
  }
  return d;
}

int structure_device(iem_model *m, int64_t *d_rows, int64_t *d_cols, int base, bool hess) {
  // int64 columns referenced by index expressions → device table
  std::map<int, int> slot;
  std::vector<const long long *> ptrs;
  for (const iem::Template &t : m->model.tpl)
    for (const iem::FieldDesc &f : t.ifields)
      if (f.mode == IEM_F_GATHER && !slot.count(f.arr)) {
        int rc = upload_array(m, f.arr, true);
        if (rc) return rc;
        slot[f.arr] = (int)ptrs.size();
        ptrs.push_back((const long long *)m->d_arrays[f.arr]);
      }
  // scratch descriptors live until the structure kernels have drained; freed on every exit path
  // (hipFree waits for the device)
  struct Scratch {
    std::vector<void *> v;
    void push_back(void *p) { v.push_back(p); }
    ~Scratch() { for (void *p : v) hipFree(p); }
  } to_free;
  const long long **d_ptrs = nullptr;
  HIP_TRY(hipMalloc((void **)&d_ptrs, std::max<size_t>(ptrs.size(), 1) * 8));
  to_free.push_back((void *)d_ptrs);
  if (!ptrs.empty()) HIP_TRY(hipMemcpy(d_ptrs, ptrs.data(), ptrs.size() * 8, hipMemcpyHostToDevice));
  int rc = IEM_OK;
  if (hess && !m->prog.hess_classes.empty()) {
    for (const iem::HessClass &hc : m->prog.hess_classes) {
      const int ns = (int)hc.idx_i.size();
      if (ns == 0) continue;
      std::vector<IdxDescH> a(ns), b(ns);
      for (int s = 0; s < ns; ++s) {
        a[s] = make_idx_desc(m->model.tpl[hc.idx_i[s] / 65536], hc.idx_i[s] % 65536, slot);
        b[s] = make_idx_desc(m->model.tpl[hc.idx_j[s] / 65536], hc.idx_j[s] % 65536, slot);
      }
      IdxDescH *da = nullptr, *db = nullptr;
      HIP_TRY(hipMalloc((void **)&da, sizeof(IdxDescH) * ns)); to_free.push_back(da);
      HIP_TRY(hipMalloc((void **)&db, sizeof(IdxDescH) * ns)); to_free.push_back(db);
      HIP_TRY(hipMemcpy(da, a.data(), sizeof(IdxDescH) * ns, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(db, b.data(), sizeof(IdxDescH) * ns, hipMemcpyHostToDevice));
      const iem::Template &t = m->model.tpl[hc.tpl];
      StructArgsH A{(long long *)d_rows, (long long *)d_cols, da, db, d_ptrs, t.dims[0], t.dims[1], hc.n_items, hc.o, 0, base, ns};
      size_t sz = sizeof A;
      void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
      long long total = hc.n_items * ns;
      if (total == 0) continue;   // template without items (e.g. difference rows of a one-support grid)
      hipError_t e = hipModuleLaunchKernel(m->fn_struct, (unsigned)((total + 255) / 256), 1, 1, 256, 1, 1, 0, m->stream, nullptr, cfg);
      if (e != hipSuccess) { rc = fail(IEM_E_HIP, std::string("structure kernel: ") + hipGetErrorString(e)); break; }
    }
  } else
  for (const iem::Template &t : m->model.tpl) {
    int ns = hess ? t.o2step : t.o1step;
    if (ns == 0 || (!hess && t.kind != IEM_T_CON)) continue;
    std::vector<IdxDescH> a(ns), b(ns);
    for (int s = 0; s < ns; ++s) {
      a[s] = make_idx_desc(t, hess ? t.slot2_i[s] : t.slot1_idx[s], slot);
      if (hess) b[s] = make_idx_desc(t, t.slot2_j[s], slot);
    }
    IdxDescH *da = nullptr, *db = nullptr;
    HIP_TRY(hipMalloc((void **)&da, sizeof(IdxDescH) * ns));
    to_free.push_back(da);
    HIP_TRY(hipMemcpy(da, a.data(), sizeof(IdxDescH) * ns, hipMemcpyHostToDevice));
    if (hess) {
      HIP_TRY(hipMalloc((void **)&db, sizeof(IdxDescH) * ns));
      to_free.push_back(db);
      HIP_TRY(hipMemcpy(db, b.data(), sizeof(IdxDescH) * ns, hipMemcpyHostToDevice));
    }
    StructArgsH A{(long long *)d_rows, (long long *)d_cols, da, db, d_ptrs, t.dims[0], t.dims[1], t.n_items,
                  hess ? t.o2 : t.o1, t.o0, base, ns};
    size_t sz = sizeof A;
    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    long long total = t.n_items * ns;
    if (total == 0) continue;
    hipError_t e = hipModuleLaunchKernel(m->fn_struct, (unsigned)((total + 255) / 256), 1, 1, 256, 1, 1, 0, m->stream, nullptr, cfg);
    if (e != hipSuccess) { rc = fail(IEM_E_HIP, std::string("structure kernel: ") + hipGetErrorString(e)); break; }
  }
  hipError_t e = hipStreamSynchronize(m->stream);
  if (rc) return rc;
  if (e != hipSuccess) return fail(IEM_E_HIP, std::string("structure kernel: ") + hipGetErrorString(e));
  return IEM_OK;
}

void jac_structure_host(const iem::Model &m, int64_t *rows, int64_t *cols, int base) {
  for (const iem::Template &t : m.tpl) {
    if (t.kind != IEM_T_CON || t.o1step == 0) continue;
    ItemIdx it(m, t);
    for (int64_t k = 0; k < t.n_items; ++k) {
      it.eval(k);
      int64_t o = t.o1 + (int64_t)t.o1step * k;
      for (int s = 0; s < t.o1step; ++s) {
        rows[o + s] = t.o0 + k + base;
        cols[o + s] = it.idx[t.slot1_idx[s]] - 1 + base;
      }
    }
  }
}

void hess_structure_merged_host(const iem::Model &m, const std::vector<iem::HessClass> &classes, int64_t *rows,
                                int64_t *cols, int base) {
  for (const iem::HessClass &hc : classes) {
    std::map<int, ItemIdx> evals;
    for (size_t s = 0; s < hc.idx_i.size(); ++s)
      for (int v : {hc.idx_i[s], hc.idx_j[s]})
        if (!evals.count(v / 65536)) evals.emplace(v / 65536, ItemIdx(m, m.tpl[v / 65536]));
    const int ns = (int)hc.idx_i.size();
    for (int64_t k = 0; k < hc.n_items; ++k) {
      for (auto &kv : evals) kv.second.eval(k);
      for (int s = 0; s < ns; ++s) {
        int64_t a = evals.at(hc.idx_i[s] / 65536).idx[hc.idx_i[s] % 65536];
        int64_t b = evals.at(hc.idx_j[s] / 65536).idx[hc.idx_j[s] % 65536];
        rows[hc.o + (int64_t)ns * k + s] = (a >= b ? a : b) - 1 + base;
        cols[hc.o + (int64_t)ns * k + s] = (a >= b ? b : a) - 1 + base;
      }
    }
  }
}

void hess_structure_host(const iem::Model &m, int64_t *rows, int64_t *cols, int base) {
  for (const iem::Template &t : m.tpl) {
    if (t.o2step == 0) continue;
    ItemIdx it(m, t);
    for (int64_t k = 0; k < t.n_items; ++k) {
      it.eval(k);
      int64_t o = t.o2 + (int64_t)t.o2step * k;
      for (int s = 0; s < t.o2step; ++s) {
        int64_t a = it.idx[t.slot2_i[s]], b = it.idx[t.slot2_j[s]];
        rows[o + s] = (a >= b ? a : b) - 1 + base;
        cols[o + s] = (a >= b ? b : a) - 1 + base;
      }
    }
  }
}

}  // namespace

extern "C" {
static int create_impl(const void *blob, size_t nbytes, int device, const iem_option_t *opts, int n_opts, int shard_group,
                       int rank, int world, iem_model **out);


const char *iem_last_error(void) { return g_err.c_str(); }
const char *iem_version(void) { return "iem-hip 0.1 (gfx950)"; }

void iem_free(void *p) { std::free(p); }

static int apply_option(iem::Options &o, int &poll_obj, const char *name, int64_t value) {
  if (!name) return fail(IEM_E_ARG, "null option name");
  if (std::strcmp(name, "store_mode") == 0) { o.store_mode = (int)value; return IEM_OK; }
  if (std::strcmp(name, "nt_stores") == 0) { o.nt_stores = (int)value; return IEM_OK; }
  if (std::strcmp(name, "no_fuse") == 0) { o.no_fuse = (int)value; return IEM_OK; }
  if (std::strcmp(name, "poll_obj") == 0) { poll_obj = (int)value; return IEM_OK; }
  if (std::strcmp(name, "wide_stores") == 0) { o.wide_stores = (int)value; return IEM_OK; }
  if (std::strcmp(name, "overlap") == 0) { o.overlap = (int)value; return IEM_OK; }
  if (std::strcmp(name, "xcd_remap") == 0) { o.xcd_remap = (int)value; return IEM_OK; }
  if (std::strcmp(name, "split_small") == 0) { o.split_small = (int)value; return IEM_OK; }
  if (std::strcmp(name, "fuse_groups") == 0) { o.fuse_groups = (int)value; return IEM_OK; }
  if (std::strcmp(name, "fuse_zero") == 0) { o.fuse_zero = (int)value; return IEM_OK; }
  if (std::strcmp(name, "hess_merge") == 0) { o.hess_merge = (int)value; return IEM_OK; }
  if (std::strcmp(name, "ablate") == 0) { o.ablate = (int)value; return IEM_OK; }
  if (std::strcmp(name, "block") == 0) {
    if (value != 0 && (value < 64 || value > 1024 || value % 64)) return fail(IEM_E_ARG, "block must be 0 (chosen per model) or a multiple of 64 in 64..1024");
    o.block = (int)value;
    return IEM_OK;
  }
  if (std::strcmp(name, "lds_slots") == 0) { o.lds_slots = (int)value; return IEM_OK; }
  if (std::strcmp(name, "reorder") == 0) { o.reorder = (int)value; return IEM_OK; }
  if (std::strcmp(name, "min_waves") == 0) { o.min_waves = (int)value; return IEM_OK; }
  if (std::strcmp(name, "fp_contract") == 0) { o.fp_contract = (int)value; return IEM_OK; }
  if (std::strcmp(name, "obj_wgs") == 0) {
    if (value < 1 || value > 65536) return fail(IEM_E_ARG, "obj_wgs must be in 1..65536");
    o.obj_wgs = (int)value;
    return IEM_OK;
  }
  if (std::strcmp(name, "det_shared") == 0) { o.det_shared = (int)value; return IEM_OK; }
  if (std::strcmp(name, "obj_unroll") == 0) { o.obj_unroll = (int)value; return IEM_OK; }
  if (std::strcmp(name, "flat2d") == 0) { o.flat2d = (int)value; return IEM_OK; }
  if (std::strcmp(name, "flush32") == 0) { o.flush32 = (int)value; return IEM_OK; }
  if (std::strcmp(name, "autotune") == 0) { o.autotune = (int)value; return IEM_OK; }
  if (std::strcmp(name, "pull_scatter") == 0) { o.pull_scatter = (int)value; return IEM_OK; }
  if (std::strcmp(name, "fold_colloc") == 0) { if (value < 0 || value > 2) return IEM_E_ARG; o.fold_colloc = (int)value; return IEM_OK; }
  if (std::strcmp(name, "fold_max_n") == 0) { if (value < 2 || value > 16) return IEM_E_ARG; o.fold_max_n = (int)value; return IEM_OK; }
  if (std::strcmp(name, "det_axis") == 0) { o.det_axis = (int)value; return IEM_OK; }
  if (std::strcmp(name, "det_scatter") == 0) { o.det_scatter = (int)value; return IEM_OK; }
  if (std::strcmp(name, "det_scatter_max") == 0) { o.det_scatter_max = value; return IEM_OK; }
  if (std::strcmp(name, "lazy_loads") == 0) { o.lazy_loads = (int)value; return IEM_OK; }
  if (std::strcmp(name, "lazy_min_loads") == 0) { o.lazy_min_loads = (int)value; return IEM_OK; }
  if (std::strcmp(name, "name_tag") == 0) { o.name_tag = (int)value; return IEM_OK; }
  if (std::strcmp(name, "lazy_all_kinds") == 0) { o.lazy_all_kinds = (int)value; return IEM_OK; }
  if (std::strcmp(name, "autotune_min_blocks") == 0) { o.autotune_min_blocks = (int)value; return IEM_OK; }
  if (std::strcmp(name, "big_batch_slots") == 0) { o.big_batch_slots = (int)value; return IEM_OK; }
  if (std::strcmp(name, "big_batch_jac") == 0) { o.big_batch_jac = value; return IEM_OK; }
  if (std::strcmp(name, "big_batch_hess") == 0) { o.big_batch_hess = value; return IEM_OK; }
  if (std::strcmp(name, "big_xcd") == 0) { o.big_xcd = (int)value; return IEM_OK; }
  if (std::strcmp(name, "big_tile") == 0) {
    if (value != 0 && (value < 64 || value > 1024 || value % 64)) return fail(IEM_E_ARG, "big_tile must be 0 (keep the model's tile) or a multiple of 64 in 64..1024");
    o.big_tile = (int)value;
    return IEM_OK;
  }
  if (std::strcmp(name, "pair_kernel") == 0) { o.pair_kernel = (int)value; return IEM_OK; }
  if (std::strcmp(name, "store_wait") == 0) { o.store_wait = (int)value; return IEM_OK; }
  if (std::strcmp(name, "carrier") == 0) { o.carrier = value != 0; return IEM_OK; }
  if (std::strcmp(name, "phase_kernels") == 0) { o.phase_kernels = value != 0; return IEM_OK; }
  if (std::strcmp(name, "jac_split") == 0) { if (value < 0 || value > 2) return fail(IEM_E_ARG, "jac_split must be 0, 1 or 2"); o.jac_split = (int)value; return IEM_OK; }
  if (std::strcmp(name, "jac_split_min") == 0) { if (value < 0) return fail(IEM_E_ARG, "jac_split_min must be >= 0"); o.jac_split_min = value; return IEM_OK; }
  if (std::strcmp(name, "pair_inter") == 0) { o.pair_inter = value != 0; return IEM_OK; }
  if (std::strcmp(name, "split_shift") == 0) { o.split_shift = value != 0; return IEM_OK; }
  if (std::strcmp(name, "cons_direct_2d") == 0) { o.cons_direct_2d = value != 0; return IEM_OK; }
  if (std::strcmp(name, "comm_timeout_ms") == 0) {
    if (value < 1 || value > 600000) return fail(IEM_E_ARG, "comm_timeout_ms must be in 1..600000");
    o.comm_timeout_ms = (int)value;
    return IEM_OK;
  }
  return fail(IEM_E_ARG, std::string("unknown option ") + name);
}

int iem_set_option(const char *name, int64_t value) { return apply_option(g_opt, g_opt_poll_obj, name, value); }

int iem_emit_source(const void *blob, size_t nbytes, char **out_src, uint64_t *out_key) {
  try {
    iem::Model model;
    iem::parse_blob(blob, nbytes, model);
    iem::Program p = iem::generate(model, g_opt);
    std::string s = full_source(p, g_opt);
    if (out_src) {
      *out_src = (char *)std::malloc(s.size() + 1);
      std::memcpy(*out_src, s.c_str(), s.size() + 1);
    }
    if (out_key) *out_key = iem::fnv1a64(s);
    return IEM_OK;
  } catch (const std::exception &e) {
    return fail(IEM_E_BLOB, e.what());
  }
}

int iem_emit_launch_plan(const void *blob, size_t nbytes, char **out_txt) {
  try {
    iem::Model model;
    iem::parse_blob(blob, nbytes, model);
    iem::Program p = iem::generate(model, g_opt);
    std::ostringstream os;
    os.precision(17);
    os << "partials " << p.n_partials << "\n";
    for (int kind = 0; kind < iem::KK_COUNT; ++kind)   // shared-entry reduction buffer of a scatter kind: values x workgroups (+ tickets)
      if (p.red_values[kind] > 0) os << "reduce " << kind << " " << p.red_values[kind] << " " << p.red_wgs[kind] << "\n";
    for (int kind = 0; kind < iem::KK_COUNT; ++kind) {   // whole aux buffer of a scatter kind, and its axis sums {c, k0, n0, rows, off}
      if (p.aux_doubles[kind] > 0) os << "aux " << kind << " " << p.aux_doubles[kind] << "\n";
      for (auto &a : p.axis[kind]) os << "axis " << kind << " " << a.c << " " << a.k0 << " " << a.n0 << " " << a.rows << " " << a.off << "\n";
      const iem::Program::Gather &G = p.gather[kind];
      if (!G.dest.empty()) {   // plan-driven gather: header, then the three arrays
        os << "gather " << kind << " " << G.aux_off << " " << G.park_doubles << " " << G.dest.size() << " " << G.perm.size() << "\n";
        os << "gdest"; for (int64_t v : G.dest) os << " " << v; os << "\n";
        os << "gseg"; for (int64_t v : G.seg) os << " " << v; os << "\n";
        os << "gperm"; for (int64_t v : G.perm) os << " " << v; os << "\n";
      }
    }
    for (int kind = 0; kind < iem::KK_COUNT; ++kind)   // ranges the runtime memsets before launching a scatter kind
      for (auto &z : p.zero_ranges[kind]) os << "zero " << kind << " " << z.first << " " << z.second << "\n";
    for (const iem::KernelDesc &kd : p.kernels) {
      os << "kernel " << kd.name << " kind " << kd.kind << " grid " << kd.grid[0] << " " << kd.grid[1] << " " << kd.grid[2]
         << " lds " << kd.lds_bytes << " rbytes " << kd.alg_bytes_read << " wbytes " << kd.alg_bytes_written << " block " << kd.block << " tim " << (kd.tables_in_memory ? 1 : 0) << "\n";
      os << "ip " << kd.ip.size(); for (int64_t v : kd.ip) os << " " << v; os << "\n";
      os << "dp " << kd.dp.size(); for (double v : kd.dp) { uint64_t b; std::memcpy(&b, &v, 8); os << " " << b; } os << "\n";
      os << "fa " << kd.fa.size(); for (int v : kd.fa) os << " " << v; os << "\n";
      os << "ia " << kd.ia.size(); for (int v : kd.ia) os << " " << v; os << "\n";
    }
    std::string s = os.str();
    if (out_txt) {
      *out_txt = (char *)std::malloc(s.size() + 1);
      std::memcpy(*out_txt, s.c_str(), s.size() + 1);
    }
    return IEM_OK;
  } catch (const std::exception &e) {
    return fail(IEM_E_BLOB, e.what());
  }
}

int iem_blob_array(const void *blob, size_t nbytes, int id, double **out_vals, int64_t *out_n) {
  if (!out_vals || !out_n) return fail(IEM_E_ARG, "null argument");
  try {
    iem::Model model;
    iem::parse_blob(blob, nbytes, model);
    if (id < 0 || id >= (int)model.arrs.size()) return fail(IEM_E_ARG, "array id out of range");
    const iem::ArrayDesc &a = model.arrs[id];
    double *v = (double *)std::malloc(sizeof(double) * (size_t)std::max<int64_t>(a.n, 1));
    for (int64_t j = 0; j < a.n; ++j) v[j] = a.f(j);
    *out_vals = v; *out_n = a.n;
    return IEM_OK;
  } catch (const std::exception &e) {
    return fail(IEM_E_BLOB, e.what());
  }
}

int iem_blob_hess_structure(const void *blob, size_t nbytes, int base, int64_t **out_rows, int64_t **out_cols, int64_t *out_nnz) {
  try {
    iem::Model model;
    iem::parse_blob(blob, nbytes, model);
    iem::Program p = iem::generate(model, g_opt);
    int64_t n = p.hess_classes.empty() ? model.nnzh : p.nnzh_merged;
    int64_t *r = (int64_t *)std::malloc(sizeof(int64_t) * (size_t)std::max<int64_t>(n, 1));
    int64_t *c = (int64_t *)std::malloc(sizeof(int64_t) * (size_t)std::max<int64_t>(n, 1));
    if (!p.hess_classes.empty()) hess_structure_merged_host(model, p.hess_classes, r, c, base);
    else hess_structure_host(model, r, c, base);
    *out_rows = r; *out_cols = c; *out_nnz = n;
    return IEM_OK;
  } catch (const std::exception &e) {
    return fail(IEM_E_BLOB, e.what());
  }
}

int iem_create(const void *blob, size_t nbytes, int device, iem_model **out) {
  return iem_create_opts(blob, nbytes, device, nullptr, 0, out);
}

int iem_create_opts(const void *blob, size_t nbytes, int device, const iem_option_t *opts, int n_opts, iem_model **out) {
  return create_impl(blob, nbytes, device, opts, n_opts, 0, 0, 1, out);
}

int iem_create_sharded(const void *blob, size_t nbytes, int device, int group, int rank, int world,
                       const iem_option_t *opts, int n_opts, iem_model **out) {
  if (group < 1) return fail(IEM_E_ARG, "iem_create_sharded: group must be >= 1");
  return create_impl(blob, nbytes, device, opts, n_opts, group, rank, world, out);
}

static int create_impl(const void *blob, size_t nbytes, int device, const iem_option_t *opts, int n_opts, int shard_group,
                       int rank, int world, iem_model **out) {
  if (!blob || !out || n_opts < 0 || (n_opts && !opts)) return fail(IEM_E_ARG, "null argument");
  *out = nullptr;
  // the handle's options: the process defaults (iem_set_option) with this call's overrides on top
  iem::Options hopt = g_opt;
  int hpoll = g_opt_poll_obj;
  for (int i = 0; i < n_opts; ++i) {
    int rc = apply_option(hopt, hpoll, opts[i].name, opts[i].value);
    if (rc) return rc;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(IEM_E_NODEVICE, "no HIP device visible: libiem_hip has no CPU path");
  if (device < 0 || device >= ndev) return fail(IEM_E_ARG, "device ordinal out of range");
  iem_model *m = new iem_model();
  m->device = device;
  m->opt = hopt;
  m->poll_obj = hpoll;
  try {
    iem::parse_blob(blob, nbytes, m->model);
    if (shard_group > 0) {   // cut this rank's window out of the global model (iem_shard.hpp)
      iem::shard_model(m->model, shard_group, rank, world, m->shard);
      m->sharded = true;
      m->opt.carrier = 1;    // only a sharded handle's kernels carry the halo-exchange prologue (iem_halo_exchange_async)
      // continue from the shard's own blob — byte for byte what iem_shard_blob hands out — so that the
      // generated source (hence the code-object cache key) is the one an offline build of that blob gets,
      // and the global arrays nothing references any more (x0/lvar/uvar of the whole model) are released
      std::vector<int64_t> local = iem::serialize_model(m->model);
      m->model = iem::Model();
      iem::parse_blob(local.data(), local.size() * 8, m->model);
    }
    m->prog = iem::generate(m->model, m->opt);
  } catch (const std::exception &e) {
    delete m;
    return fail(IEM_E_BLOB, e.what());
  }
  int rc = IEM_OK;
  auto bail = [&](int code) { iem_destroy(m); return code; };
  if (hipSetDevice(device) != hipSuccess) return bail(fail(IEM_E_HIP, "hipSetDevice failed"));
  if ((rc = compile_or_load(m)) != IEM_OK) return bail(rc);
  const iem::Model &M = m->model;
  m->theta_host.resize((size_t)std::max<int64_t>(M.npar, 1));
  for (int64_t i = 0; i < M.npar; ++i) m->theta_host[i] = M.arrs[M.arr_theta].f(i);
  if (hipMalloc((void **)&m->d_theta, m->theta_host.size() * 8) != hipSuccess) return bail(fail(IEM_E_HIP, "hipMalloc theta"));
  if (hipMemcpy(m->d_theta, m->theta_host.data(), m->theta_host.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
    return bail(fail(IEM_E_HIP, "upload theta"));
  // partials + the ticket counters behind them (iem_block_partial: 1 top + one per 32 workgroups),
  // zeroed once — the workgroups that complete a count reset it
  {
    const size_t np = (size_t)std::max<int64_t>(m->prog.n_partials, 1);
    const size_t words = np + 1 + (np + 31) / 32;
    if (hipMalloc((void **)&m->d_partials, words * 8) != hipSuccess || hipMemset(m->d_partials, 0, words * 8) != hipSuccess)
      return bail(fail(IEM_E_HIP, "hipMalloc partials"));
  }
  for (int kind = 0; kind < iem::KK_COUNT; ++kind) {
    // aux buffer of a scatter kind: shared-entry values x workgroups + ticket words (zeroed once), then the rows of its axis sums
    const size_t words = (size_t)m->prog.aux_doubles[kind];
    if (words == 0) continue;
    if (hipMalloc((void **)&m->d_red[kind], words * 8) != hipSuccess || hipMemset(m->d_red[kind], 0, words * 8) != hipSuccess)
      return bail(fail(IEM_E_HIP, "hipMalloc reduction buffer"));
    if (!m->prog.gather[kind].dest.empty()) {
      const iem::Program::Gather &G = m->prog.gather[kind];
      const size_t nd = G.dest.size(), np = G.perm.size();
      const bool wide = G.park_doubles >= (1LL << 32);   // parked positions fit 32 bits otherwise: half the plan traffic
      std::vector<uint32_t> p32;
      if (!wide) { p32.resize(np); for (size_t k = 0; k < np; ++k) p32[k] = (uint32_t)G.perm[k]; }
      if (hipMalloc((void **)&m->d_gather[kind], (2 * nd + 1) * 8 + np * (wide ? 8 : 4)) != hipSuccess ||
          hipMemcpy(m->d_gather[kind], G.dest.data(), nd * 8, hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(m->d_gather[kind] + nd, G.seg.data(), (nd + 1) * 8, hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(m->d_gather[kind] + 2 * nd + 1, wide ? (const void *)G.perm.data() : (const void *)p32.data(), np * (wide ? 8 : 4), hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(IEM_E_HIP, "hipMalloc gather plan"));
    }
    if (!m->prog.axis[kind].empty()) {
      std::vector<long long> tab;
      for (auto &a : m->prog.axis[kind]) { tab.push_back(a.c); tab.push_back(a.k0); tab.push_back(a.n0); tab.push_back(a.rows); tab.push_back(a.off); }
      if (hipMalloc((void **)&m->d_axis[kind], tab.size() * 8) != hipSuccess ||
          hipMemcpy(m->d_axis[kind], tab.data(), tab.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(IEM_E_HIP, "hipMalloc axis-sum table"));
    }
  }
  if (hipMalloc((void **)&m->d_obj, 8) != hipSuccess) return bail(fail(IEM_E_HIP, "hipMalloc obj"));
  if (hipHostMalloc((void **)&m->h_obj, 16, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void **)&m->d_hobj, m->h_obj, 0) != hipSuccess) return bail(fail(IEM_E_HIP, "hipHostMalloc"));
  m->h_status = reinterpret_cast<volatile unsigned long long *>(m->h_obj + 1);   // second word: comm time-outs (IemCommErr::hstatus)
  m->d_hstatus = reinterpret_cast<unsigned long long *>(m->d_hobj + 1);
  *m->h_status = 0;
  if (m->sharded) {
    // which kinds can touch a halo entry of x / of a variable-space v: from the live loads of the generated kernels
    const auto &flag = m->shard.var_flag;
    std::vector<int64_t> pre(flag.size() + 1, 0);
    for (size_t i = 0; i < flag.size(); ++i) pre[i + 1] = pre[i] + ((flag[i] & 4) ? 1 : 0);
    auto hits = [&](const std::vector<std::pair<int64_t, int64_t>> &rs) {
      for (auto &r : rs) {
        const int64_t lo = std::max<int64_t>(0, r.first), hi = std::min<int64_t>((int64_t)flag.size() - 1, r.second);
        if (lo <= hi && pre[(size_t)hi + 1] - pre[(size_t)lo] > 0) return true;
      }
      return false;
    };
    for (const iem::KernelDesc &kd : m->prog.kernels) {
      if (kd.kind < 0 || kd.kind > iem::KK_LAST) continue;
      m->reads_halo_x[kd.kind] = m->reads_halo_x[kd.kind] || hits(kd.x_ranges);
      m->reads_halo_v[kd.kind] = m->reads_halo_v[kd.kind] || hits(kd.v_ranges);
    }
    // a kind can carry a deferred exchange when EVERY kernel of it has the carrier prologue (a phase kernel whose grad! member
    // reduces shared entries has none: one extra workgroup there would shift every tile)
    bool any[iem::KK_LAST + 1] = {}, all[iem::KK_LAST + 1];
    for (bool &a : all) a = true;
    for (const iem::KernelDesc &kd : m->prog.kernels) {
      if (kd.kind < 0 || kd.kind > iem::KK_LAST) continue;
      any[kd.kind] = true; all[kd.kind] = all[kd.kind] && kd.carries;
    }
    for (int k = 0; k <= iem::KK_LAST; ++k) m->carrier[k] = any[k] && all[k];
  }
  if ((rc = prepare_program(m, m->prog, m->d_tables, m->argbuf)) != IEM_OK) return bail(rc);
  // second code object for the tuner: only for block-store models with a large jac/hess grid (below ~2e5 supports
  // the larger batch loses), never for the experiment knobs
  if (m->opt.autotune && m->opt.store_mode == 2 && m->opt.lds_slots < 48 && !m->opt.no_fuse && !m->opt.ablate) {
    int64_t big = 0;
    for (const iem::KernelDesc &kd : m->prog.kernels)
      if (kd.kind == iem::KK_JAC || kd.kind == iem::KK_HESS) big = std::max(big, kd.n_blocks);
    if (big >= m->opt.autotune_min_blocks) {
      iem::Options ob = m->opt;
      ob.lds_slots = 48;
      ob.name_tag = 48;
      try {
        m->alt.prog = iem::generate(m->model, ob);
      } catch (const std::exception &e) {
        return bail(fail(IEM_E_BLOB, e.what()));
      }
      if ((rc = load_program(m, m->alt.prog, ob, &m->alt.mod, &m->alt.fns)) != IEM_OK) return bail(rc);
      if ((rc = prepare_program(m, m->alt.prog, m->alt.d_tables, m->alt.argbuf)) != IEM_OK) return bail(rc);
      m->alt.on = true;
    }
  }
  for (int kind : {(int)iem::KK_GRAD, (int)iem::KK_JTPROD, (int)iem::KK_HPROD}) {
    // ranges the kernels of the kind do not overwrite completely.  Neighbouring ranges separated by a SHORT
    // fully-overwritten stretch are zeroed as one (what lies between is written afterwards, on the same
    // stream): one memset launch instead of two around pandemic's u(t) slab (grad! 11.5 -> 6 us of memsets)
    auto &zr = m->zero_ranges[kind];
    for (auto &z : m->prog.zero_ranges[kind]) {
      // the gap is measured against the two NEIGHBOURING ranges (never the whole span so far) and capped: a memset
      // never swallows more than 64 K doubles that a kernel is about to overwrite anyway
      const int64_t near = zr.empty() ? 0 : std::min<int64_t>(65536, std::max<int64_t>(8192, ((zr.back().second - zr.back().first) + (z.second - z.first)) / 16));
      if (!zr.empty() && z.first - zr.back().second <= near) zr.back().second = z.second;
      else zr.push_back(z);
    }
  }
  m->grad_zero = m->zero_ranges[iem::KK_GRAD];
  if (hipEventCreate(&m->ev0) != hipSuccess || hipEventCreate(&m->ev1) != hipSuccess) return bail(fail(IEM_E_HIP, "hipEventCreate"));
  *out = m;
  return IEM_OK;
}

int iem_destroy(iem_model *m) {
  if (!m) return IEM_OK;
  DevGuard dg_(m->device);
  if (m->d_theta) hipFree(m->d_theta);
  if (m->d_partials) hipFree(m->d_partials);
  if (m->d_obj) hipFree(m->d_obj);
  for (double *r : m->d_red) if (r) hipFree(r);
  for (long long *r : m->d_axis) if (r) hipFree(r);
  for (long long *r : m->d_gather) if (r) hipFree(r);
  for (void *p : m->ipc_opened) hipIpcCloseMemHandle(p);
  if (m->mailbox) hipFree(m->mailbox);
  if (m->d_peers) hipFree(m->d_peers);
  if (m->d_halo_src) hipFree(m->d_halo_src);
  if (m->d_halo_dst) hipFree(m->d_halo_dst);
  if (m->d_shared) hipFree(m->d_shared);
  if (m->h_obj) hipHostFree(m->h_obj);
  if (m->d_comm) hipFree(m->d_comm);
  for (auto &kv : m->kkt_mods) if (kv.second.mod) hipModuleUnload(kv.second.mod);
  for (auto &kv : m->d_arrays) hipFree(kv.second);
  for (void *t : m->d_tables) if (t) hipFree(t);
  for (void *t : m->alt.d_tables) if (t) hipFree(t);
  if (m->alt.mod) hipModuleUnload(m->alt.mod);
  for (auto &S : m->tune) for (auto &T : S.slot) if (T.have_events) for (auto &e : T.ev) for (auto &q : e) hipEventDestroy(q);
  if (m->ev0) hipEventDestroy(m->ev0);
  if (m->ev1) hipEventDestroy(m->ev1);
  if (m->mod) hipModuleUnload(m->mod);
  delete m;
  return IEM_OK;
}

int iem_meta(const iem_model *m, iem_meta_t *out) {
  if (!m || !out) return fail(IEM_E_ARG, "null argument");
  out->nvar = m->model.nvar; out->ncon = m->model.ncon; out->npar = m->model.npar;
  out->nnzj = m->model.nnzj;
  out->nnzh = m->prog.hess_classes.empty() ? m->model.nnzh : m->prog.nnzh_merged;
  out->n_templates = (int64_t)m->model.tpl.size();
  out->minimize = m->model.minimize;
  out->n_kernels = (int32_t)m->prog.kernels.size();
  return IEM_OK;
}

int iem_template_info(const iem_model *m, int64_t i, iem_template_info_t *out) {
  if (!m || !out || i < 0 || i >= (int64_t)m->model.tpl.size()) return fail(IEM_E_ARG, "bad template index");
  const iem::Template &t = m->model.tpl[i];
  out->kind = t.kind; out->n_items = t.n_items; out->o0 = t.o0; out->o1 = t.o1; out->o2 = t.o2;
  out->o1step = t.o1step; out->o2step = t.o2step;
  return IEM_OK;
}

int iem_kernel_info(const iem_model *m, int k, iem_kernel_info_t *out) {
  if (!m || !out || k < 0 || k >= (int)m->prog.kernels.size()) return fail(IEM_E_ARG, "bad kernel index");
  const iem::KernelDesc &kd = m->prog.kernels[k];
  std::memset(out, 0, sizeof *out);
  std::strncpy(out->name, kd.name.c_str(), sizeof(out->name) - 1);
  out->kind = kd.kind;
  for (int d = 0; d < 3; ++d) out->grid[d] = kd.grid[d];
  out->lds_bytes = kd.lds_bytes;
  out->alg_bytes_read = kd.alg_bytes_read;
  out->alg_bytes_written = kd.alg_bytes_written;
  out->jit = m->jit ? 1 : 0;
  return IEM_OK;
}

int iem_get_host(const iem_model *m, int which, double *h_out) {
  if (!m || !h_out) return fail(IEM_E_ARG, "null argument");
  const iem::Model &M = m->model;
  switch (which) {
    case IEM_X0: case IEM_LVAR: case IEM_UVAR: {
      const iem::ArrayDesc &a = M.arrs[which == IEM_X0 ? M.arr_x0 : which == IEM_LVAR ? M.arr_lvar : M.arr_uvar];
      for (int64_t i = 0; i < M.nvar; ++i) h_out[i] = a.f(i);
      return IEM_OK;
    }
    case IEM_LCON: case IEM_UCON:
      for (const iem::Template &t : M.tpl) {
        if (t.kind != IEM_T_CON) continue;
        int mode = which == IEM_LCON ? t.lmode : t.umode;
        for (int64_t k = 0; k < t.n_items; ++k)
          h_out[t.o0 + k] = mode ? M.arrs[which == IEM_LCON ? t.larr : t.uarr].f(k) : (which == IEM_LCON ? t.lval : t.uval);
      }
      return IEM_OK;
    case IEM_Y0:
      for (int64_t i = 0; i < M.ncon; ++i) h_out[i] = 0.0;
      return IEM_OK;
    case IEM_THETA:
      std::memcpy(h_out, m->theta_host.data(), (size_t)M.npar * 8);
      return IEM_OK;
  }
  return fail(IEM_E_ARG, "unknown array selector");
}

int iem_set_stream(iem_model *m, void *hip_stream) {
  if (!m) return fail(IEM_E_ARG, "null handle");
  m->stream = (hipStream_t)hip_stream;
  return IEM_OK;
}

int iem_synchronize(iem_model *m) {
  if (!m) return fail(IEM_E_ARG, "null handle");
  DevGuard dg_(m->device);
  { int rc = halo_flush(m); if (rc) return rc; }   // a deferred exchange belongs to what the caller waits for
  HIP_TRY(hipStreamSynchronize(m->stream));
  return comm_check(m);
}

int iem_set_parameter(iem_model *m, int64_t off, int64_t len, const double *h_vals) {
  if (!m || !h_vals) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  if (off < 0 || len < 0 || off + len > m->model.npar) return fail(IEM_E_ARG, "parameter range out of bounds");
  std::memcpy(m->theta_host.data() + off, h_vals, (size_t)len * 8);
  HIP_TRY(hipMemcpyAsync(m->d_theta + off, m->theta_host.data() + off, (size_t)len * 8, hipMemcpyHostToDevice, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return IEM_OK;
}

int iem_obj_device(iem_model *m, const double *d_x, double *d_out) {
  if (!m || !d_x || !d_out) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  if (m->prog.n_partials == 0) {   // no objective template: f = 0
    HIP_TRY(hipMemsetAsync(d_out, 0, 8, m->stream));
    return IEM_OK;
  }
  // the last workgroup of the objective kernel(s) to finish writes the scalar to d_out
  return launch_kind(m, iem::KK_OBJ, d_x, nullptr, m->d_partials, 0.0, nullptr, d_out);
}

// The objective as a host scalar, in two halves (a solver that evaluates obj, grad!, cons!, jac_coord!, hess_coord! at one
// point — ext/InfiniteExaModelsIpopt.jl:48-49 — needs the VALUE only after the five launches are enqueued):
//   iem_obj_begin   arms the mapped host slot with a sentinel NaN and enqueues the objective kernel; returns at once
//   iem_obj_end     waits for the slot (polling, ~200 us, then a stream synchronise) and returns the value
// The last workgroup writes the scalar straight into mapped pinned host memory; the 8-byte store is atomic, so the first
// value that is not the sentinel is the result.  iem_obj = begin + end (the host round trip of one launch: ~14 us).
static const uint64_t kObjSentinel = 0x7ff8dead0bad0b1eULL;

int iem_obj_begin(iem_model *m, const double *d_x) {
  if (!m || !d_x) return fail(IEM_E_ARG, "null argument");
  if (m->obj_armed) return fail(IEM_E_ARG, "iem_obj_begin: the previous iem_obj_begin has not been collected (iem_obj_end)");
  if (m->prog.n_partials == 0) { m->obj_armed = true; return IEM_OK; }
  *reinterpret_cast<volatile uint64_t *>(m->h_obj) = kObjSentinel;
  int rc = iem_obj_device(m, d_x, m->d_hobj);
  if (rc) return rc;
  m->obj_armed = true;
  return IEM_OK;
}

int iem_obj_end(iem_model *m, double *h_out) {
  if (!m || !h_out) return fail(IEM_E_ARG, "null argument");
  if (!m->obj_armed) return fail(IEM_E_ARG, "iem_obj_end without iem_obj_begin");
  m->obj_armed = false;
  if (m->prog.n_partials == 0) { *h_out = 0.0; return comm_check(m); }
  DevGuard dg_(m->device);
  volatile uint64_t *slot = reinterpret_cast<volatile uint64_t *>(m->h_obj);
  bool got = false;
  if (m->poll_obj) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0;; ++spin) {
      if (*slot != kObjSentinel) { got = true; break; }
      if ((spin & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
    }
  }
  if (!got) HIP_TRY(hipStreamSynchronize(m->stream));   // large models, or a result that happens to BE the sentinel
  uint64_t bits = *slot;
  std::memcpy(h_out, &bits, 8);
  return comm_check(m);
}

int iem_obj(iem_model *m, const double *d_x, double *h_out) {
  if (!m || !h_out) return fail(IEM_E_ARG, "null argument");
  int rc = iem_obj_begin(m, d_x);
  if (rc) return rc;
  return iem_obj_end(m, h_out);
}

int iem_grad(iem_model *m, const double *d_x, double *d_g) {
  if (!m || !d_x || !d_g) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  for (auto &z : m->grad_zero)   // zero only what the kernels do not overwrite completely
    HIP_TRY(hipMemsetAsync(d_g + z.first, 0, (size_t)(z.second - z.first) * 8, m->stream));
  return launch_kind(m, iem::KK_GRAD, d_x, nullptr, d_g, 0.0, nullptr, m->d_red[iem::KK_GRAD]);
}

/* NLPModels.jprod!(m, x, v, Jv) */
int iem_jprod(iem_model *m, const double *d_x, const double *d_v, double *d_Jv) {
  if (!m || !d_x || !d_v || (!d_Jv && m->model.ncon)) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  return launch_kind(m, iem::KK_JPROD, d_x, nullptr, d_Jv, 0.0, d_v);
}

/* NLPModels.jtprod!(m, x, v, Jtv) */
int iem_jtprod(iem_model *m, const double *d_x, const double *d_v, double *d_Jtv) {
  if (!m || !d_x || (!d_v && m->model.ncon) || !d_Jtv) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  for (auto &z : m->zero_ranges[iem::KK_JTPROD])
    HIP_TRY(hipMemsetAsync(d_Jtv + z.first, 0, (size_t)(z.second - z.first) * 8, m->stream));
  return launch_kind(m, iem::KK_JTPROD, d_x, nullptr, d_Jtv, 0.0, d_v, m->d_red[iem::KK_JTPROD]);
}

/* NLPModels.hprod!(m, x, y, v, Hv; obj_weight) */
int iem_hprod(iem_model *m, const double *d_x, const double *d_y, const double *d_v, double obj_weight, double *d_Hv) {
  if (!m || !d_x || (!d_y && m->model.ncon) || !d_v || !d_Hv) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  for (auto &z : m->zero_ranges[iem::KK_HPROD])
    HIP_TRY(hipMemsetAsync(d_Hv + z.first, 0, (size_t)(z.second - z.first) * 8, m->stream));
  return launch_kind(m, iem::KK_HPROD, d_x, d_y, d_Hv, obj_weight, d_v, m->d_red[iem::KK_HPROD]);
}

int iem_cons(iem_model *m, const double *d_x, double *d_c) {
  if (!m || !d_x || (!d_c && m->model.ncon)) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  return launch_kind(m, iem::KK_CONS, d_x, nullptr, d_c, 0.0);
}

int iem_jac_coord(iem_model *m, const double *d_x, double *d_vals) {
  if (!m || !d_x || (!d_vals && m->model.nnzj)) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  return launch_tuned(m, 0, iem::KK_JAC, d_x, nullptr, d_vals, 0.0);
}

int iem_hess_coord(iem_model *m, const double *d_x, const double *d_y, double obj_weight, double *d_vals) {
  if (!m || !d_x || (!d_y && m->model.ncon) || (!d_vals && m->model.nnzh)) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  return launch_tuned(m, 1, iem::KK_HESS, d_x, d_y, d_vals, obj_weight);
}

/* jac_coord!(m, x, jac) and hess_coord!(m, x, y, hess; obj_weight) in ONE launch (kernel kind KK_PAIR): the two calls are
 * independent given x and y, so their workgroups share a launch — one ramp and one drain, and on a shard-sized grid both
 * kinds are resident together.  Identical bytes to the two separate calls.  Handles without a fused kernel (option
 * "pair_kernel" = 0, or only one of the two kinds exists) make the two calls. */
int iem_jac_hess_coord(iem_model *m, const double *d_x, const double *d_y, double obj_weight, double *d_jac, double *d_hess) {
  if (!m || !d_x || (!d_y && m->model.ncon) || (!d_jac && m->model.nnzj) || (!d_hess && m->model.nnzh)) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  for (size_t k = 0; k < m->prog.kernels.size(); ++k)
    if (m->prog.kernels[k].kind == iem::KK_PAIR) {
      bool carry = false;
      int rc = halo_plan(m, iem::KK_PAIR, d_x, nullptr, &carry);
      if (rc == IEM_OK) rc = launch_one(m, m->prog.kernels[k], m->fns[k], m->argbuf[k], d_x, d_y, d_jac, obj_weight, nullptr, d_hess, carry);
      return rc;
    }
  int rc = iem_jac_coord(m, d_x, d_jac);
  if (rc == IEM_OK) rc = iem_hess_coord(m, d_x, d_y, obj_weight, d_hess);
  return rc;
}

/* One launch per solver phase (kernel kinds KK_TRIAL / KK_ACCEPTED): an interior-point solver evaluates obj + cons! at every
 * TRIAL point of its line search and grad! + jac_coord! + hess_coord! once per ACCEPTED point (the reference's solvers:
 * ext/InfiniteExaModelsMadNLP.jl:49-50,64, ext/InfiniteExaModelsIpopt.jl:48-49).  The member kinds' bodies — the very
 * functions the separate calls run — sit behind one workgroup-id dispatcher: identical bytes, one launch instead of two /
 * three (5-7 us each on the grids the reference benchmarks, ESCAPE34/run_cases_gpu.jl:89-102).  Handles without the kernel
 * (option "phase_kernels" = 0, a model without objective or without constraints, kinds of different workgroup sizes) make
 * the separate calls. */
int iem_eval_trial(iem_model *m, const double *d_x, double *d_c, double *h_obj) {
  if (!m || !d_x || (!d_c && m->model.ncon)) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  for (size_t k = 0; k < m->prog.kernels.size(); ++k)
    if (m->prog.kernels[k].kind == iem::KK_TRIAL && m->prog.n_partials > 0) {
      if (m->obj_armed) return fail(IEM_E_ARG, "iem_eval_trial: the previous iem_obj_begin / iem_eval_trial has not been collected (iem_obj_end)");
      *reinterpret_cast<volatile uint64_t *>(m->h_obj) = kObjSentinel;
      bool carry = false;
      int rc = halo_plan(m, iem::KK_TRIAL, d_x, nullptr, &carry);
      // out = c, aux = the objective scalar (mapped host memory), p2 = the objective's partials
      if (rc == IEM_OK) rc = launch_one(m, m->prog.kernels[k], m->fns[k], m->argbuf[k], d_x, nullptr, d_c, 0.0, nullptr, m->d_hobj, carry, m->d_partials);
      if (rc) return rc;
      m->obj_armed = true;
      return h_obj ? iem_obj_end(m, h_obj) : IEM_OK;
    }
  int rc = iem_obj_begin(m, d_x);
  if (rc == IEM_OK) rc = iem_cons(m, d_x, d_c);
  if (rc == IEM_OK && h_obj) rc = iem_obj_end(m, h_obj);
  return rc;
}

int iem_eval_accepted(iem_model *m, const double *d_x, const double *d_y, double obj_weight, double *d_g, double *d_jac, double *d_hess) {
  if (!m || !d_x || !d_g || (!d_y && m->model.ncon) || (!d_jac && m->model.nnzj) || (!d_hess && m->model.nnzh)) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  for (size_t k = 0; k < m->prog.kernels.size(); ++k)
    if (m->prog.kernels[k].kind == iem::KK_ACCEPTED) {
      for (auto &z : m->grad_zero)   // zero only what grad!'s bodies do not overwrite completely
        HIP_TRY(hipMemsetAsync(d_g + z.first, 0, (size_t)(z.second - z.first) * 8, m->stream));
      bool carry = false;
      int rc = halo_plan(m, iem::KK_ACCEPTED, d_x, nullptr, &carry);
      // out = jac values, aux = hess values, p2 = g, p3 = grad!'s reduction buffer
      if (rc == IEM_OK) rc = launch_one(m, m->prog.kernels[k], m->fns[k], m->argbuf[k], d_x, d_y, d_jac, obj_weight, nullptr, d_hess, carry, d_g, m->d_red[iem::KK_GRAD]);
      if (rc == IEM_OK) rc = kind_followups(m, iem::KK_GRAD, d_g, m->d_red[iem::KK_GRAD]);
      return rc;
    }
  int rc = iem_grad(m, d_x, d_g);
  if (rc == IEM_OK) rc = iem_jac_hess_coord(m, d_x, d_y, obj_weight, d_jac, d_hess);
  return rc;
}

/* obj, cons!, grad!, jac_coord!, hess_coord! of ONE point in ONE launch (kernel kind KK_ALL): the five evaluations a solver
 * makes when its first trial point is accepted.  h_obj as in iem_eval_trial. */
int iem_eval_all(iem_model *m, const double *d_x, const double *d_y, double obj_weight, double *d_c, double *d_g, double *d_jac, double *d_hess,
                 double *h_obj) {
  if (!m || !d_x || !d_g || (!d_c && m->model.ncon) || (!d_y && m->model.ncon) || (!d_jac && m->model.nnzj) || (!d_hess && m->model.nnzh))
    return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  for (size_t k = 0; k < m->prog.kernels.size(); ++k)
    if (m->prog.kernels[k].kind == iem::KK_ALL && m->prog.n_partials > 0) {
      if (m->obj_armed) return fail(IEM_E_ARG, "iem_eval_all: the previous iem_obj_begin / iem_eval_trial / iem_eval_all has not been collected (iem_obj_end)");
      for (auto &z : m->grad_zero)
        HIP_TRY(hipMemsetAsync(d_g + z.first, 0, (size_t)(z.second - z.first) * 8, m->stream));
      *reinterpret_cast<volatile uint64_t *>(m->h_obj) = kObjSentinel;
      bool carry = false;
      int rc = halo_plan(m, iem::KK_ALL, d_x, nullptr, &carry);
      if (rc == IEM_OK) rc = launch_one(m, m->prog.kernels[k], m->fns[k], m->argbuf[k], d_x, d_y, d_jac, obj_weight, nullptr, d_hess, carry, d_g, m->d_red[iem::KK_GRAD],
                                        d_c, m->d_partials, m->d_hobj);
      if (rc == IEM_OK) rc = kind_followups(m, iem::KK_GRAD, d_g, m->d_red[iem::KK_GRAD]);
      if (rc) return rc;
      m->obj_armed = true;
      return h_obj ? iem_obj_end(m, h_obj) : IEM_OK;
    }
  int rc = iem_eval_trial(m, d_x, d_c, nullptr);
  if (rc == IEM_OK) rc = iem_eval_accepted(m, d_x, d_y, obj_weight, d_g, d_jac, d_hess);
  if (rc == IEM_OK && h_obj) rc = iem_obj_end(m, h_obj);
  return rc;
}

int iem_jac_structure(iem_model *m, int64_t *h_rows, int64_t *h_cols, int base) {
  if (!m || ((!h_rows || !h_cols) && m->model.nnzj)) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  jac_structure_host(m->model, h_rows, h_cols, base);
  return IEM_OK;
}

int iem_hess_structure(iem_model *m, int64_t *h_rows, int64_t *h_cols, int base) {
  if (!m || ((!h_rows || !h_cols) && m->model.nnzh)) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  if (!m->prog.hess_classes.empty()) hess_structure_merged_host(m->model, m->prog.hess_classes, h_rows, h_cols, base);
  else hess_structure_host(m->model, h_rows, h_cols, base);
  return IEM_OK;
}

int iem_jac_structure_device(iem_model *m, int64_t *d_rows, int64_t *d_cols, int base) {
  if (!m || ((!d_rows || !d_cols) && m->model.nnzj)) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  return structure_device(m, d_rows, d_cols, base, false);
}

int iem_hess_structure_device(iem_model *m, int64_t *d_rows, int64_t *d_cols, int base) {
  if (!m || ((!d_rows || !d_cols) && m->model.nnzh)) return fail(IEM_E_ARG, "null argument");
  DevGuard dg_(m->device);
  return structure_device(m, d_rows, d_cols, base, true);
}

int iem_csr_values(iem_model *m, int64_t n_csr, const int64_t *d_seg, const int64_t *d_perm, const double *d_coo,
                   double *d_csr) {
  if (!m || n_csr < 0 || (n_csr && (!d_seg || !d_perm || !d_coo || !d_csr))) return fail(IEM_E_ARG, "bad argument");
  DevGuard dg_(m->device);
  if (n_csr == 0) return IEM_OK;
  long long n = n_csr;
  void *args[] = {(void *)&d_seg, (void *)&d_perm, (void *)&d_coo, (void *)&d_csr, (void *)&n};
  HIP_TRY(hipModuleLaunchKernel(m->fn_csr, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, m->stream, args, nullptr));
  return IEM_OK;
}

int iem_csr_values32(iem_model *m, int64_t n_csr, const uint32_t *d_seg, const uint32_t *d_perm, const double *d_coo,
                     double *d_csr) {
  if (!m || n_csr < 0 || (n_csr && (!d_seg || !d_perm || !d_coo || !d_csr))) return fail(IEM_E_ARG, "bad argument");
  DevGuard dg_(m->device);
  if (n_csr == 0) return IEM_OK;
  long long n = n_csr;
  void *args[] = {(void *)&d_seg, (void *)&d_perm, (void *)&d_coo, (void *)&d_csr, (void *)&n};
  HIP_TRY(hipModuleLaunchKernel(m->fn_csr32, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, m->stream, args, nullptr));
  return IEM_OK;
}

int iem_csr_spmv(iem_model *m, int64_t n, const int32_t *d_rowptr, const int32_t *d_colind, const double *d_vals, const double *d_x, double *d_y,
                 int64_t n_long, const int64_t *d_long_rows) {
  if (!m || n < 0 || n_long < 0 || (n && (!d_rowptr || !d_colind || !d_vals || !d_x || !d_y)) || (n_long && !d_long_rows)) return fail(IEM_E_ARG, "bad argument");
  DevGuard dg_(m->device);
  if (n == 0) return IEM_OK;
  long long nn = n;
  int skip = n_long > 0 ? IEM_SPMV_LONG_ROW : 0;
  void *args[] = {(void *)&d_rowptr, (void *)&d_colind, (void *)&d_vals, (void *)&d_x, (void *)&d_y, (void *)&nn, (void *)&skip};
  HIP_TRY(hipModuleLaunchKernel(m->fn_spmv, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, m->stream, args, nullptr));
  if (n_long > 0) {
    void *largs[] = {(void *)&d_rowptr, (void *)&d_colind, (void *)&d_vals, (void *)&d_x, (void *)&d_y, (void *)&d_long_rows};
    HIP_TRY(hipModuleLaunchKernel(m->fn_spmv_long, (unsigned)n_long, 1, 1, 256, 1, 1, 0, m->stream, largs, nullptr));
  }
  return IEM_OK;
}

/* ---- sharding (multi-GPU) ------------------------------------------------------------------- */
namespace {

void fill_shard_t(const iem::Model &M, const iem::ShardInfo &si, iem_shard_t *o) {
  std::memset(o, 0, sizeof *o);
  o->group = si.group; o->rank = si.rank; o->world = si.world; o->mailbox_kind = 0;
  o->n_global = si.n_global; o->own_lo = si.own_lo; o->own_n = si.own_n; o->halo = si.halo;
  o->halo_reach = si.halo_reach; o->halo_doubles = si.halo_doubles;
  o->nvar_global = si.nvar_global; o->ncon_global = si.ncon_global; o->nnzj_global = si.nnzj_global; o->nnzh_global = si.nnzh_global;
  o->nvar = M.nvar; o->ncon = M.ncon; o->nnzj = M.nnzj; o->nnzh = M.nnzh; o->n_templates = (int64_t)M.tpl.size();
  int64_t ns = 0;
  for (unsigned char f : si.var_flag) ns += (f & 2) ? 1 : 0;
  o->n_shared = ns;
}

void fill_shard_tpl(const iem::Model &M, const iem::ShardInfo &si, size_t i, iem_shard_template_t *o) {
  const iem::ShardTpl &st = si.tpl[i];
  const iem::Template &t = M.tpl[i];
  o->global_index = st.gindex; o->n_items = t.n_items;
  for (int d = 0; d < 3; ++d) { o->klo[d] = st.klo[d]; o->dims[d] = t.dims[d]; o->global_dims[d] = st.gdims[d]; }
  o->o0 = t.o0; o->o1 = t.o1; o->o2 = t.o2; o->global_o0 = st.go0; o->global_o1 = st.go1; o->global_o2 = st.go2;
  o->o1step = t.o1step; o->o2step = t.o2step; o->kind = t.kind;
  o->items_offset = -1;
  if (st.explicit_items) {
    int64_t off = 0;
    for (size_t j = 0; j < i; ++j) if (si.tpl[j].explicit_items) off += (int64_t)si.tpl[j].items.size();
    o->items_offset = off;
  }
}

std::vector<int64_t> shard_items(const iem::ShardInfo &si) {
  std::vector<int64_t> all;
  for (const iem::ShardTpl &st : si.tpl) if (st.explicit_items) all.insert(all.end(), st.items.begin(), st.items.end());
  return all;
}

// mailbox words: see iem_device.h
// the all-reduce runs on G workgroups, each on its own chunk of the NR doubles (one per 1 024, at most 64)
size_t mailbox_words(int64_t W, int64_t NH, int64_t NR) {
  const int64_t G = reduce_chunks(NR);
  return (size_t)(12 + G + 2 * W * G + 2 * NH + 2 * W * NR + 2 * NH);   // header, reduce flags, halo data, reduce data, fold data
}

struct CommHandle {   // what iem_comm_export writes (IEM_COMM_HANDLE_BYTES)
  hipIpcMemHandle_t ipc;
  int32_t pid, device, rank, world, kind, pad;
  int64_t words;
  uint64_t local_ptr;
  char bus[16];       // PCI bus id of the GPU the mailbox lives on ("0000:05:00.0"): device ordinals are per process
  uint64_t nonce;     // per-process random word: "same process" = same pid AND same nonce (pids repeat across PID namespaces)
};

uint64_t process_nonce() {
  static uint64_t n = 0;
  if (n == 0) {
    uint64_t v = 0;
    if (FILE *f = std::fopen("/dev/urandom", "rb")) { if (std::fread(&v, 8, 1, f) != 1) v = 0; std::fclose(f); }
    if (v == 0) v = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9e3779b97f4a7c15ULL ^ ((uint64_t)getpid() << 32);
    n = v | 1;
  }
  return n;
}
static_assert(sizeof(CommHandle) <= IEM_COMM_HANDLE_BYTES, "comm handle too large");

}  // namespace

int iem_shard_blob(const void *blob, size_t nbytes, int group, int rank, int world, void **out_blob, size_t *out_nbytes,
                   iem_shard_t *out_info, int64_t **out_var_map, uint8_t **out_var_flag, iem_shard_template_t **out_tpl,
                   int64_t **out_items) {
  if (!blob || !out_blob || !out_nbytes) return fail(IEM_E_ARG, "null argument");
  try {
    iem::Model model;
    iem::ShardInfo si;
    iem::parse_blob(blob, nbytes, model);
    iem::shard_model(model, group, rank, world, si);
    std::vector<int64_t> w = iem::serialize_model(model);
    void *b = std::malloc(w.size() * 8);
    std::memcpy(b, w.data(), w.size() * 8);
    *out_blob = b; *out_nbytes = w.size() * 8;
    if (out_info) fill_shard_t(model, si, out_info);
    if (out_var_map) {
      *out_var_map = (int64_t *)std::malloc(sizeof(int64_t) * std::max<size_t>(si.var_map.size(), 1));
      if (!si.var_map.empty()) std::memcpy(*out_var_map, si.var_map.data(), si.var_map.size() * 8);
    }
    if (out_var_flag) {
      *out_var_flag = (uint8_t *)std::malloc(std::max<size_t>(si.var_flag.size(), 1));
      if (!si.var_flag.empty()) std::memcpy(*out_var_flag, si.var_flag.data(), si.var_flag.size());
    }
    if (out_tpl) {
      *out_tpl = (iem_shard_template_t *)std::malloc(sizeof(iem_shard_template_t) * std::max<size_t>(si.tpl.size(), 1));
      for (size_t i = 0; i < si.tpl.size(); ++i) fill_shard_tpl(model, si, i, *out_tpl + i);
    }
    if (out_items) {
      const std::vector<int64_t> all = shard_items(si);
      *out_items = (int64_t *)std::malloc(sizeof(int64_t) * std::max<size_t>(all.size(), 1));
      if (!all.empty()) std::memcpy(*out_items, all.data(), all.size() * 8);
    }
    return IEM_OK;
  } catch (const std::exception &e) {
    return fail(IEM_E_BLOB, e.what());
  }
}

int iem_shard_info(const iem_model *m, iem_shard_t *out) {
  if (!m || !out) return fail(IEM_E_ARG, "null argument");
  if (!m->sharded) return fail(IEM_E_ARG, "not a sharded handle (iem_create_sharded)");
  fill_shard_t(m->model, m->shard, out);
  out->mailbox_kind = m->mailbox_kind;
  return IEM_OK;
}

int iem_shard_var_map(const iem_model *m, int64_t *h_map, uint8_t *h_flag) {
  if (!m) return fail(IEM_E_ARG, "null argument");
  if (!m->sharded) return fail(IEM_E_ARG, "not a sharded handle (iem_create_sharded)");
  if (h_map) std::memcpy(h_map, m->shard.var_map.data(), m->shard.var_map.size() * 8);
  if (h_flag) std::memcpy(h_flag, m->shard.var_flag.data(), m->shard.var_flag.size());
  return IEM_OK;
}

int iem_shard_template_info(const iem_model *m, int64_t i, iem_shard_template_t *out) {
  if (!m || !out) return fail(IEM_E_ARG, "null argument");
  if (!m->sharded) return fail(IEM_E_ARG, "not a sharded handle (iem_create_sharded)");
  if (i < 0 || i >= (int64_t)m->shard.tpl.size()) return fail(IEM_E_ARG, "bad template index");
  fill_shard_tpl(m->model, m->shard, (size_t)i, out);
  return IEM_OK;
}

int iem_shard_template_items(const iem_model *m, int64_t *h_items, int64_t *out_n) {
  if (!m || !out_n) return fail(IEM_E_ARG, "null argument");
  if (!m->sharded) return fail(IEM_E_ARG, "not a sharded handle (iem_create_sharded)");
  const std::vector<int64_t> all = shard_items(m->shard);
  *out_n = (int64_t)all.size();
  if (h_items && !all.empty()) std::memcpy(h_items, all.data(), all.size() * 8);
  return IEM_OK;
}

/* ---- mailboxes: halo exchange + the small all-reduce ------------------------------------------ */
int iem_comm_export(iem_model *m, void *out_handle) {
  if (!m || !out_handle) return fail(IEM_E_ARG, "null argument");
  if (!m->sharded) return fail(IEM_E_ARG, "not a sharded handle (iem_create_sharded)");
  DevGuard dg_(m->device);
  const iem::ShardInfo &si = m->shard;
  if (si.halo_reach > 0)
    for (int r = 0; r < si.world; ++r) {   // a stencil must not reach past the neighbouring rank
      int64_t a, b;
      iem::partition_block(si.n_global, si.world, r, a, b);
      if (b - a < si.halo_reach) return fail(IEM_E_ARG, "a rank owns fewer supports than the stencil reaches");
    }
  int64_t ns = 0;
  for (unsigned char f : si.var_flag) ns += (f & 2) ? 1 : 0;
  m->n_shared = ns;
  m->mailbox_words = mailbox_words(si.world, si.halo_doubles, 1 + ns);
  CommHandle h;
  std::memset(&h, 0, sizeof h);
  if (!m->mailbox) {
    // Peers write into this memory while a kernel here polls it: it must be FINE-GRAINED (coherent across
    // agents inside a kernel) — uncached device memory first, fine-grained next; plain hipMalloc memory is
    // the last resort (coherent only between processes sharing this GPU's L2: the one-GPU rehearsal) and
    // is reported through iem_shard_info().mailbox_kind so a multi-GPU host can refuse it.
    const unsigned flags[3] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained, 0};
    for (int attempt = 0; attempt < 3 && !m->mailbox; ++attempt) {
      void *p = nullptr;
      hipError_t e = attempt < 2 ? hipExtMallocWithFlags(&p, m->mailbox_words * 8, flags[attempt]) : hipMalloc(&p, m->mailbox_words * 8);
      if (e != hipSuccess) { (void)hipGetLastError(); continue; }
      if (hipIpcGetMemHandle(&h.ipc, p) != hipSuccess) { (void)hipGetLastError(); hipFree(p); continue; }
      m->mailbox = (unsigned long long *)p;
      m->mailbox_kind = attempt == 0 ? 1 : attempt == 1 ? 3 : 2;
    }
    if (!m->mailbox) return fail(IEM_E_HIP, "could not allocate an IPC-exportable mailbox");
    HIP_TRY(hipMemset(m->mailbox, 0, m->mailbox_words * 8));
    HIP_TRY(hipDeviceSynchronize());
  } else {
    HIP_TRY(hipIpcGetMemHandle(&h.ipc, m->mailbox));
  }
  h.pid = (int32_t)getpid(); h.device = m->device; h.rank = si.rank; h.world = si.world; h.words = (int64_t)m->mailbox_words;
  h.kind = m->mailbox_kind;
  h.local_ptr = (uint64_t)(uintptr_t)m->mailbox;
  h.nonce = process_nonce();
  {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, m->device) != hipSuccess) { (void)hipGetLastError(); bus[0] = 0; }
    std::memcpy(h.bus, bus, sizeof(h.bus) - 1);   // h was zeroed: stays terminated
  }
  std::memset(out_handle, 0, IEM_COMM_HANDLE_BYTES);
  std::memcpy(out_handle, &h, sizeof h);
  return IEM_OK;
}

int iem_comm_connect(iem_model *m, const void *all_handles) {
  if (!m || !all_handles) return fail(IEM_E_ARG, "null argument");
  if (!m->sharded || !m->mailbox) return fail(IEM_E_ARG, "iem_comm_connect: call iem_comm_export on this handle first");
  if (m->connected) return fail(IEM_E_ARG, "already connected");
  DevGuard dg_(m->device);
  const iem::ShardInfo &si = m->shard;
  m->peers.assign((size_t)si.world, nullptr);
  CommHandle mine;
  std::memcpy(&mine, (const char *)all_handles + (size_t)si.rank * IEM_COMM_HANDLE_BYTES, sizeof mine);
  {
    // One process may drive several handles (one per GPU).  More than two on ONE device cannot work: their exchange
    // kernels spin on each other's flags and the streams of one process share hardware queues from the third on, so
    // a spinning kernel starves the peer it waits for (each call would end in the bounded time-out).  Refused here.
    int same = 0;
    for (int r = 0; r < si.world; ++r) {
      CommHandle h;
      std::memcpy(&h, (const char *)all_handles + (size_t)r * IEM_COMM_HANDLE_BYTES, sizeof h);
      if (h.pid == mine.pid && h.nonce == mine.nonce && std::strncmp(h.bus, mine.bus, sizeof h.bus) == 0) ++same;
    }
    if (same > 2)
      return fail(IEM_E_ARG, "iem_comm_connect: " + std::to_string(same) + " ranks of this communicator are handles of ONE process on ONE device; "
                             "at most two can exchange (one process per GPU is the supported form)");
  }
  for (int r = 0; r < si.world; ++r) {
    CommHandle h;
    std::memcpy(&h, (const char *)all_handles + (size_t)r * IEM_COMM_HANDLE_BYTES, sizeof h);
    if (h.rank != r || h.world != si.world || h.words != (int64_t)m->mailbox_words)
      return fail(IEM_E_ARG, "iem_comm_connect: handle " + std::to_string(r) + " does not belong to this communicator");
    // a kernel polls its mailbox while peers write into it: across GPUs that is only coherent for
    // fine-grained memory — refuse plain hipMalloc mailboxes there instead of timing out at run time
    if ((h.kind == 2 || mine.kind == 2) && std::strncmp(h.bus, mine.bus, sizeof h.bus) != 0)
      return fail(IEM_E_HIP, "iem_comm_connect: this runtime gave no fine-grained IPC memory for the mailboxes; "
                             "ranks on different GPUs cannot use them — fall back to the host's collective (RCCL) for obj/grad and copy the halo");
    if (r == si.rank) { m->peers[r] = m->mailbox; continue; }
    if (h.pid == (int32_t)getpid() && h.nonce == process_nonce()) {   // same process (one process driving several handles): the pointer itself
      if ((int)h.device != m->device) {
        hipError_t e = hipDeviceEnablePeerAccess((int)h.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(IEM_E_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        (void)hipGetLastError();
      }
      m->peers[r] = (unsigned long long *)(uintptr_t)h.local_ptr;
      continue;
    }
    void *p = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&p, h.ipc, hipIpcMemLazyEnablePeerAccess));
    m->ipc_opened.push_back(p);
    m->peers[r] = (unsigned long long *)p;
  }
  HIP_TRY(hipMalloc((void **)&m->d_peers, sizeof(void *) * (size_t)si.world));
  HIP_TRY(hipMemcpy(m->d_peers, m->peers.data(), sizeof(void *) * (size_t)si.world, hipMemcpyHostToDevice));
  // halo positions: what goes to the right neighbour (my last `reach` owned supports of every sharded
  // slab), where the left neighbour's arrive (my first `reach` window entries) — one canonical order
  std::vector<long long> src, dst;
  for (const iem::HaloSeg &sg : si.segs)
    for (int64_t o = 0; o < sg.outer; ++o)
      for (int64_t j = 0; j < si.halo_reach * sg.inner; ++j) {
        src.push_back(sg.loff + o * sg.wn * sg.inner + (sg.wn - si.halo_reach) * sg.inner + j);
        dst.push_back(sg.loff + o * sg.wn * sg.inner + j);
      }
  if ((int64_t)src.size() != si.halo_doubles) return fail(IEM_E_ARG, "internal: halo size mismatch");
  if (!src.empty()) {
    HIP_TRY(hipMalloc((void **)&m->d_halo_src, src.size() * 8));
    HIP_TRY(hipMalloc((void **)&m->d_halo_dst, dst.size() * 8));
    HIP_TRY(hipMemcpy(m->d_halo_src, src.data(), src.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->d_halo_dst, dst.data(), dst.size() * 8, hipMemcpyHostToDevice));
  }
  std::vector<long long> sh;
  for (size_t i = 0; i < si.var_flag.size(); ++i) if (si.var_flag[i] & 2) sh.push_back((long long)i);
  if (!sh.empty()) {
    HIP_TRY(hipMalloc((void **)&m->d_shared, sh.size() * 8));
    HIP_TRY(hipMemcpy(m->d_shared, sh.data(), sh.size() * 8, hipMemcpyHostToDevice));
  }
  {
    HaloArgsH A = halo_args(m, nullptr);
    HIP_TRY(hipMalloc(&m->d_comm, sizeof A));
    HIP_TRY(hipMemcpy(m->d_comm, &A, sizeof A, hipMemcpyHostToDevice));
  }
  m->connected = true;
  return IEM_OK;
}

int iem_halo_exchange(iem_model *m, double *d_x) {
  if (!m || !d_x) return fail(IEM_E_ARG, "null argument");
  if (!m->connected) return fail(IEM_E_ARG, "iem_halo_exchange: not connected (iem_comm_connect)");
  const iem::ShardInfo &si = m->shard;
  if (si.halo_doubles == 0 || si.world == 1) return IEM_OK;   // no stencil crosses the shard boundary
  DevGuard dg_(m->device);
  int rc = halo_flush(m);   // one exchange at a time (the mailbox has two parity slots)
  if (rc) return rc;
  return halo_launch(m, d_x, m->stream);
}

int iem_halo_exchange_async(iem_model *m, double *d_x) {
  if (!m || !d_x) return fail(IEM_E_ARG, "null argument");
  if (!m->connected) return fail(IEM_E_ARG, "iem_halo_exchange_async: not connected (iem_comm_connect)");
  const iem::ShardInfo &si = m->shard;
  if (si.halo_doubles == 0 || si.world == 1) return IEM_OK;
  DevGuard dg_(m->device);
  int rc = halo_flush(m);   // an earlier exchange nobody carried: it goes first, in order
  if (rc) return rc;
  m->halo_deferred = true;
  m->halo_vec = d_x;
  return IEM_OK;
}

int iem_halo_wait(iem_model *m) {
  if (!m) return fail(IEM_E_ARG, "null handle");
  DevGuard dg_(m->device);
  return halo_flush(m);
}

int iem_halo_reads(const iem_model *m, int kind, int *out_x, int *out_v, int *out_carrier) {
  if (!m || kind < 0 || kind > iem::KK_LAST) return fail(IEM_E_ARG, "bad argument");
  if (out_x) *out_x = m->reads_halo_x[kind] ? 1 : 0;
  if (out_v) *out_v = m->reads_halo_v[kind] ? 1 : 0;
  if (out_carrier) *out_carrier = (m->carrier[kind] && !m->reads_halo_x[kind]) ? 1 : 0;
  return IEM_OK;
}

int iem_halo_fold(iem_model *m, double *d_vec) {
  if (!m || !d_vec) return fail(IEM_E_ARG, "null argument");
  if (!m->connected) return fail(IEM_E_ARG, "iem_halo_fold: not connected (iem_comm_connect)");
  const iem::ShardInfo &si = m->shard;
  if (si.halo_doubles == 0 || si.world == 1) return IEM_OK;
  DevGuard dg_(m->device);
  { int rc = halo_flush(m); if (rc) return rc; }
  struct { double *vec; unsigned long long *mine, *left, *right; const long long *src, *dst; long long NH, W, G, NR; unsigned long long *hstatus; long long ticks; } A = {
      d_vec, m->mailbox, si.rank > 0 ? m->peers[si.rank - 1] : nullptr, si.rank + 1 < si.world ? m->peers[si.rank + 1] : nullptr,
      m->d_halo_src, m->d_halo_dst, (long long)si.halo_doubles, (long long)si.world, (long long)reduce_chunks(1 + m->n_shared),
      (long long)(1 + m->n_shared), m->d_hstatus, (long long)m->opt.comm_timeout_ms * 100000LL};
  size_t sz = sizeof A;
  void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  HIP_TRY(hipModuleLaunchKernel(m->fn_fold, 1, 1, 1, 256, 1, 1, 0, m->stream, nullptr, cfg));
  return IEM_OK;
}

int iem_allreduce_obj_grad(iem_model *m, double *d_obj, double *d_g) {
  if (!m) return fail(IEM_E_ARG, "null argument");
  if (!m->connected) return fail(IEM_E_ARG, "iem_allreduce_obj_grad: not connected (iem_comm_connect)");
  if (!d_g && m->n_shared) return fail(IEM_E_ARG, "null gradient but the model has replicated variables");
  DevGuard dg_(m->device);
  // Every rank must issue its mailbox kernels in the SAME order.  Whether an evaluation call in between flushed a deferred
  // exchange depends on the rank (rank 0 holds no halo copies: its reads_halo is false), so a collective never overtakes a
  // deferred exchange: it goes first, as in iem_halo_fold / iem_comm_status / iem_synchronize.
  { int rc = halo_flush(m); if (rc) return rc; }
  const iem::ShardInfo &si = m->shard;
  const long long G = (long long)reduce_chunks(1 + m->n_shared);
  struct { double *obj, *g; const long long *shared; unsigned long long *const *peers; long long NR, NH, W, rank, G; unsigned long long *hstatus; long long ticks; } A = {
      d_obj, d_g, m->d_shared, m->d_peers, (long long)(1 + m->n_shared), (long long)si.halo_doubles, (long long)si.world, (long long)si.rank, G,
      m->d_hstatus, (long long)m->opt.comm_timeout_ms * 100000LL};
  size_t sz = sizeof A;
  void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  HIP_TRY(hipModuleLaunchKernel(m->fn_reduce, (unsigned)G, 1, 1, 256, 1, 1, 0, m->stream, nullptr, cfg));
  return IEM_OK;
}

int iem_comm_status(iem_model *m, int64_t *out_status) {
  if (!m || !out_status) return fail(IEM_E_ARG, "null argument");
  if (!m->mailbox) return fail(IEM_E_ARG, "no mailbox");
  DevGuard dg_(m->device);
  { int rc = halo_flush(m); if (rc) return rc; }
  HIP_TRY(hipStreamSynchronize(m->stream));
  unsigned long long st = 0;
  HIP_TRY(hipMemcpy(&st, m->mailbox, 8, hipMemcpyDeviceToHost));
  *out_status = (int64_t)st;
  return IEM_OK;
}

/* ---- chain KKT solver ------------------------------------------------------------------------------ */
namespace {
// Shape of a kkt_eliminate workgroup: waves (KKT_WMAX of csrc/iem_kkt_device.h; a wave owns every KKT_WMAX-th 16-row tile row)
// and the register budget as waves per SIMD (KKT_WPE).  The block inverse is a chain of dependent panel steps, each of which
// reads the whole row panel from LDS IN EVERY WAVE: few waves with many tiles each — and many workgroups per CU to hide the
// chain's latency — beat one wave per tile row (quadrotor_oc3, 84 x 84 blocks: 11.6 ms with four waves, 10.3 with six, 5.8
// with two; 40 x 40: 2.60 -> 2.40 with one).  With a border the products Z = D^-1 E and E' Z dominate and want the waves
// (OPF, 60 + 52: 3.6 ms with four, 4.6 with two, 7.0 with one).  profiles/r03_kkt_shape_ab.txt
struct KktKnobs { int wmax = -1, wpe = -1, rowwise = -1, wpg = -1; std::string contract = "off", defs; };
KktKnobs kkt_knobs() {
  KktKnobs k;
  const char *on = getenv("IEM_KKT_EXPERIMENTS");
  if (!on || std::strcmp(on, "1") != 0) return k;     // production: the environment cannot change kernel shape, numerics or source
  if (const char *e = getenv("IEM_KKT_WMAX")) { const int v = atoi(e); if (v >= 1 && v <= 6) k.wmax = v; }
  if (const char *e = getenv("IEM_KKT_WPE")) { const int v = atoi(e); if (v >= 0 && v <= 8) k.wpe = v; }
  if (const char *e = getenv("IEM_KKT_WPG")) { const int v = atoi(e); if (v == 1 || v == 2) k.wpg = v; }
  if (const char *e = getenv("IEM_KKT_ROWWISE")) { const int v = atoi(e); if (v >= 0 && v <= 2) k.rowwise = v; }
  if (const char *e = getenv("IEM_KKT_CONTRACT")) { if (!std::strcmp(e, "fast") || !std::strcmp(e, "on") || !std::strcmp(e, "off")) k.contract = e; }
  if (const char *e = getenv("IEM_KKT_DEFS")) {        // extra "#define NAME VALUE" lines only: [A-Za-z0-9_ #\n]
    bool ok = true;
    for (const char *c = e; *c; ++c) ok = ok && (std::isalnum((unsigned char)*c) || *c == '_' || *c == ' ' || *c == '#' || *c == '\n');
    if (ok) k.defs = e;
  }
  return k;
}
// kkt_eliminate's other form (KKT_ROWWISE of csrc/iem_kkt_device.h): lane = row, 64 / nb blocks per one-wave workgroup — blocks
// without a border that fit the lanes of a wave
// ... returns the rows a lane holds (KKT_RPL): 0 = the matrix-core panel form
int kkt_rowwise(int nb, int ne) {
  static const KktKnobs knobs = kkt_knobs();
  if (ne != 0 || nb > 48) return 0;
  if (knobs.rowwise >= 0) return knobs.rowwise;
  // two rows per lane up to 24 x 24 (hovercraft 20 x 20 at 1e5 supports: 0.74 ms against 0.84 with one row and 1.66 in the panel form); one
  // row up to 40 (40 x 40: 1.96 ms against 2.40 in the panel form, three blocks per two-wave workgroup — kkt_row_wpg; two rows of 40 leave one
  // wave per SIMD: 2.9 ms; 44 x 44 spills: 7.3 ms);
  // beyond that the panel form.  profiles/r04_kkt_rpl_ab.txt
  return nb <= 24 ? 2 : nb <= 40 ? 1 : 0;
}
// ne = -1: no border and ONE block per launch (the pivot blocks of the hubs' dense LDL', one after the other): nothing hides the
// block's own latency, so every tile row gets a wave (96 x 96: 50 -> 38 us per block; profiles/r04_kkt_leaf_shape_ab.txt)
// waves per workgroup of the lane-per-row eliminate: two when that fills the lanes better (40 lanes per block: 3 blocks on 128 lanes
// instead of 1 on 64 — the kernel is bound by the LDS reads every lane of the workgroup makes, idle or not)
int kkt_row_wpg(int nb, int ne) {
  static const KktKnobs knobs = kkt_knobs();
  const int rpl = kkt_rowwise(nb, ne);
  if (!rpl) return 1;
  if (knobs.wpg >= 1) return knobs.wpg;
  const int lpb = nb / rpl;
  return (128 / lpb) * 64 > (64 / lpb) * 128 ? 2 : 1;      // blocks per lane, two waves against one
}
void kkt_shape(int nb, int ne, int *wmax, int *wpe) {
  const int R = (nb + 15) / 16;
  if (ne < 0) { *wmax = 6; *wpe = 0; return; }
  if (ne > 0 || R <= 2) { *wmax = 4; *wpe = nb <= 48 ? 4 : 0; }
  else if (R == 3) { *wmax = 1; *wpe = 3; }
  else if (R == 4) { *wmax = 2; *wpe = 3; }
  else { *wmax = 2; *wpe = 2; }
  // experiment overrides (tools/sessions/kkt_shape_ab.sh): honoured only when IEM_KKT_EXPERIMENTS=1, and read ONCE per process —
  // the launch shape of a cached module can never drift from the KKT_T it was compiled with
  static const KktKnobs knobs = kkt_knobs();
  if (knobs.wmax >= 1) *wmax = knobs.wmax;
  if (knobs.wpe >= 0) *wpe = knobs.wpe;
}
std::string kkt_source(int nb, int ne, int nc) {
  // (fused multiply-adds would be allowed here — nothing compares these kernels bit for bit — and take 1 000 of the 2 350 FP64
  // instructions out of a 40 x 40 kkt_eliminate, but the factorisation does not get faster for it: 2.52 against 2.45 ms at 1e5
  // quadrotor supports, 3.22 against 3.54 for the bordered OPF blocks, solves 9 % slower — the panel steps wait on their
  // dependency chain, not on issue slots.  IEM_KKT_CONTRACT=fast switches them on; profiles/r03_kkt_shape_ab.txt)
  static const KktKnobs knobs = kkt_knobs();
  std::string s = std::string("// iem-flags: -O3 -ffp-contract=") + knobs.contract + " -std=c++17\n#ifndef __HIPCC_RTC__\n#include <hip/hip_runtime.h>\n#endif\n";
  s += "#define KKT_NB " + std::to_string(nb) + "\n#define KKT_NE " + std::to_string(std::max(ne, 0)) + "\n#define KKT_NC " + std::to_string(nc) + "\n";
  int wmax, wpe;
  kkt_shape(nb, ne, &wmax, &wpe);
  if (wpe > 0) s += "#define KKT_WPE " + std::to_string(wpe) + "\n";
  if (wmax != 4) s += "#define KKT_WMAX " + std::to_string(wmax) + "\n";
  if (kkt_rowwise(nb, ne)) s += "#define KKT_ROWWISE " + std::to_string(kkt_rowwise(nb, ne)) + "\n#define KKT_ROW_WPG " + std::to_string(kkt_row_wpg(nb, ne)) + "\n";
  if (!knobs.defs.empty()) s += knobs.defs + "\n";     // (experiments, IEM_KKT_EXPERIMENTS=1 only: extra #define lines)
  s += kKktSource;
  return s;
}
// LDS of kkt_eliminate: the panel buffers of the inverse, and with a border one NB x NB and two NB x NE tiles; of kkt_update:
// seven NC x NC tiles
bool kkt_fits(int nb, int ne, int nc) {
  const long long elim = 8LL * (10 * nb + 8 * (nb + 1)) + (ne > 0 ? 8LL * nb * (nb + 1) + 16LL * nb * (ne + 1) : 0) + 1024;
  const long long upd = 56LL * nc * (nc + 1) + 1024;
  return elim <= 160 * 1024 && upd <= 160 * 1024;
}
int kkt_module(iem_model *m, int nb, int ne, int nc, iem_model::KktMod **out) {
  if (nb < 4 || nb > 96 || nb % 4 || ne < -1 || ne > 128 || (ne > 0 && ne % 4) || nc < 4 || nc > 48 || nc % 4 || nc > nb || !kkt_fits(nb, ne, nc))
    return fail(IEM_E_ARG, "chain KKT: block size must be a multiple of 4 in 4..96, border size a multiple of 4 in 0..128, coupling width a multiple of 4 in 4..48 (and <= the block size), tiles within the LDS of a CU");
  auto it = m->kkt_mods.find({nb, ne * 64 + nc});
  if (it == m->kkt_mods.end()) {
    iem_model::KktMod km;
    int rc = load_source(m, kkt_source(nb, ne, nc), &km.mod);
    if (rc) return rc;
    HIP_TRY(hipModuleGetFunction(&km.elim, km.mod, "kkt_eliminate"));
    int wmax, wpe_;
    kkt_shape(nb, ne, &wmax, &wpe_);
    km.elim_wg = 64u * (unsigned)std::min((nb + 15) / 16, wmax);   // KKT_T of csrc/iem_kkt_device.h
    if (kkt_rowwise(nb, ne)) { km.elim_wg = 64u * (unsigned)kkt_row_wpg(nb, ne); km.elim_bpw = (int)km.elim_wg / (nb / kkt_rowwise(nb, ne)); }
    HIP_TRY(hipModuleGetFunction(&km.upd, km.mod, "kkt_update"));
    HIP_TRY(hipModuleGetFunction(&km.fwd, km.mod, "kkt_forward"));
    HIP_TRY(hipModuleGetFunction(&km.bwd, km.mod, "kkt_backward"));
    HIP_TRY(hipModuleGetFunction(&km.gather, km.mod, "kkt_gather"));
    HIP_TRY(hipModuleGetFunction(&km.move, km.mod, "kkt_move"));
    HIP_TRY(hipModuleGetFunction(&km.colsum, km.mod, "kkt_colsum"));
    HIP_TRY(hipModuleGetFunction(&km.hub_z, km.mod, "kkt_hub_z"));
    HIP_TRY(hipModuleGetFunction(&km.hub_widen, km.mod, "kkt_hub_widen"));
    HIP_TRY(hipModuleGetFunction(&km.hub_mask, km.mod, "kkt_hub_mask"));
    HIP_TRY(hipModuleGetFunction(&km.hub_ety, km.mod, "kkt_hub_ety"));
    HIP_TRY(hipModuleGetFunction(&km.hub_ex, km.mod, "kkt_hub_ex"));
    HIP_TRY(hipModuleGetFunction(&km.hub_leaf, km.mod, "kkt_hub_leaf"));
    HIP_TRY(hipModuleGetFunction(&km.hub_diagmax, km.mod, "kkt_hub_diagmax"));
    if (kkt_rowwise(nb, ne) && nb <= 32) {      // the lane-per-row solves (csrc/iem_kkt_device.h: kkt_fz / kkt_fs / kkt_bw) for the shapes whose eliminate is lane-per-row too:
                                    // at 40 x 40 they lose 12 % to the 64-thread kernels (two launches per level, 40 of 64 lanes), at 20 x 20 they win 30 %
      HIP_TRY(hipModuleGetFunction(&km.fz, km.mod, "kkt_fz"));
      HIP_TRY(hipModuleGetFunction(&km.fs, km.mod, "kkt_fs"));
      HIP_TRY(hipModuleGetFunction(&km.bw, km.mod, "kkt_bw"));
      km.solve_bpw = 64 / nb;
    }
    it = m->kkt_mods.emplace(std::make_pair(nb, ne * 64 + nc), km).first;
  }
  *out = &it->second;
  return IEM_OK;
}
struct KktArgsH { double *D, *Bt, *BR, *E, *Z, *Gp; const int *rows, *cols; long long *info; long long S, s; int final_block; double tiny; long long T = 0; };
struct KktSolveArgsH { const double *D, *Bt, *BR, *Z; const int *rows, *cols; double *r, *z, *rBp; const double *xB; long long S, s; int final_block; long long T = 0; };
int kkt_launch_raw(iem_model *m, hipFunction_t fn, void *args, size_t sz, long long grid, unsigned block) {
  if (grid <= 0) return IEM_OK;
  void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  HIP_TRY(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, block, 1, 1, 0, m->stream, nullptr, cfg));
  return IEM_OK;
}
int kkt_launch(iem_model *m, hipFunction_t fn, KktArgsH a, long long grid, unsigned block) { return kkt_launch_raw(m, fn, &a, sizeof a, grid, block); }
int kkt_launch_solve(iem_model *m, hipFunction_t fn, KktSolveArgsH a, long long grid, unsigned block) { return kkt_launch_raw(m, fn, &a, sizeof a, grid, block); }
// kkt_eliminate over `blocks` blocks of the level: a workgroup takes elim_bpw of them
int kkt_launch_elim(iem_model *m, const iem_model::KktMod *km, KktArgsH a, long long blocks) {
  return kkt_launch_raw(m, km->elim, &a, sizeof a, (blocks + km->elim_bpw - 1) / km->elim_bpw, km->elim_wg);
}
}  // namespace

int iem_kkt_source(int nb, int ne, int nc, char **out_src, uint64_t *out_key) {
  const std::string s = kkt_source(nb, ne, nc);
  if (out_src) { *out_src = (char *)std::malloc(s.size() + 1); std::memcpy(*out_src, s.c_str(), s.size() + 1); }
  if (out_key) *out_key = iem::fnv1a64(s);
  return IEM_OK;
}

int iem_kkt_chain_factor(iem_model *m, int64_t S, int nb, int ne, int nc, double *d_D, double *d_Bt, double *d_BR, const int32_t *d_rows,
                         const int32_t *d_cols, double *d_E, double *d_Z, double *d_Gp, int64_t *d_info, double tiny) {
  const bool chained = d_Bt != nullptr;   // d_Bt == NULL: the blocks do not couple to each other (reach 0), only to the border
  if (!m || S < 1 || !d_D || (chained && (!d_BR || !d_rows || !d_cols)) || !d_info || (ne > 0 && (!d_E || !d_Z || !d_Gp))) return fail(IEM_E_ARG, "bad argument");
  DevGuard dg_(m->device);
  iem_model::KktMod *km = nullptr;
  int rc = kkt_module(m, nb, ne, nc, &km);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(d_info, 0, 24, m->stream));
  KktArgsH A{d_D, d_Bt, d_BR, d_E, d_Z, d_Gp, d_rows, d_cols, (long long *)d_info, (long long)S, 1, 0, tiny};
  if (!chained) {   // one launch: every block against the border
    A.final_block = 2;
    return kkt_launch_elim(m, km, A, S);
  }
  for (long long s = 1; s < S; s *= 2) {   // level: eliminate the odd multiples of s, fold them into the even ones
    A.s = s;
    const long long n_elim = (S - s + 2 * s - 1) / (2 * s), n_surv = (S + 2 * s - 1) / (2 * s);
    if ((rc = kkt_launch_elim(m, km, A, n_elim)) != IEM_OK) return rc;
    if ((rc = kkt_launch(m, km->upd, A, n_surv, nc <= 8 ? 64u : nc <= 16 ? 128u : 256u)) != IEM_OK) return rc;   // KKT_TU
  }
  A.final_block = 1;
  return kkt_launch_elim(m, km, A, 1);
}

/* ONE step of the same reduction, for a caller that interleaves work of its own between the levels (kkt_chain.HubChainKKT: the
 * span-sparse border of a laned 2-D grid).  what = 0: eliminate the blocks (2t+1)s (in-place inverses, BR kept), 1: fold them
 * into the survivors 2ts, 2: the last remaining block (index 0), 3: clear the pivot counters.  No border (ne = 0 module). */
int iem_kkt_chain_level(iem_model *m, int64_t S, int64_t lane_len, int nb, int nc, double *d_D, double *d_Bt, double *d_BR, const int32_t *d_rows,
                        const int32_t *d_cols, int64_t *d_info, double tiny, int64_t s, int what) {
  const long long T = lane_len > 0 ? lane_len : S;
  if (!m || S < 1 || S % T || !d_D || !d_Bt || !d_BR || !d_rows || !d_cols || !d_info || s < 1 || what < 0 || what > 3) return fail(IEM_E_ARG, "bad argument");
  DevGuard dg_(m->device);
  if (what == 3) { HIP_TRY(hipMemsetAsync(d_info, 0, 24, m->stream)); return IEM_OK; }
  iem_model::KktMod *km = nullptr;
  int rc = kkt_module(m, nb, 0, nc, &km);
  if (rc) return rc;
  const long long lanes = S / T;
  KktArgsH A{d_D, d_Bt, d_BR, nullptr, nullptr, nullptr, d_rows, d_cols, (long long *)d_info, (long long)S, (long long)s, 0, tiny, T};
  if (what == 2) { A.final_block = 1; A.s = 1; return kkt_launch_elim(m, km, A, lanes); }
  if (s >= T) return IEM_OK;
  const long long n_elim = lanes * ((T - s + 2 * s - 1) / (2 * s)), n_surv = lanes * ((T + 2 * s - 1) / (2 * s));
  if (what == 0) return kkt_launch_elim(m, km, A, n_elim);
  return kkt_launch(m, km->upd, A, n_surv, nc <= 8 ? 64u : nc <= 16 ? 128u : 256u);
}

/* The span-sparse border columns of HubChainKKT between two levels (csrc/iem_kkt_device.h: kkt_hub_z, kkt_hub_widen). */
int iem_kkt_hub_level(iem_model *m, int64_t S, int64_t lane_len, int nb, int nc, const double *d_Dinv, const double *d_Bt, const int32_t *d_q, int nq,
                      const int32_t *d_qr, int nr, const int32_t *d_qc, int ncq, int hw, int64_t s, const double *d_E, double *d_Z, double *d_En, int last) {
  const long long T = lane_len > 0 ? lane_len : S;
  if (!m || S < 1 || S % T || !d_Dinv || !d_Bt || !d_q || !d_qr || !d_qc || nq < 1 || nq > 24 || nr < 0 || ncq < 0 || nr > nc || ncq > nc || hw < 1 || s < 1 || !d_E || !d_Z ||
      (!last && (!d_En || s >= T)))
    return fail(IEM_E_ARG, "bad argument");
  DevGuard dg_(m->device);
  iem_model::KktMod *km = nullptr;
  int rc = kkt_module(m, nb, 0, nc, &km);
  if (rc) return rc;
  struct { const double *D, *Bt, *E, *Z; double *out; const int *q, *qr, *qc; long long T, lanes, s, n_e, n_s; int nb, nc, nq, nr, ncq, hw, last; } A{
      d_Dinv, d_Bt, d_E, nullptr, d_Z, d_q, d_qr, d_qc, T, S / T, (long long)s, 0, 0, nb, nc, nq, nr, ncq, hw, last ? 1 : 0};
  const long long alive = (T + s - 1) / s, W = (2 * s - 1) * hw, W2 = (4 * s - 1) * hw;
  A.n_e = alive / 2; A.n_s = alive - A.n_e;
  const long long nz = (last ? 1 : A.n_e) * A.lanes * W;
  if ((rc = kkt_launch_raw(m, km->hub_z, &A, sizeof A, (nz + 255) / 256, 256)) != IEM_OK || last) return rc;
  A.Z = d_Z; A.out = d_En;
  return kkt_launch_raw(m, km->hub_widen, &A, sizeof A, (A.n_s * A.lanes * W2 + 255) / 256, 256);
}

int iem_kkt_chain_solve(iem_model *m, int64_t S, int nb, int ne, int nc, const double *d_Dinv, const double *d_Bt, const double *d_BR,
                        const int32_t *d_rows, const int32_t *d_cols, const double *d_Z, double *d_r, double *d_z, double *d_rBp,
                        const double *d_xB, int phase) {
  return iem_kkt_chain_solve_lanes(m, S, S, nb, ne, nc, d_Dinv, d_Bt, d_BR, d_rows, d_cols, d_Z, d_r, d_z, d_rBp, d_xB, phase);
}

int iem_kkt_chain_solve_lanes(iem_model *m, int64_t S, int64_t lane_len, int nb, int ne, int nc, const double *d_Dinv, const double *d_Bt, const double *d_BR,
                              const int32_t *d_rows, const int32_t *d_cols, const double *d_Z, double *d_r, double *d_z, double *d_rBp,
                              const double *d_xB, int phase) {
  const bool chained = d_Bt != nullptr;   // as in iem_kkt_chain_factor
  const long long T = lane_len > 0 ? lane_len : S;
  if (!m || S < 1 || S % T || (T != S && ne > 0) || !d_Dinv || (chained && (!d_BR || !d_rows || !d_cols || !d_z)) || !d_r || (ne > 0 && (!d_Z || (phase == 0 && !d_rBp) || (phase == 1 && !d_xB)))) return fail(IEM_E_ARG, "bad argument");
  DevGuard dg_(m->device);
  iem_model::KktMod *km = nullptr;
  int rc = kkt_module(m, nb, ne, nc, &km);
  if (rc) return rc;
  const long long lanes = S / T;
  KktSolveArgsH A{d_Dinv, d_Bt, d_BR, d_Z, d_rows, d_cols, d_r, d_z, d_rBp, d_xB, (long long)S, 1, 0, T};
  static const bool old_solves = [] { const char *e = getenv("IEM_KKT_EXPERIMENTS"), *o = getenv("IEM_KKT_OLD_SOLVES"); return e && !std::strcmp(e, "1") && o && !std::strcmp(o, "1"); }();
  const bool rowwise = km->fz && !old_solves;        // no border, blocks that fit a wave: a lane per row
  const long long bpw = rowwise ? km->solve_bpw : 1;
  if (!chained) {            // independent blocks: the border terms of all blocks (forward), every block's own solve (backward)
    if (phase != 0 && phase != 1) return fail(IEM_E_ARG, "phase must be 0 (forward) or 1 (backward)");
    A.final_block = 2;
    if (phase == 0) return ne > 0 ? kkt_launch_solve(m, km->fwd, A, S, 64) : IEM_OK;
    return rowwise ? kkt_launch_solve(m, km->fz, A, (S + bpw - 1) / bpw, 64) : kkt_launch_solve(m, km->bwd, A, S, 64);
  }
  if (phase == 0 && rowwise) {
    for (long long s = 1; s < T; s *= 2) {
      A.s = s;
      const long long n_elim = lanes * ((T - s + 2 * s - 1) / (2 * s)), n_surv = lanes * ((T + 2 * s - 1) / (2 * s));
      if ((rc = kkt_launch_solve(m, km->fz, A, (n_elim + bpw - 1) / bpw, 64)) != IEM_OK) return rc;      // z of the level's eliminated blocks ...
      if ((rc = kkt_launch_solve(m, km->fs, A, (n_surv + 63) / 64, 64)) != IEM_OK) return rc;            // ... folded into the survivors' right-hand sides
    }
    return IEM_OK;
  }
  if (phase == 0) {          // forward: levels up, then the last block's border contribution
    for (long long s = 1; s < T; s *= 2) {
      A.s = s;
      const long long n_elim = lanes * ((T - s + 2 * s - 1) / (2 * s)), n_surv = lanes * ((T + 2 * s - 1) / (2 * s));
      if ((rc = kkt_launch_solve(m, km->fwd, A, n_surv + n_elim, 64)) != IEM_OK) return rc;
    }
    A.final_block = 1;
    return ne > 0 ? kkt_launch_solve(m, km->fwd, A, lanes, 64) : IEM_OK;
  }
  if (phase != 1) return fail(IEM_E_ARG, "phase must be 0 (forward) or 1 (backward)");
  A.final_block = 1;
  if ((rc = rowwise ? kkt_launch_solve(m, km->fz, A, (lanes + bpw - 1) / bpw, 64) : kkt_launch_solve(m, km->bwd, A, lanes, 64)) != IEM_OK) return rc;
  A.final_block = 0;
  long long top = 1;
  while (top * 2 < T) top *= 2;
  for (long long s = top; s >= 1; s /= 2) {
    if (s >= T) continue;
    A.s = s;
    const long long n_elim = lanes * ((T - s + 2 * s - 1) / (2 * s));
    if ((rc = rowwise ? kkt_launch_solve(m, km->bw, A, (n_elim + bpw - 1) / bpw, 64) : kkt_launch_solve(m, km->bwd, A, n_elim, 64)) != IEM_OK) return rc;
  }
  return IEM_OK;
}

/* ---- the chain KKT solver as ONE object behind the C-ABI (what a host without the Python layer binds) ---------------------- */
struct iem_kkt {
  iem_model *m = nullptr;
  iem::KktLayout L;
  iem_model::KktMod *km = nullptr;
  double *d_flat = nullptr, *d_BR = nullptr, *d_Z = nullptr, *d_Gp = nullptr, *d_r = nullptr, *d_z = nullptr, *d_rBp = nullptr, *d_xB = nullptr, *d_part = nullptr;
  int32_t *d_rows = nullptr, *d_cols = nullptr;
  long long *d_dest = nullptr, *d_on = nullptr, *d_pos = nullptr, *d_border = nullptr, *d_bloc = nullptr, *d_info = nullptr;
  unsigned *d_seg = nullptr, *d_perm = nullptr;
  int64_t n_dest = 0, n_on = 0, n_h = 0, n_j = 0;
  std::vector<double> Gs;        // the border's Schur complement (host), set by iem_kkt_factor
  bool factored = false;
  struct KktHub *hub = nullptr;  // hub mode (L.hubs): the span-sparse border's buffers and the library handle of its GEMMs
};

// rocBLAS for the plain library GEMMs / GEMVs of the hubs' Schur complement — loaded on first use (dlopen), so that the library
// does not depend on it for anything else
struct RocBlas {
  void *lib = nullptr, *handle = nullptr;
  int (*create)(void **) = nullptr;
  int (*destroy)(void *) = nullptr;
  int (*set_stream)(void *, hipStream_t) = nullptr;
  int (*dgemm)(void *, int, int, int, int, int, const double *, const double *, int, const double *, int, const double *, double *, int) = nullptr;
  int (*dgemm_sb)(void *, int, int, int, int, int, const double *, const double *, int, long long, const double *, int, long long, const double *, double *, int, long long, int) = nullptr;
  int (*dgemv)(void *, int, int, int, const double *, const double *, int, const double *, int, const double *, double *, int) = nullptr;
  int (*dgemv_sb)(void *, int, int, int, const double *, const double *, int, long long, const double *, int, long long, const double *, double *, int, long long, int) = nullptr;
  int (*dgeam)(void *, int, int, int, int, const double *, const double *, int, const double *, const double *, int, double *, int) = nullptr;
};
struct KktHub {
  RocBlas bl;
  int32_t *d_q = nullptr, *d_qr = nullptr, *d_qc = nullptr;
  double *E[2] = {nullptr, nullptr}, *Z = nullptr, *Sd = nullptr, *Lmat = nullptr, *LT = nullptr, *Dinv = nullptr, *Linv = nullptr, *Nmat = nullptr, *X[2] = {nullptr, nullptr}, *eye96 = nullptr,
         *eyeP = nullptr, *x = nullptr, *w = nullptr, *tmp = nullptr, *r2 = nullptr;
  long long *d_dinfo = nullptr;
  int64_t e_cap = 0, steps = 0, npanels = 0;
};

namespace {
int kkt_upload_bytes(void **d, const void *h, size_t bytes) {
  HIP_TRY(hipMalloc(d, std::max<size_t>(bytes, 8)));
  if (bytes) HIP_TRY(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
  return IEM_OK;
}
#define kkt_upload(dptr, vec) kkt_upload_bytes((void **)(dptr), (vec).data(), (vec).size() * sizeof((vec)[0]))
// column sums of a device S x w matrix, to the host
int kkt_colsum_host(iem_kkt *k, const double *d_in, int64_t rows, int64_t w, std::vector<double> &out) {
  // two launches: up to 512 row chunks x column chunks of 256 (every CU busy: the border terms of 1e4 OPF scenarios are 216 MB),
  // then the partials into one row; w doubles come home
  iem_model *m = k->m;
  const int64_t ncc = (w + 255) / 256, per = (rows + 511) / 512, nrc = (rows + per - 1) / per;
  struct Sum { const double *in; double *out; long long rows, w, rows_per_wg; };
  Sum A{d_in, k->d_part, (long long)rows, (long long)w, (long long)per};
  int rc = kkt_launch_raw(m, k->km->colsum, &A, sizeof A, nrc * ncc, 256);
  if (rc) return rc;
  double *d_row = k->d_part + 512 * w;
  Sum B{k->d_part, d_row, (long long)nrc, (long long)w, (long long)nrc};
  if ((rc = kkt_launch_raw(m, k->km->colsum, &B, sizeof B, ncc, 256))) return rc;
  out.assign((size_t)w, 0.0);
  HIP_TRY(hipStreamSynchronize(m->stream));
  HIP_TRY(hipMemcpy(out.data(), d_row, (size_t)w * 8, hipMemcpyDeviceToHost));
  return IEM_OK;
}
}  // namespace


/* ---- hub mode of the KKT object: a laned 2-D grid whose border is kept as span-sparse hub columns (kkt_chain.HubChainKKT is the
 * Python-held form of the same pipeline; tests compare the two) ---------------------------------------------------------------- */
namespace {
constexpr int HUB_LEAF = 96, HUB_PW = 5 * HUB_LEAF, HUB_CH = 10 * HUB_LEAF, HUB_CLIP = 16;
constexpr int RB_N = 111, RB_T = 112;      // rocblas_operation_none / _transpose
int hub_blas_load(iem_kkt *k) {
  RocBlas &b = k->hub->bl;
  if (b.handle) return b.set_stream(b.handle, k->m->stream) == 0 ? IEM_OK : fail(IEM_E_HIP, "rocblas_set_stream");      // (the handle's stream may have been changed since)
  b.lib = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
  if (!b.lib) b.lib = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_GLOBAL);
  if (!b.lib) return fail(IEM_E_HIP, std::string("chain KKT (hub border): cannot load librocblas.so (") + dlerror() + ")");
  auto sym = [&](const char *n) { return dlsym(b.lib, n); };
  b.create = (decltype(b.create))sym("rocblas_create_handle");
  b.destroy = (decltype(b.destroy))sym("rocblas_destroy_handle");
  b.set_stream = (decltype(b.set_stream))sym("rocblas_set_stream");
  b.dgemm = (decltype(b.dgemm))sym("rocblas_dgemm");
  b.dgemm_sb = (decltype(b.dgemm_sb))sym("rocblas_dgemm_strided_batched");
  b.dgemv = (decltype(b.dgemv))sym("rocblas_dgemv");
  b.dgemv_sb = (decltype(b.dgemv_sb))sym("rocblas_dgemv_strided_batched");
  b.dgeam = (decltype(b.dgeam))sym("rocblas_dgeam");
  if (!b.create || !b.destroy || !b.set_stream || !b.dgemm || !b.dgemm_sb || !b.dgemv || !b.dgemv_sb || !b.dgeam) return fail(IEM_E_HIP, "chain KKT (hub border): librocblas.so lacks an entry point");
  if (b.create(&b.handle) != 0) return fail(IEM_E_HIP, "rocblas_create_handle");
  if (b.set_stream(b.handle, k->m->stream) != 0) return fail(IEM_E_HIP, "rocblas_set_stream");
  return IEM_OK;
}
#define RB_TRY(call) do { if ((call) != 0) return fail(IEM_E_HIP, "rocBLAS call failed: " #call); } while (0)
// ROW-MAJOR  C[m x n] = alpha op(A) op(B) + beta C   through the column-major library (C' = op(B)' op(A)')
int gemm_rm(iem_kkt *k, bool ta, bool tb, int64_t m, int64_t n, int64_t kk, double alpha, const double *A, int64_t lda, const double *B, int64_t ldb, double beta, double *C, int64_t ldc) {
  if (m <= 0 || n <= 0) return IEM_OK;
  RB_TRY(k->hub->bl.dgemm(k->hub->bl.handle, tb ? RB_T : RB_N, ta ? RB_T : RB_N, (int)n, (int)m, (int)kk, &alpha, B, (int)ldb, A, (int)lda, &beta, C, (int)ldc));
  return IEM_OK;
}
// ROW-MAJOR  y = alpha op(A) x + beta y,  A [m x n]
int gemv_rm(iem_kkt *k, bool ta, int64_t m, int64_t n, double alpha, const double *A, int64_t lda, const double *x, double beta, double *y) {
  if (m <= 0 || n <= 0) return IEM_OK;
  RB_TRY(k->hub->bl.dgemv(k->hub->bl.handle, ta ? RB_N : RB_T, (int)n, (int)m, &alpha, A, (int)lda, x, 1, &beta, y, 1));
  return IEM_OK;
}
int copy2d(iem_kkt *k, double *dst, int64_t ldd, const double *src, int64_t lds, int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return IEM_OK;
  HIP_TRY(hipMemcpy2DAsync(dst, (size_t)ldd * 8, src, (size_t)lds * 8, (size_t)cols * 8, (size_t)rows, hipMemcpyDeviceToDevice, k->m->stream));
  return IEM_OK;
}
// Sv -= A' Z  (symmetric; Sv: w x w at hub h0 of the Schur complement, row length ld; A, Z: K x w, row length ldw): wide products only
// into the block lower triangle, column chunks cut at the multiples of HUB_CH hubs (the pivot blocks of the LDL' never straddle them)
int hub_sub_lower(iem_kkt *k, double *Sv, int64_t ld, const double *A, const double *Z, int64_t ldw, int64_t K, int64_t w, int64_t h0) {
  if (w <= 2 * HUB_CH) return gemm_rm(k, true, false, w, w, K, -1.0, A, ldw, Z, ldw, 1.0, Sv, ld);
  int64_t c0 = 0;
  while (c0 < w) {
    const int64_t c1 = std::min<int64_t>(w, ((h0 + c0) / HUB_CH + 1) * HUB_CH - h0);
    int rc = gemm_rm(k, true, false, w - c0, c1 - c0, K, -1.0, A + c0, ldw, Z + c0, ldw, 1.0, Sv + c0 * ld + c0, ld);
    if (rc) return rc;
    c0 = c1;
  }
  return IEM_OK;
}
int hub_level_launch(iem_kkt *k, int64_t s, const double *E, double *Z, double *En, int last) {
  const iem::KktLayout &L = k->L;
  return iem_kkt_hub_level(k->m, L.S, L.Tp, L.nb, L.nc, k->d_flat + L.oD(), k->d_flat + L.oB(), k->hub->d_q, L.nq, k->hub->d_qr, L.nr, k->hub->d_qc, L.ncq, (int)L.hw, s, E, Z, En, last);
}
int hub_alloc(iem_kkt *k) {
  const iem::KktLayout &L = k->L;
  KktHub *h = k->hub = new KktHub;
  int rc;
  if ((rc = kkt_upload(&h->d_q, L.Q)) || (rc = kkt_upload(&h->d_qr, L.qR)) || (rc = kkt_upload(&h->d_qc, L.qC))) return rc;
  const int64_t per = L.lanes * L.nq;
  int64_t cap = L.Tp * L.hw, P = 1;
  for (int64_t s = 1; s < L.Tp; s *= 2) {
    const int64_t alive = (L.Tp + s - 1) / s, ns = alive - alive / 2;
    cap = std::max(cap, std::max(alive * (2 * s - 1), ns * (4 * s - 1)) * L.hw);
    P = 2 * s;
  }
  cap = std::max(cap, (2 * P - 1) * L.hw);
  h->e_cap = cap * per;
  for (int i = 0; i < 2; ++i) HIP_TRY(hipMalloc((void **)&h->E[i], (size_t)h->e_cap * 8));
  HIP_TRY(hipMalloc((void **)&h->Z, (size_t)h->e_cap * 8));
  const int64_t H = L.H;
  h->steps = (H + HUB_LEAF - 1) / HUB_LEAF; h->npanels = (H + HUB_PW - 1) / HUB_PW;
  HIP_TRY(hipMalloc((void **)&h->Sd, (size_t)(H * H) * 8));
  HIP_TRY(hipMalloc((void **)&h->Lmat, (size_t)(H * H) * 8));      // L (the scaled columns); Sd keeps L D, the columns before scaling
  HIP_TRY(hipMalloc((void **)&h->LT, (size_t)(H * H) * 8));        // L' panel by panel: the backward substitution reads rows, like the forward one
  HIP_TRY(hipMalloc((void **)&h->Dinv, (size_t)(h->steps * HUB_LEAF * HUB_LEAF) * 8));
  HIP_TRY(hipMalloc((void **)&h->Linv, (size_t)(h->npanels * HUB_PW * HUB_PW) * 8));
  HIP_TRY(hipMalloc((void **)&h->Nmat, (size_t)(HUB_PW * HUB_PW) * 8));
  for (int i = 0; i < 2; ++i) HIP_TRY(hipMalloc((void **)&h->X[i], (size_t)(HUB_PW * HUB_PW) * 8));
  std::vector<double> eye((size_t)(HUB_PW * HUB_PW), 0.0), eye96((size_t)(HUB_LEAF * HUB_LEAF), 0.0);
  for (int i = 0; i < HUB_PW; ++i) eye[(size_t)(i * HUB_PW + i)] = 1.0;
  for (int i = 0; i < HUB_LEAF; ++i) eye96[(size_t)(i * HUB_LEAF + i)] = 1.0;
  if ((rc = kkt_upload(&h->eyeP, eye)) || (rc = kkt_upload(&h->eye96, eye96))) return rc;
  const int64_t hv = std::max<int64_t>(h->steps * HUB_LEAF, L.Tp * L.hw);
  HIP_TRY(hipMalloc((void **)&h->x, (size_t)hv * 8));
  HIP_TRY(hipMalloc((void **)&h->w, (size_t)hv * 8));
  HIP_TRY(hipMalloc((void **)&h->tmp, (size_t)HUB_PW * 8));
  HIP_TRY(hipMalloc((void **)&h->r2, (size_t)(L.S * L.nb) * 8));
  HIP_TRY(hipMalloc((void **)&h->d_dinfo, (size_t)(h->steps * 3) * 8));
  return IEM_OK;
}
void hub_free(iem_kkt *k) {
  KktHub *h = k->hub;
  if (!h) return;
  for (void *p : {(void *)h->d_q, (void *)h->d_qr, (void *)h->d_qc, (void *)h->E[0], (void *)h->E[1], (void *)h->Z, (void *)h->Sd, (void *)h->Lmat, (void *)h->LT, (void *)h->Dinv, (void *)h->Linv,
                  (void *)h->Nmat, (void *)h->X[0], (void *)h->X[1], (void *)h->eye96, (void *)h->eyeP, (void *)h->x, (void *)h->w, (void *)h->tmp, (void *)h->r2, (void *)h->d_dinfo})
    if (p) hipFree(p);
  if (h->bl.handle && h->bl.destroy) h->bl.destroy(h->bl.handle);
  delete h;
  k->hub = nullptr;
}
// the hubs' Schur complement Sd (H x H, row-major, lower triangle + the pivot blocks' squares valid): block LDL' in panels of HUB_PW,
// pivot blocks of HUB_LEAF inverted by kkt_eliminate (which counts their pivot signs).  Sd keeps the columns BEFORE scaling (L D),
// Lmat receives L = (L D) D^-1 — the updates are L (L D)': no copy of a panel is ever needed — and LT its transpose, panel by panel.
int hub_dense_factor(iem_kkt *k) {
  const iem::KktLayout &L = k->L;
  KktHub *h = k->hub;
  iem_model *m = k->m;
  const int64_t n = L.H;
  double *Sd = h->Sd, *Lm = h->Lmat;
  int rc;
  iem_model::KktMod *leaf = nullptr;
  if ((rc = kkt_module(m, HUB_LEAF, -1, 4, &leaf))) return rc;
  HIP_TRY(hipMemsetAsync(h->d_dinfo, 0, (size_t)(h->steps * 3) * 8, m->stream));
  // pivot threshold of the hubs' blocks: RELATIVE to the largest diagonal entry of their Schur complement — a pivot at rounding level of
  // that scale is neither sign and is reported as doubtful (as the dense border's near-zero eigenvalues are: ADVICE r03), not as positive
  double tiny = 1e-30;
  {
    struct { const double *S; double *out; long long n, ld; } A{Sd, h->tmp, (long long)n, (long long)n};
    if ((rc = kkt_launch_raw(m, k->km->hub_diagmax, &A, sizeof A, 1, 256))) return rc;
    double dmax = 0.0;
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemcpy(&dmax, h->tmp, 8, hipMemcpyDeviceToHost));
    if (dmax > 0.0 && std::isfinite(dmax)) tiny = std::max(tiny, 1e-14 * dmax);
  }
  for (int64_t p0 = 0, pi = 0; p0 < n; p0 += HUB_PW, ++pi) {
    const int64_t p1 = std::min<int64_t>(p0 + HUB_PW, n), pw = p1 - p0;
    for (int64_t kk = p0; kk < p1; kk += HUB_LEAF) {
      const int64_t e = std::min<int64_t>(kk + HUB_LEAF, p1), w = e - kk, ki = kk / HUB_LEAF;
      double *blk = h->Dinv + ki * HUB_LEAF * HUB_LEAF;
      { struct { const double *src; double *dst; long long ld; int w, nb; } A{Sd + kk * n + kk, blk, (long long)n, (int)w, HUB_LEAF};
        if ((rc = kkt_launch_raw(m, k->km->hub_leaf, &A, sizeof A, (HUB_LEAF * HUB_LEAF + 255) / 256, 256))) return rc; }
      {   // in-place inverse + pivot signs: one block, no chain, no border
        KktArgsH A{blk, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, h->d_dinfo + 3 * ki, 1, 1, 2, tiny};
        if ((rc = kkt_launch_elim(m, leaf, A, 1))) return rc;
      }
      if (e < n) {
        if ((rc = gemm_rm(k, false, false, n - e, w, w, 1.0, Sd + e * n + kk, n, blk, HUB_LEAF, 0.0, Lm + e * n + kk, n))) return rc;                       // L = (L D) D^-1
        if (e < p1 && (rc = gemm_rm(k, false, true, n - e, p1 - e, w, -1.0, Lm + e * n + kk, n, Sd + e * n + kk, n, 1.0, Sd + e * n + e, n))) return rc;    // the panel's own columns
      }
    }
    {   // the panel's unit lower triangle L_pp = I + N (N strictly block lower: N^5 = 0), inverted once: (I + N)^-1 = I - N (I - N (I - ...))
      struct { const double *src; double *N, *X; long long ld; int pw, ldp, leaf; } A{Lm + p0 * n + p0, h->Nmat, h->X[0], (long long)n, (int)pw, HUB_PW, HUB_LEAF};
      if ((rc = kkt_launch_raw(m, k->km->hub_mask, &A, sizeof A, (pw * pw + 255) / 256, 256))) return rc;
      int cur = 0;
      for (int64_t it = 0; it < (pw - 1) / HUB_LEAF - 1; ++it) {
        HIP_TRY(hipMemcpyAsync(h->X[1 - cur], h->eyeP, (size_t)(HUB_PW * HUB_PW) * 8, hipMemcpyDeviceToDevice, m->stream));
        if ((rc = gemm_rm(k, false, false, pw, pw, pw, -1.0, h->Nmat, HUB_PW, h->X[cur], HUB_PW, 1.0, h->X[1 - cur], HUB_PW))) return rc;
        cur = 1 - cur;
      }
      HIP_TRY(hipMemcpyAsync(h->Linv + pi * HUB_PW * HUB_PW, h->X[cur], (size_t)(HUB_PW * HUB_PW) * 8, hipMemcpyDeviceToDevice, m->stream));
    }
    if (p1 < n) {   // LT[p0:p1, p1:] = L[p1:, p0:p1]'  (column-major view: C ((n - p1) x pw) = A' with A = the pw x (n - p1) view of the panel)
      const double one = 1.0, zero = 0.0;
      RB_TRY(h->bl.dgeam(h->bl.handle, RB_T, RB_N, (int)(n - p1), (int)pw, &one, Lm + p1 * n + p0, (int)n, &zero, h->LT + p0 * n + p1, (int)n, h->LT + p0 * n + p1, (int)n));
    }
    for (int64_t c0 = p1; c0 < n; c0 += HUB_CH) {   // the rest of the matrix, lower triangle only, once per panel (inner dimension = the panel's width)
      const int64_t c1 = std::min<int64_t>(c0 + HUB_CH, n);
      if ((rc = gemm_rm(k, false, true, n - c0, c1 - c0, pw, -1.0, Lm + c0 * n + p0, n, Sd + c0 * n + p0, n, 1.0, Sd + c0 * n + c0, n))) return rc;
    }
  }
  return IEM_OK;
}
// x (H) in place: S x = b with the factors above
int hub_dense_solve(iem_kkt *k, double *x) {
  const iem::KktLayout &L = k->L;
  KktHub *h = k->hub;
  iem_model *m = k->m;
  const int64_t n = L.H;
  int rc;
  for (int64_t p0 = 0, pi = 0; p0 < n; p0 += HUB_PW, ++pi) {       // L z = b
    const int64_t p1 = std::min<int64_t>(p0 + HUB_PW, n), pw = p1 - p0;
    if ((rc = gemv_rm(k, false, pw, pw, 1.0, h->Linv + pi * HUB_PW * HUB_PW, HUB_PW, x + p0, 0.0, h->tmp))) return rc;
    HIP_TRY(hipMemcpyAsync(x + p0, h->tmp, (size_t)pw * 8, hipMemcpyDeviceToDevice, m->stream));
    if (p1 < n && (rc = gemv_rm(k, false, n - p1, pw, -1.0, h->Lmat + p1 * n + p0, n, x + p0, 1.0, x + p1))) return rc;
  }
  {   // D^-1: one 96 x 96 product per pivot block (x is padded up to the last whole block with zeros)
    const double one = 1.0, zero = 0.0;
    if (h->steps * HUB_LEAF > n) HIP_TRY(hipMemsetAsync(x + n, 0, (size_t)(h->steps * HUB_LEAF - n) * 8, m->stream));
    RB_TRY(h->bl.dgemv_sb(h->bl.handle, RB_T, HUB_LEAF, HUB_LEAF, &one, h->Dinv, HUB_LEAF, (long long)HUB_LEAF * HUB_LEAF, x, 1, HUB_LEAF, &zero, h->w, 1, HUB_LEAF, (int)h->steps));
  }
  double *w = h->w;
  for (int64_t pi = h->npanels - 1; pi >= 0; --pi) {               // L' x = w  (rows of LT: the same fast matrix-vector kernel as on the way down)
    const int64_t p0 = pi * HUB_PW, p1 = std::min<int64_t>(p0 + HUB_PW, n), pw = p1 - p0;
    if (p1 < n && (rc = gemv_rm(k, false, pw, n - p1, -1.0, h->LT + p0 * n + p1, n, w + p1, 1.0, w + p0))) return rc;
    if ((rc = gemv_rm(k, true, pw, pw, 1.0, h->Linv + pi * HUB_PW * HUB_PW, HUB_PW, w + p0, 0.0, h->tmp))) return rc;
    HIP_TRY(hipMemcpyAsync(w + p0, h->tmp, (size_t)pw * 8, hipMemcpyDeviceToDevice, m->stream));
  }
  HIP_TRY(hipMemcpyAsync(x, w, (size_t)n * 8, hipMemcpyDeviceToDevice, m->stream));
  return IEM_OK;
}
int hub_factor(iem_kkt *k, int64_t *out_inertia) {
  const iem::KktLayout &L = k->L;
  KktHub *h = k->hub;
  iem_model *m = k->m;
  int rc;
  if ((rc = hub_blas_load(k))) return rc;
  double *D = k->d_flat + L.oD(), *Bt = k->d_flat + L.oB(), *E0 = k->d_flat + L.oE(), *Sbig = k->d_flat + L.oG();
  const int64_t Tp = L.Tp, hw = L.hw, Hp = L.Hp, H = L.H, K = L.lanes * L.nq;
  auto level = [&](int64_t s, int what) { return iem_kkt_chain_level(m, L.S, Tp, L.nb, L.nc, D, Bt, k->d_BR, k->d_rows, k->d_cols, (int64_t *)k->d_info, 1e-30, s, what); };
  if ((rc = level(1, 3))) return rc;
  HIP_TRY(hipMemcpyAsync(h->E[0], E0, (size_t)(Tp * K * hw) * 8, hipMemcpyDeviceToDevice, m->stream));
  int cur = 0;
  int64_t s = 1;
  for (; s < Tp; s *= 2) {
    if ((rc = level(s, 0))) return rc;                                      // D of the blocks t = (2a+1)s now holds their inverses
    const int64_t alive = (Tp + s - 1) / s, n_e = alive / 2, W = (2 * s - 1) * hw;
    const double *E = h->E[cur];
    if ((rc = hub_level_launch(k, s, E, h->Z, h->E[1 - cur], 0))) return rc;   // Z = D^-1[Q, Q] E of the eliminated, the survivors' widened columns
    if (n_e > HUB_CLIP) {                                                     // many small intervals: one batched product onto disjoint diagonal blocks of S
      const double alpha = -1.0, beta = 1.0;
      // row-major C_i -= A_i' Z_i  ==  column-major  C_i' -= Z_i(c) A_i(c)'   with X(c) = X' (W x K, leading dimension W)
      RB_TRY(h->bl.dgemm_sb(h->bl.handle, RB_N, RB_T, (int)W, (int)W, (int)K, &alpha, h->Z, (int)W, (long long)(K * W), E + K * W, (int)W, (long long)(2 * K * W), &beta,
                            Sbig + hw * (Hp + 1), (int)Hp, (long long)(2 * s * hw * (Hp + 1)), (int)n_e));
    } else {                                                                  // few wide ones: clipped to the hubs that exist
      for (int64_t mi = 0; mi < n_e; ++mi) {
        const int64_t te = (2 * mi + 1) * s, h0 = (te - s + 1) * hw, w = std::min<int64_t>(W, H - h0);
        if (w <= 0) continue;
        if ((rc = hub_sub_lower(k, Sbig + h0 * (Hp + 1), Hp, E + (2 * mi + 1) * K * W, h->Z + mi * K * W, W, K, w, h0))) return rc;
      }
    }
    cur = 1 - cur;
    if ((rc = level(s, 1))) return rc;                                      // fold the inverses into the survivors' blocks and couplings
  }
  // one block per lane is left (t = 0), coupled to every hub (s is the power of two the levels stopped at)
  if ((rc = level(1, 2))) return rc;
  {
    const int64_t W = (2 * s - 1) * hw;
    if ((rc = hub_level_launch(k, s, h->E[cur], h->Z, nullptr, 1))) return rc;
    if ((rc = hub_sub_lower(k, Sbig, Hp, h->E[cur] + (s - 1) * hw, h->Z + (s - 1) * hw, W, K, H, 0))) return rc;
  }
  if ((rc = copy2d(k, h->Sd, H, Sbig, Hp, H, H))) return rc;
  if ((rc = hub_dense_factor(k))) return rc;
  long long info[3] = {0, 0, 0};
  std::vector<long long> dinfo((size_t)(h->steps * 3));
  HIP_TRY(hipStreamSynchronize(m->stream));
  HIP_TRY(hipMemcpy(info, k->d_info, 24, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(dinfo.data(), h->d_dinfo, dinfo.size() * 8, hipMemcpyDeviceToHost));
  int64_t neg = info[0], doubtful = info[1];
  for (int64_t i = 0; i < h->steps; ++i) { neg += dinfo[(size_t)(3 * i)]; doubtful += dinfo[(size_t)(3 * i + 1)]; }
  k->factored = true;
  if (out_inertia) { out_inertia[0] = L.nvar + L.ncon - neg; out_inertia[1] = neg; out_inertia[2] = doubtful; }
  return IEM_OK;
}
int hub_solve(iem_kkt *k, const double *d_rhs, double *d_sol) {
  const iem::KktLayout &L = k->L;
  KktHub *h = k->hub;
  iem_model *m = k->m;
  int rc;
  if ((rc = hub_blas_load(k))) return rc;
  const double *Dinv = k->d_flat + L.oD(), *Bt = k->d_flat + L.oB(), *E0 = k->d_flat + L.oE();
  struct Mv { double *dst; const double *src; const long long *di, *si; long long n; };
  struct Vec { const double *E0, *v; double *out; const int *q; long long T, lanes; int nb, nq, hw; };
  const size_t rbytes = (size_t)(L.S * L.nb) * 8;
  auto chain = [&](double *r) {
    int e = iem_kkt_chain_solve_lanes(m, L.S, L.Tp, L.nb, 0, L.nc, Dinv, Bt, k->d_BR, k->d_rows, k->d_cols, nullptr, r, k->d_z, nullptr, nullptr, 0);
    return e ? e : iem_kkt_chain_solve_lanes(m, L.S, L.Tp, L.nb, 0, L.nc, Dinv, Bt, k->d_BR, k->d_rows, k->d_cols, nullptr, r, k->d_z, nullptr, nullptr, 1);
  };
  HIP_TRY(hipMemsetAsync(h->r2, 0, rbytes, m->stream));
  { Mv A{h->r2, d_rhs, k->d_pos, k->d_on, (long long)k->n_on}; if ((rc = kkt_launch_raw(m, k->km->move, &A, sizeof A, (k->n_on + 255) / 256, 256))) return rc; }
  HIP_TRY(hipMemcpyAsync(k->d_r, h->r2, rbytes, hipMemcpyDeviceToDevice, m->stream));
  if ((rc = chain(k->d_r))) return rc;                                       // y = K_c^-1 r_c
  // r_B - E' y on the hubs, the dense solve, then x_c = K_c^-1 (r_c - E x_B)
  HIP_TRY(hipMemsetAsync(h->x, 0, (size_t)std::max<int64_t>(h->steps * HUB_LEAF, L.Tp * L.hw) * 8, m->stream));
  { Mv A{h->x, d_rhs, k->d_bloc, k->d_border, (long long)L.n_border}; if ((rc = kkt_launch_raw(m, k->km->move, &A, sizeof A, (L.n_border + 255) / 256, 256))) return rc; }
  { Vec A{E0, k->d_r, h->x, h->d_q, (long long)L.Tp, (long long)L.lanes, L.nb, L.nq, (int)L.hw}; if ((rc = kkt_launch_raw(m, k->km->hub_ety, &A, sizeof A, L.Tp, 64))) return rc; }
  if ((rc = hub_dense_solve(k, h->x))) return rc;
  if (L.Tp * L.hw > L.H) HIP_TRY(hipMemsetAsync(h->x + L.H, 0, (size_t)(L.Tp * L.hw - L.H) * 8, m->stream));
  { Vec A{E0, h->x, h->r2, h->d_q, (long long)L.Tp, (long long)L.lanes, L.nb, L.nq, (int)L.hw};
    if ((rc = kkt_launch_raw(m, k->km->hub_ex, &A, sizeof A, (L.Tp * L.lanes * L.nq + 255) / 256, 256))) return rc; }
  if ((rc = chain(h->r2))) return rc;
  { Mv A{d_sol, h->r2, k->d_on, k->d_pos, (long long)k->n_on}; if ((rc = kkt_launch_raw(m, k->km->move, &A, sizeof A, (k->n_on + 255) / 256, 256))) return rc; }
  { Mv A{d_sol, h->x, k->d_border, k->d_bloc, (long long)L.n_border}; if ((rc = kkt_launch_raw(m, k->km->move, &A, sizeof A, (L.n_border + 255) / 256, 256))) return rc; }
  return IEM_OK;
}
}  // namespace

static void kkt_fill_info(const iem::KktLayout &L, iem_kkt_info_t *out) {
  out->S = L.S; out->n = L.nvar + L.ncon; out->n_border = L.n_border; out->nb = L.nb; out->ne = L.ne; out->nc = L.nc;
  out->reach = L.reach; out->group = L.group; out->phase = L.phase; out->block_doubles = L.total();
  out->lanes = L.lanes; out->hub_ld = L.Hp; out->hubs = L.hubs ? 1 : 0; out->hub_rows = L.nq; out->hubs_per_block = (int32_t)L.hw; out->reserved_ = 0;
}

int iem_kkt_analyse_blob(const void *blob, size_t nbytes, int group, iem_kkt_info_t *info, int64_t **out_blk, int64_t **out_loc, int32_t **out_rows,
                         int32_t **out_cols, int64_t **out_dest, uint32_t **out_seg, uint32_t **out_perm, int64_t *out_n_dest, int64_t *out_n_perm) {
  if (!blob || !info) return fail(IEM_E_ARG, "null argument");
  try {
    iem::Model M;
    iem::parse_blob(blob, nbytes, M);
    std::vector<int64_t> jr((size_t)M.nnzj), jc((size_t)M.nnzj), hr((size_t)M.nnzh), hc((size_t)M.nnzh);
    jac_structure_host(M, jr.data(), jc.data(), 0);
    hess_structure_host(M, hr.data(), hc.data(), 0);
    iem::KktLayout L = iem::kkt_layout(M, jr, jc, hr, hc, group, 96, 128, 48, true);
    iem::KktPlan P = iem::kkt_plan(L, hr, hc, jr, jc);
    kkt_fill_info(L, info);
    auto dup = [](const void *src, size_t bytes) { void *p = std::malloc(std::max<size_t>(bytes, 8)); if (bytes) std::memcpy(p, src, bytes); return p; };
    if (out_blk) *out_blk = (int64_t *)dup(L.blk.data(), L.blk.size() * 8);
    if (out_loc) *out_loc = (int64_t *)dup(L.loc.data(), L.loc.size() * 8);
    if (out_rows) *out_rows = (int32_t *)dup(L.rowsR.data(), L.rowsR.size() * 4);
    if (out_cols) *out_cols = (int32_t *)dup(L.colsC.data(), L.colsC.size() * 4);
    if (out_dest) *out_dest = (int64_t *)dup(P.dest.data(), P.dest.size() * 8);
    if (out_seg) *out_seg = (uint32_t *)dup(P.seg.data(), P.seg.size() * 4);
    if (out_perm) *out_perm = (uint32_t *)dup(P.perm.data(), P.perm.size() * 4);
    if (out_n_dest) *out_n_dest = (int64_t)P.dest.size();
    if (out_n_perm) *out_n_perm = (int64_t)P.perm.size();
  } catch (const std::exception &e) {
    return fail(IEM_E_ARG, e.what());
  }
  return IEM_OK;
}

int iem_kkt_create(iem_model *m, int group, iem_kkt **out) {
  if (!m || !out) return fail(IEM_E_ARG, "null argument");
  if (m->sharded) return fail(IEM_E_ARG, "iem_kkt_create: a sharded handle holds one rank's window; the chain solver wants the whole model");
  DevGuard dg_(m->device);
  struct Drop { void operator()(iem_kkt *p) const { iem_kkt_destroy(p); } };     // (a failed create frees what it had uploaded)
  std::unique_ptr<iem_kkt, Drop> k(new iem_kkt);
  k->m = m;
  try {
    const iem::Model &M = m->model;
    std::vector<int64_t> jr((size_t)M.nnzj), jc((size_t)M.nnzj), hr((size_t)M.nnzh), hc((size_t)M.nnzh);
    jac_structure_host(M, jr.data(), jc.data(), 0);
    if (m->opt.hess_merge) return fail(IEM_E_ARG, "iem_kkt_create: the merged Hessian layout is not supported here");
    hess_structure_host(M, hr.data(), hc.data(), 0);
    k->L = iem::kkt_layout(M, jr, jc, hr, hc, group, 96, 128, 48, true);
    if (!kkt_fits(k->L.nb, k->L.ne, k->L.nc)) return fail(IEM_E_ARG, "chain KKT: the tiles of a block do not fit the LDS of a CU");
    iem::KktPlan P = iem::kkt_plan(k->L, hr, hc, jr, jc);
    k->n_dest = (int64_t)P.dest.size(); k->n_h = P.n_h; k->n_j = P.n_j;
    int rc = kkt_module(m, k->L.nb, k->L.ne, k->L.nc, &k->km);
    if (rc) return rc;
    const iem::KktLayout &L = k->L;
    std::vector<long long> dest(P.dest.begin(), P.dest.end()), on, pos, border, bloc;
    for (int64_t u = 0; u < L.nvar + L.ncon; ++u) {
      if (L.blk[(size_t)u] >= 0) { on.push_back(u); pos.push_back(L.blk[(size_t)u] * L.nb + L.loc[(size_t)u]); }
      else { border.push_back(u); bloc.push_back(L.hubs ? L.hub_of[(size_t)u] : L.loc[(size_t)u]); }
    }
    k->n_on = (int64_t)on.size();
    if ((rc = kkt_upload(&k->d_dest, dest)) || (rc = kkt_upload(&k->d_seg, P.seg)) || (rc = kkt_upload(&k->d_perm, P.perm)) || (rc = kkt_upload(&k->d_on, on)) ||
        (rc = kkt_upload(&k->d_pos, pos)) || (rc = kkt_upload(&k->d_border, border)) || (rc = kkt_upload(&k->d_bloc, bloc)) ||
        (rc = kkt_upload(&k->d_rows, L.rowsR)) || (rc = kkt_upload(&k->d_cols, L.colsC)))
      return rc;
    const int64_t S = L.S, nb = L.nb, ne = L.ne, nc = L.nc;
    HIP_TRY(hipMalloc((void **)&k->d_flat, (size_t)L.total() * 8));
    HIP_TRY(hipMalloc((void **)&k->d_BR, (size_t)std::max<int64_t>(S * nc * nc, 1) * 8));
    HIP_TRY(hipMalloc((void **)&k->d_Z, (size_t)std::max<int64_t>(S * nb * ne, 1) * 8));
    HIP_TRY(hipMalloc((void **)&k->d_Gp, (size_t)std::max<int64_t>(S * ne * ne, 1) * 8));
    HIP_TRY(hipMalloc((void **)&k->d_r, (size_t)(S * nb) * 8));
    HIP_TRY(hipMalloc((void **)&k->d_z, (size_t)(S * nb) * 8));
    HIP_TRY(hipMalloc((void **)&k->d_rBp, (size_t)std::max<int64_t>(S * ne, 1) * 8));
    HIP_TRY(hipMalloc((void **)&k->d_xB, (size_t)std::max<int64_t>(ne, 1) * 8));
    HIP_TRY(hipMalloc((void **)&k->d_part, (size_t)(513 * std::max<int64_t>(ne * ne, 1)) * 8));      // kkt_colsum_host: 512 partial rows + the result
    HIP_TRY(hipMalloc((void **)&k->d_info, 32));
    if (L.hubs && (rc = hub_alloc(k.get()))) return rc;
  } catch (const std::exception &e) {
    return fail(IEM_E_ARG, e.what());
  }
  *out = k.release();
  return IEM_OK;
}

int iem_kkt_destroy(iem_kkt *k) {
  if (!k) return IEM_OK;
  DevGuard dg_(k->m->device);
  hub_free(k);
  for (void *p : {(void *)k->d_flat, (void *)k->d_BR, (void *)k->d_Z, (void *)k->d_Gp, (void *)k->d_r, (void *)k->d_z, (void *)k->d_rBp, (void *)k->d_xB, (void *)k->d_part,
                  (void *)k->d_rows, (void *)k->d_cols, (void *)k->d_dest, (void *)k->d_on, (void *)k->d_pos, (void *)k->d_border, (void *)k->d_bloc, (void *)k->d_info,
                  (void *)k->d_seg, (void *)k->d_perm})
    if (p) hipFree(p);
  delete k;
  return IEM_OK;
}

int iem_kkt_info(const iem_kkt *k, iem_kkt_info_t *out) {
  if (!k || !out) return fail(IEM_E_ARG, "null argument");
  const iem::KktLayout &L = k->L;
  kkt_fill_info(L, out);
  return IEM_OK;
}

int iem_kkt_layout(const iem_kkt *k, int64_t *h_blk, int64_t *h_loc, int32_t *h_rows, int32_t *h_cols) {
  if (!k) return fail(IEM_E_ARG, "null argument");
  const iem::KktLayout &L = k->L;
  if (h_blk) std::copy(L.blk.begin(), L.blk.end(), h_blk);
  if (h_loc) std::copy(L.loc.begin(), L.loc.end(), h_loc);
  if (h_rows) std::copy(L.rowsR.begin(), L.rowsR.end(), h_rows);
  if (h_cols) std::copy(L.colsC.begin(), L.colsC.end(), h_cols);
  return IEM_OK;
}

int iem_kkt_assemble(iem_kkt *k, const double *d_hess, const double *d_jac, const double *d_sigma, double delta_w, double delta_c) {
  if (!k || (!d_hess && k->n_h) || (!d_jac && k->n_j)) return fail(IEM_E_ARG, "null argument");
  iem_model *m = k->m;
  DevGuard dg_(m->device);
  const iem::KktLayout &L = k->L;
  HIP_TRY(hipMemsetAsync(k->d_flat, 0, (size_t)L.total() * 8, m->stream));
  struct { double *flat; const long long *dest; const unsigned *seg, *perm; const double *hess, *jac, *sigma; double dw, dc; long long n_dest, n_h, n_j, n_var, n_con; } A{
      k->d_flat, k->d_dest, k->d_seg, k->d_perm, d_hess, d_jac, d_sigma, delta_w, delta_c, (long long)k->n_dest, (long long)k->n_h, (long long)k->n_j,
      (long long)L.nvar, (long long)L.ncon};
  k->factored = false;
  return kkt_launch_raw(m, k->km->gather, &A, sizeof A, (k->n_dest + 255) / 256, 256);
}

int iem_kkt_factor(iem_kkt *k, int64_t *out_inertia) {
  if (!k) return fail(IEM_E_ARG, "null argument");
  iem_model *m = k->m;
  DevGuard dg_(m->device);
  const iem::KktLayout &L = k->L;
  if (L.hubs) return hub_factor(k, out_inertia);
  const bool chained = L.reach > 0;
  double *D = k->d_flat + L.oD(), *Bt = k->d_flat + L.oB(), *E = k->d_flat + L.oE();
  int rc = iem_kkt_chain_factor(m, L.S, L.nb, L.ne, L.nc, D, chained ? Bt : nullptr, chained ? k->d_BR : nullptr, chained ? k->d_rows : nullptr,
                                chained ? k->d_cols : nullptr, E, k->d_Z, k->d_Gp, (int64_t *)k->d_info, 1e-30);
  if (rc) return rc;
  int64_t neg = 0, doubtful = 0;
  if (L.ne > 0) {      // the border: G - sum_k Gp[k] on the host (ne <= 64)
    std::vector<double> gp, G((size_t)(L.ne * L.ne));
    if ((rc = kkt_colsum_host(k, k->d_Gp, L.S, (int64_t)L.ne * L.ne, gp))) return rc;
    HIP_TRY(hipMemcpy(G.data(), k->d_flat + L.oG(), G.size() * 8, hipMemcpyDeviceToHost));
    k->Gs.resize(G.size());
    for (size_t i = 0; i < G.size(); ++i) k->Gs[i] = G[i] - gp[i];
    if (L.ne <= 64) {
      std::vector<double> ev;
      iem::sym_eigenvalues(k->Gs, L.ne, ev);
      double emax = 0.0;
      for (double e : ev) emax = std::max(emax, std::fabs(e));
      for (double e : ev) {
        // a zero / near-zero eigenvalue is neither sign: reported as DOUBTFUL so that a host doing the usual inertia correction
        // (neg == ncon && doubtful == 0) shifts and factorises again instead of solving with a singular border
        if (std::fabs(e) <= 1e-14 * emax || emax == 0.0) ++doubtful;
        else if (e < 0.0) ++neg;
      }
    } else {      // (a laned 2-D grid's border, up to 128: LDL' pivot signs instead of Jacobi sweeps)
      int64_t bn = 0, bd = 0;
      iem::sym_inertia_ldl(k->Gs, L.ne, bn, bd);
      // the padding's unit diagonal (n_border .. ne) is positive: nothing to subtract
      neg += bn; doubtful += bd;
    }
  }
  long long info[3] = {0, 0, 0};
  HIP_TRY(hipStreamSynchronize(m->stream));
  HIP_TRY(hipMemcpy(info, k->d_info, 24, hipMemcpyDeviceToHost));
  neg += info[0]; doubtful += info[1];
  k->factored = true;
  if (out_inertia) { out_inertia[0] = L.nvar + L.ncon - neg; out_inertia[1] = neg; out_inertia[2] = doubtful; }
  return IEM_OK;
}

int iem_kkt_solve(iem_kkt *k, const double *d_rhs, double *d_sol) {
  if (!k || !d_rhs || !d_sol) return fail(IEM_E_ARG, "null argument");
  if (!k->factored) return fail(IEM_E_ARG, "iem_kkt_solve: no factorisation (iem_kkt_assemble + iem_kkt_factor first)");
  iem_model *m = k->m;
  DevGuard dg_(m->device);
  const iem::KktLayout &L = k->L;
  if (L.hubs) return hub_solve(k, d_rhs, d_sol);
  const bool chained = L.reach > 0;
  const double *Dinv = k->d_flat + L.oD(), *Bt = k->d_flat + L.oB();
  HIP_TRY(hipMemsetAsync(k->d_r, 0, (size_t)(L.S * L.nb) * 8, m->stream));
  struct Mv { double *dst; const double *src; const long long *di, *si; long long n; };
  int rc;
  { Mv A{k->d_r, d_rhs, k->d_pos, k->d_on, (long long)k->n_on}; if ((rc = kkt_launch_raw(m, k->km->move, &A, sizeof A, (k->n_on + 255) / 256, 256))) return rc; }
  if ((rc = iem_kkt_chain_solve(m, L.S, L.nb, L.ne, L.nc, Dinv, chained ? Bt : nullptr, chained ? k->d_BR : nullptr, chained ? k->d_rows : nullptr,
                                chained ? k->d_cols : nullptr, k->d_Z, k->d_r, chained ? k->d_z : nullptr, k->d_rBp, nullptr, 0)))
    return rc;
  if (L.ne > 0) {      // border system on the host: Gs xB = rB - sum_k rBp[k]
    std::vector<double> rbp, rB((size_t)L.ne, 0.0);
    if ((rc = kkt_colsum_host(k, k->d_rBp, L.S, L.ne, rbp))) return rc;
    std::vector<double> rhs_b((size_t)L.n_border);
    std::vector<long long> border((size_t)L.n_border);
    if (L.n_border) {   // (a handful of entries: gathered through a staging vector on the device)
      Mv A{k->d_xB, d_rhs, k->d_bloc, k->d_border, (long long)L.n_border};
      if ((rc = kkt_launch_raw(m, k->km->move, &A, sizeof A, (L.n_border + 255) / 256, 256))) return rc;
      HIP_TRY(hipStreamSynchronize(m->stream));
      HIP_TRY(hipMemcpy(rB.data(), k->d_xB, (size_t)L.n_border * 8, hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < L.ne; ++i) rB[(size_t)i] = (i < L.n_border ? rB[(size_t)i] : 0.0) - rbp[(size_t)i];
    if (!iem::dense_solve(k->Gs, L.ne, rB)) return fail(IEM_E_ARG, "iem_kkt_solve: the border's Schur complement is singular");
    HIP_TRY(hipMemcpyAsync(k->d_xB, rB.data(), (size_t)L.ne * 8, hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));      // (rB is a host temporary)
  }
  if ((rc = iem_kkt_chain_solve(m, L.S, L.nb, L.ne, L.nc, Dinv, chained ? Bt : nullptr, chained ? k->d_BR : nullptr, chained ? k->d_rows : nullptr,
                                chained ? k->d_cols : nullptr, k->d_Z, k->d_r, chained ? k->d_z : nullptr, k->d_rBp, L.ne > 0 ? k->d_xB : nullptr, 1)))
    return rc;
  { Mv A{d_sol, k->d_r, k->d_on, k->d_pos, (long long)k->n_on}; if ((rc = kkt_launch_raw(m, k->km->move, &A, sizeof A, (k->n_on + 255) / 256, 256))) return rc; }
  if (L.n_border) { Mv A{d_sol, k->d_xB, k->d_border, k->d_bloc, (long long)L.n_border}; if ((rc = kkt_launch_raw(m, k->km->move, &A, sizeof A, (L.n_border + 255) / 256, 256))) return rc; }
  return IEM_OK;
}

int iem_tuner_choice(iem_model *m, int kind, const double *d_vals, int *out_choice) {
  if (!m || !out_choice || kind < 0 || kind > 1) return fail(IEM_E_ARG, "bad argument");
  *out_choice = -1;
  if (m->alt.on)
    for (auto &t : m->tune[kind].slot)
      if (t.out == d_vals) *out_choice = t.choice;
  return IEM_OK;
}

int iem_tune(iem_model *m, const double *d_x, const double *d_y, double obj_weight, double *d_jac, double *d_hess) {
  if (!m || !d_x || (d_hess && !d_y)) return fail(IEM_E_ARG, "bad argument");
  if (!m->alt.on) return IEM_OK;
  DevGuard dg_(m->device);
  // Per kind and variant, twice: six warm launches, then ten launches between ONE pair of events (the faster block counts) — the steady state of that
  // kernel into that buffer.  (Events around every single launch, as the implicit tuner of a running solve uses,
  // perturb a pipeline of un-synchronised launches, the large-batch object more than the default: measured
  // 0.112 ms where ten back-to-back launches take 0.090 — profiles/r02_ab_autotune.txt.)
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return fail(IEM_E_HIP, "hipEventCreate"); }
  int rc = IEM_OK;
  for (int which = 0; which < 2 && rc == IEM_OK; ++which) {
    double *out = which ? d_hess : d_jac;
    if (!out) continue;
    const int kind = which ? iem::KK_HESS : iem::KK_JAC;
    float ms[2] = {1e30f, 1e30f};
    for (int round = 0; round < 2 && rc == IEM_OK; ++round)   // default, large batch, default, large batch: the first blocks of a process run on a GPU that is still ramping up
    for (int v = 0; v < 2 && rc == IEM_OK; ++v) {
      auto go = [&]() { return v ? launch_kind_alt(m, kind, d_x, d_y, out, obj_weight) : launch_kind(m, kind, d_x, d_y, out, obj_weight); };
      for (int i = 0; i < 6 && rc == IEM_OK; ++i) rc = go();   // warm: the first launches of a code object run cold
      if (rc == IEM_OK && hipEventRecord(e0, m->stream) != hipSuccess) rc = fail(IEM_E_HIP, "hipEventRecord");
      for (int i = 0; i < 10 && rc == IEM_OK; ++i) rc = go();
      float t = 0.f;
      if (rc == IEM_OK && (hipEventRecord(e1, m->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                           hipEventElapsedTime(&t, e0, e1) != hipSuccess)) rc = fail(IEM_E_HIP, "event timing");
      ms[v] = std::min(ms[v], t);
      if (getenv("IEM_TUNER_LOG")) fprintf(stderr, "iem tuner (iem_tune): kind %d round %d variant %d block %.4f ms per launch\n", kind, round, v, t / 10);
    }
    if (rc != IEM_OK) break;
    iem_model::TuneSet &S = m->tune[which];
    iem_model::Tune *hit = nullptr;
    for (auto &t : S.slot) if (t.out == out) hit = &t;
    if (!hit) { hit = &S.slot[S.next]; S.next = (S.next + 1) % 4; hit->out = out; }
    hit->calls = IEM_TUNE_CALLS;
    hit->choice = ms[1] < 0.98f * ms[0] ? 1 : 0;
    if (getenv("IEM_TUNER_LOG"))
      fprintf(stderr, "iem tuner (iem_tune): kind %d buffer %p: default %.4f ms, large batch %.4f ms -> %s\n", kind, (const void *)out,
              ms[0] / 10, ms[1] / 10, hit->choice ? "large batch" : "default");
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  if (rc == IEM_OK) HIP_TRY(hipStreamSynchronize(m->stream));
  return rc;
}

int iem_time_kernels(iem_model *m, const double *d_x, const double *d_y, double *d_jac, double *d_hess, int iters,
                     double *h_ms_jac, double *h_ms_hess) {
  if (!m || iters <= 0) return fail(IEM_E_ARG, "bad argument");
  DevGuard dg_(m->device);
  // Average launch duration in steady state: the jac/hess pair is enqueued `iters` times back to
  // back (as a solver loop does) with an event pair around EVERY launch, all on the launch
  // stream, and nothing synchronises until the end — the same quantity rocprofv3
  // --kernel-trace --stats reports as the kernel's average duration.
  std::vector<hipEvent_t> ev((size_t)iters * 4);
  for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
  int rc = IEM_OK;
  for (int w = 0; w < 3 && !rc; ++w) {   // warm-up pairs
    rc = iem_jac_coord(m, d_x, d_jac);
    if (!rc) rc = iem_hess_coord(m, d_x, d_y, 1.0, d_hess);
  }
  for (int it = 0; it < iters && !rc; ++it) {
    HIP_TRY(hipEventRecord(ev[4 * it + 0], m->stream));
    rc = iem_jac_coord(m, d_x, d_jac);
    HIP_TRY(hipEventRecord(ev[4 * it + 1], m->stream));
    HIP_TRY(hipEventRecord(ev[4 * it + 2], m->stream));
    if (!rc) rc = iem_hess_coord(m, d_x, d_y, 1.0, d_hess);
    HIP_TRY(hipEventRecord(ev[4 * it + 3], m->stream));
  }
  HIP_TRY(hipStreamSynchronize(m->stream));
  double tj = 0.0, th = 0.0;
  for (int it = 0; it < iters && !rc; ++it) {
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, ev[4 * it + 0], ev[4 * it + 1]));
    HIP_TRY(hipEventElapsedTime(&b, ev[4 * it + 2], ev[4 * it + 3]));
    tj += a; th += b;
  }
  for (auto &e : ev) hipEventDestroy(e);
  if (rc) return rc;
  *h_ms_jac = tj / iters;
  *h_ms_hess = th / iters;
  return IEM_OK;
}

}  // extern "C"
