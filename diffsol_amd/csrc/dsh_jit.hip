// Run-time-compiled models of libdiffsol_hip.so (gfx950): dsh_model_compile / dsh_model_release and the per-model module cache.
//
// The reference compiles a DiffSL model to host machine code at run time (Cranelift / LLVM back ends of the external `diffsl` crate behind
// OdeBuilder::build_from_diffsl, crates/diffsol/src/ode_equations/diffsl.rs) and calls it through function pointers.  The device-side equivalent:
// the generated model (diffsol_amd/host/diffsl.hpp: `struct dsh::JitModel` or the jit_* component functions) is compiled by hiprtc TOGETHER WITH
// the library's own kernel templates — the same headers the built-in models are instantiated from — so a user model gets the fused Newton
// kernels, the device-resident integrators and the 1:1 operator kernels, not a slower generic path.  One hiprtc program per (model, kernel family),
// compiled on first use and cached for the life of the model.
#include <dlfcn.h>
#include <glob.h>
#include <unistd.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <cstring>

#include <atomic>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>

#include "dsh_jit.hpp"

using namespace dsh;

namespace {

struct JitModule {
  std::vector<char> code;
  std::map<std::string, std::string> lowered;  // name expression -> symbol
  hipModule_t module = nullptr;
  std::map<std::string, hipFunction_t> functions;
};
struct JitModelRec {
  JitInfo info;
  std::string source;
  std::map<std::string, std::unique_ptr<JitModule>> modules;  // by header + group key
  // a static model with 5 <= n <= 8 has no device-resident integrator of its own: the same model in the run-time-sized form (dsh_model_set_member_twin_source), compiled
  // on the first per-member request (dsh_model_member_twin) — the wavefront-per-member kernels take it
  std::string member_twin_source;
  int64_t member_twin_dims[4] = {0, 0, 0, 0};  // n, nparams, nroots, nout
  int member_twin = -1;
  bool member_twin_failed = false;
};

std::mutex g_mu;
std::map<int, std::unique_ptr<JitModelRec>> g_models;
int g_next_id = DSH_MODEL_JIT_BASE;

std::string lib_dir() {
  Dl_info info;
  if (dladdr((const void*)&dsh_model_compile, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    size_t k = p.rfind('/');
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
  }
  return ".";
}

std::vector<std::string> include_options() {
  std::vector<std::string> opts;
  const char* over = std::getenv("DSH_JIT_INCLUDE");  // colon-separated directories holding dsh_*.hpp and diffsol_detpow.h
  if (over && *over) {
    std::string s(over);
    size_t a = 0;
    while (a <= s.size()) {
      size_t b = s.find(':', a);
      if (b == std::string::npos) b = s.size();
      if (b > a) opts.push_back("-I" + s.substr(a, b - a));
      a = b + 1;
    }
  } else {
    const std::string d = lib_dir();  // <pkg>/lib -> <pkg>/csrc and <repo>/include
    opts.push_back("-I" + d + "/../csrc");
    opts.push_back("-I" + d + "/../../include");
  }
  const char* rocm = std::getenv("ROCM_PATH");
  const std::string r = rocm && *rocm ? rocm : "/opt/rocm";
  opts.push_back("-I" + r + "/include");
  glob_t g;
  if (glob((r + "/lib/llvm/lib/clang/*/include").c_str(), 0, nullptr, &g) == 0) {
    for (size_t i = 0; i < g.gl_pathc; ++i) opts.push_back(std::string("-I") + g.gl_pathv[i]);
    globfree(&g);
  }
  return opts;
}

// ---- on-disk cache of compiled modules (DSH_JIT_CACHE=<dir>, default <package>/_jit_cache next to the library; DSH_JIT_CACHE=off disables): key = FNV-1a of the translation
// unit, the options, the requested instantiations and every header of the library the unit can include
uint64_t fnv1a(const void* data, size_t n, uint64_t h) {
  const unsigned char* p = (const unsigned char*)data;
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}
uint64_t headers_fingerprint() {
  static const uint64_t fp = [] {
    uint64_t h = 1469598103934665603ull;
    // the toolchain that generates the code: a ROCm upgrade must not load code objects of the previous compiler
    int rtc_major = 0, rtc_minor = 0, hip_rt = 0;
    (void)hiprtcVersion(&rtc_major, &rtc_minor);
    (void)hipRuntimeGetVersion(&hip_rt);
    const int ver[3] = {rtc_major, rtc_minor, hip_rt};
    h = fnv1a((const char*)ver, sizeof ver, h);
    // every resolved include option (ROCM_PATH, clang resource directory, DSH_JIT_INCLUDE) ...
    std::vector<std::string> pats;
    // (not the two package-relative directories: their CONTENTS are hashed below, and the in-tree cache must stay valid when the tree is copied)
    const std::string own = "-I" + lib_dir();
    for (const std::string& o : include_options()) if (o.rfind(own, 0) != 0) h = fnv1a(o.data(), o.size() + 1, h);
    // ... and the contents of the library's own headers in the directories the compilation will read them from
    const char* over = std::getenv("DSH_JIT_INCLUDE");
    if (over && *over) {
      for (const std::string& o : include_options()) {
        if (o.rfind("-I", 0) != 0 || o.find("/lib/llvm/") != std::string::npos || o == std::string("-I") + (std::getenv("ROCM_PATH") && *std::getenv("ROCM_PATH") ? std::getenv("ROCM_PATH") : "/opt/rocm") + "/include") continue;
        pats.push_back(o.substr(2) + "/dsh_*.hpp");
        pats.push_back(o.substr(2) + "/diffsol_*.h");
      }
    } else {
      const std::string d = lib_dir();
      pats = {d + "/../csrc/*.hpp", d + "/../../include/*.h"};
    }
    for (const std::string& pat : pats) {
      glob_t g;
      if (glob(pat.c_str(), 0, nullptr, &g) != 0) continue;
      for (size_t i = 0; i < g.gl_pathc; ++i) {
        FILE* f = fopen(g.gl_pathv[i], "rb");
        if (!f) continue;
        char buf[65536];
        size_t k;
        while ((k = fread(buf, 1, sizeof buf, f)) > 0) h = fnv1a(buf, k, h);
        fclose(f);
      }
      globfree(&g);
    }
    return h;
  }();
  return fp;
}
std::atomic<long> g_jit_compiles{0};  // modules this process compiled itself (not loaded from the cache): dsh_jit_compile_count
std::string cache_dir() {
  const char* e = std::getenv("DSH_JIT_CACHE");
  if (e && std::string(e) == "off") return "";
  if (e && *e) return e;
  return lib_dir() + "/../_jit_cache";  // next to the library, inside the package tree
}
bool cache_load(const std::string& path, const std::vector<std::string>& group, JitModule* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  bool ok = false;
  uint64_t nnames = 0, sz = 0;
  if (fread(&nnames, 8, 1, f) == 1 && nnames == group.size()) {
    ok = true;
    for (size_t i = 0; i < group.size() && ok; ++i) {
      uint64_t len = 0;
      ok = fread(&len, 8, 1, f) == 1 && len < 4096;
      std::string low(len, '\0');
      ok = ok && (len == 0 || fread(low.data(), 1, len, f) == len);
      if (ok) out->lowered[group[i]] = low;
    }
    ok = ok && fread(&sz, 8, 1, f) == 1 && sz > 0 && sz < (1ull << 31);
    if (ok) { out->code.resize(sz); ok = fread(out->code.data(), 1, sz, f) == sz; }
  }
  fclose(f);
  if (!ok) { out->lowered.clear(); out->code.clear(); }
  return ok;
}
void cache_store(const std::string& dir, const std::string& path, const std::vector<std::string>& group, const JitModule& m) {
  std::error_code ec;
  std::filesystem::create_directories(dir, ec);
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return;
  uint64_t nnames = group.size();
  fwrite(&nnames, 8, 1, f);
  for (const std::string& e : group) {
    const std::string& low = m.lowered.at(e);
    uint64_t len = low.size();
    fwrite(&len, 8, 1, f);
    fwrite(low.data(), 1, len, f);
  }
  uint64_t sz = m.code.size();
  fwrite(&sz, 8, 1, f);
  fwrite(m.code.data(), 1, sz, f);
  fclose(f);
  (void)rename(tmp.c_str(), path.c_str());  // atomic: concurrent processes (one per GPU) may compile the same model
}

int compile_module_uncached(const JitModelRec& rec, const std::string& tu, const char* header, const std::vector<std::string>& group, const std::vector<std::string>& opts,
                            JitModule* out);


// ---- manifest of compile requests (DSH_JIT_RECORD=<file>): every request for a module — served from the cache or compiled — appends (header, name expressions, translation
// unit) to <file>, once per process and request.  dsh_jit_replay compiles such a manifest into the cache without a GPU: the build step replays the committed manifests of
// bench.py and of the GPU tests (diffsol_amd/jit_manifest/), so a fresh box finds every code object it will ask for (first-use compilation inside bench.py is a bug).
// Record layout: "DSHJ" u32 version=1, u64 len + header, u64 ngroup, (u64 len + name)*, u64 len + tu.
void put_str(std::string* b, const std::string& s) { uint64_t n = s.size(); b->append((const char*)&n, 8); b->append(s); }
void record_request(const std::string& tu, const char* header, const std::vector<std::string>& group) {
  const char* path = std::getenv("DSH_JIT_RECORD");
  if (!path || !*path) return;
  uint64_t h = fnv1a(tu.data(), tu.size(), 1469598103934665603ull);
  h = fnv1a(header, strlen(header) + 1, h);
  for (const std::string& e : group) h = fnv1a(e.data(), e.size() + 1, h);
  static std::mutex mu;
  static std::set<uint64_t> seen;
  std::lock_guard<std::mutex> lk(mu);
  if (!seen.insert(h).second) return;
  std::string b("DSHJ");
  uint32_t ver = 1;
  b.append((const char*)&ver, 4);
  put_str(&b, header);
  uint64_t ng = group.size();
  b.append((const char*)&ng, 8);
  for (const std::string& e : group) put_str(&b, e);
  put_str(&b, tu);
  int fd = open(path, O_CREAT | O_WRONLY | O_APPEND, 0644);
  if (fd < 0) return;
  if (flock(fd, LOCK_EX) == 0) { (void)!write(fd, b.data(), b.size()); flock(fd, LOCK_UN); }
  close(fd);
}

int compile_tu(const JitModelRec& rec, const std::string& tu, const char* header, const std::vector<std::string>& group, JitModule* out);

// group_key ending in "#small": the SMALL-ENSEMBLE code object of the banded lane-per-member BDF — below four wavefronts per SIMD-quartet the kernel
// is bound by one wavefront's dependent chain, not by bandwidth: one wavefront per SIMD, loops over the state unrolled 8-fold, doubled streaming chunks
// (profiles/r05_c4_knobs.log: 12.1 -> 10.9 ms and 36.4 -> 33.0 ms at 32 768 members of the battery model; slower at 262 144).  Same source, same bits; the
// environment knobs still override.
int compile_module(const JitModelRec& rec, const char* header, const std::vector<std::string>& group, JitModule* out, const std::string& group_key = std::string()) {
  std::string tu = "#include <hip/hip_runtime.h>\n#include \"diffsol_detpow.h\"\n";
  if (rec.info.form == DSH_JIT_FORM_STATIC_BANDED) {
    const bool small = group_key.size() >= 6 && group_key.compare(group_key.size() - 6, 6, "#small") == 0;
    const char* w = std::getenv("DSH_BANDED_WAVES_PER_EU");  // tuning knob
    // loops over the state components stay rolled (the arrays live in per-lane memory anyway): the 42-state battery model compiles in 13 s instead of 43 s
    // and integrates 262 144 members in 0.245 s instead of 0.308 s.  DSH_BANDED_NOUNROLL=0 unrolls them like the register-resident kernels.
    const char* nu = std::getenv("DSH_BANDED_NOUNROLL");
    if (!(nu && nu[0] == '0')) tu += "#define DSH_NOUNROLL_N 1\n";
    tu += std::string("#define DSH_LANE_BANDED_WAVES_PER_EU ") + (w && *w ? w : (small ? "1" : "3")) + "\n";  // k_bdf_lane_banded, same workload: 0.192 / 0.137 / 0.141 / 0.158 / 0.160 s at 2 / 3 / 4 / 6 / 8
    const char* un = std::getenv("DSH_LANE_BANDED_UNROLL");  // tuning knob
    if (un && *un) tu += std::string("#define DSH_LANE_BANDED_UNROLL ") + un + "\n";
    else if (small) tu += "#define DSH_LANE_BANDED_UNROLL 8\n";
    const char* cs = std::getenv("DSH_LANE_BANDED_CHUNK_SCALE");  // tuning knob
    if (cs && *cs) tu += std::string("#define DSH_LANE_BANDED_CHUNK_SCALE ") + cs + "\n";
    else if (small) tu += "#define DSH_LANE_BANDED_CHUNK_SCALE 2\n";
    if (const char* pd = std::getenv("DSH_LANE_BANDED_PAD")) if (*pd) tu += std::string("#define DSH_LANE_BANDED_PAD ") + pd + "\n";  // tuning knob: extra doubles in the per-lane frame
    tu += std::string("#define DSH_ADAPTIVE_WAVES_PER_EU ") + (w && *w ? w : "4") + "\n";  // measured on the 42-state battery model, 262 144 members: 0.45 / 0.37 / 0.39 / 0.31 / 0.33 / 0.33 s at 1 / 2 / 3 / 4 / 6 / 8
  }
  tu += rec.source;
  tu += std::string("\n#include \"") + header + "\"\n";
  record_request(tu, header, group);
  return compile_tu(rec, tu, header, group, out);
}

int compile_tu(const JitModelRec& rec, const std::string& tu, const char* header, const std::vector<std::string>& group, JitModule* out) {
  // same code generation switches as the library's own objects (Makefile): no FMA contraction, so the arithmetic order in the source is the arithmetic
  std::vector<std::string> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"};
  const std::string dir = cache_dir();
  std::string path;
  if (!dir.empty()) {
    uint64_t h = headers_fingerprint();
    h = fnv1a(tu.data(), tu.size(), h);
    for (const std::string& o : opts) h = fnv1a(o.data(), o.size() + 1, h);
    for (const std::string& e : group) h = fnv1a(e.data(), e.size() + 1, h);
    char name[64];
    snprintf(name, sizeof name, "/%016llx.hsaco", (unsigned long long)h);
    path = dir + name;
    const bool hit = cache_load(path, group, out);
    if (std::getenv("DSH_JIT_DEBUG")) fprintf(stderr, "dsh_jit: %s %s (headers fingerprint %016llx, %s)\n", hit ? "hit " : "MISS", path.c_str(), (unsigned long long)headers_fingerprint(), header);
    if (hit) return DSH_OK;
  }
  for (const std::string& o : include_options()) opts.push_back(o);
  // One process per GPU: eight ranks that meet the same model on a cold cache would each spend the minutes hiprtc takes on it.  An exclusive lock on
  // <entry>.lock around "look again, compile, store" lets the first one compile and the others load what it stored (the store itself was already
  // atomic: temporary file + rename).  A cache directory that cannot be locked (read-only tree) just compiles.
  int lock_fd = -1;
  if (!path.empty()) {
    std::error_code ec;
    std::filesystem::create_directories(dir, ec);
    // The lock file is removed by its holder WHILE it holds the lock, so whoever acquires a lock checks that the file it locked is still the one on disk
    // (same inode) and otherwise starts over: a waiter that wakes up on an unlinked inode never shares the critical section with a newcomer on a fresh file.
    const std::string lock_path = path + ".lock";
    for (int attempt = 0; attempt < 64; ++attempt) {
      lock_fd = open(lock_path.c_str(), O_CREAT | O_RDWR, 0644);
      if (lock_fd < 0) break;
      if (flock(lock_fd, LOCK_EX) != 0) { close(lock_fd); lock_fd = -1; break; }
      struct stat held, disk;
      if (fstat(lock_fd, &held) == 0 && stat(lock_path.c_str(), &disk) == 0 && held.st_ino == disk.st_ino && held.st_dev == disk.st_dev) break;
      flock(lock_fd, LOCK_UN); close(lock_fd); lock_fd = -1;  // the previous holder removed this file: lock the current one
    }
    if (lock_fd >= 0 && cache_load(path, group, out)) { (void)unlink(lock_path.c_str()); flock(lock_fd, LOCK_UN); close(lock_fd); return DSH_OK; }
  }
  int rc = compile_module_uncached(rec, tu, header, group, opts, out);
  if (rc == DSH_OK) g_jit_compiles.fetch_add(1);
  if (rc == DSH_OK && !path.empty()) cache_store(dir, path, group, *out);
  if (lock_fd >= 0) { (void)unlink((path + ".lock").c_str()); flock(lock_fd, LOCK_UN); close(lock_fd); }
  return rc;
}

int compile_module_uncached(const JitModelRec& rec, const std::string& tu, const char* header, const std::vector<std::string>& group, const std::vector<std::string>& opts,
                            JitModule* out) {
  (void)rec;
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, tu.c_str(), "dsh_jit_model.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    set_error("hiprtcCreateProgram failed");
    return DSH_E_HIP;
  }
  for (const std::string& e : group) hiprtcAddNameExpression(prog, e.c_str());
  std::vector<const char*> copts;
  for (const std::string& o : opts) copts.push_back(o.c_str());
  hiprtcResult r = hiprtcCompileProgram(prog, (int)copts.size(), copts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, log.data());
    if (log.size() > 4000) log.resize(4000);
    set_error(std::string("model compilation failed (") + hiprtcGetErrorString(r) + ", " + header + "):\n" + log);
    hiprtcDestroyProgram(&prog);
    return DSH_E_INVALID;
  }
  for (const std::string& e : group) {
    const char* low = nullptr;
    if (hiprtcGetLoweredName(prog, e.c_str(), &low) != HIPRTC_SUCCESS || !low) {
      set_error("hiprtcGetLoweredName failed for " + e);
      hiprtcDestroyProgram(&prog);
      return DSH_E_HIP;
    }
    out->lowered[e] = low;
  }
  size_t sz = 0;
  hiprtcGetCodeSize(prog, &sz);
  out->code.resize(sz);
  hiprtcGetCode(prog, out->code.data());
  hiprtcDestroyProgram(&prog);
  return DSH_OK;
}

JitModelRec* find_model(int model) {
  auto it = g_models.find(model);
  return it == g_models.end() ? nullptr : it->second.get();
}

const char* ops_header(int form) { return form == DSH_JIT_FORM_STATIC ? "dsh_model_kernels.hpp" : "dsh_jit_dyn_kernels.hpp"; }

}  // namespace

namespace dsh {

const std::vector<std::string>& jit_static_op_names() {
  static const std::vector<std::string> v = {"dsh::k_static_model<dsh::JitModel, dsh::Op::Rhs>",        "dsh::k_static_model<dsh::JitModel, dsh::Op::JacMul>",
                                             "dsh::k_static_model<dsh::JitModel, dsh::Op::Jacobian>",   "dsh::k_static_model<dsh::JitModel, dsh::Op::MassGemv>",
                                             "dsh::k_static_model<dsh::JitModel, dsh::Op::MassMatrix>", "dsh::k_static_model<dsh::JitModel, dsh::Op::Init>",
                                             "dsh::k_static_model<dsh::JitModel, dsh::Op::Root>",       "dsh::k_static_model<dsh::JitModel, dsh::Op::Out>",
                                             "dsh::k_static_model<dsh::JitModel, dsh::Op::RhsSens>",    "dsh::k_static_model<dsh::JitModel, dsh::Op::InitSens>",
                                             "dsh::k_static_model<dsh::JitModel, dsh::Op::Reset>"};
  return v;
}

const JitInfo* jit_info(int model) {
  // a copy taken under the registry lock, per calling thread: the pointer stays valid if another thread releases the model meanwhile
  thread_local JitInfo copy;
  std::lock_guard<std::mutex> lk(g_mu);
  JitModelRec* rec = find_model(model);
  if (!rec) { set_error("unknown run-time-compiled model id " + std::to_string(model)); return nullptr; }
  copy = rec->info;
  return &copy;
}

int jit_get_function(int model, const char* header, const std::string& group_key, const std::vector<std::string>& group, const std::string& name,
                     hipFunction_t* f) {
  std::lock_guard<std::mutex> lk(g_mu);
  JitModelRec* rec = find_model(model);
  if (!rec) { set_error("unknown run-time-compiled model id " + std::to_string(model)); return DSH_E_INVALID; }
  const std::string key = std::string(header) + "|" + group_key;
  auto& slot = rec->modules[key];
  if (!slot) {
    auto m = std::make_unique<JitModule>();
    int rc = compile_module(*rec, header, group, m.get(), group_key);
    if (rc != DSH_OK) { rec->modules.erase(key); return rc; }
    slot = std::move(m);
  }
  JitModule& m = *slot;
  if (!m.module) DSH_HIP_CHECK(hipModuleLoadData(&m.module, m.code.data()));
  auto it = m.functions.find(name);
  if (it == m.functions.end()) {
    auto low = m.lowered.find(name);
    const std::string sym = low == m.lowered.end() ? name : low->second;
    hipFunction_t fn = nullptr;
    DSH_HIP_CHECK(hipModuleGetFunction(&fn, m.module, sym.c_str()));
    it = m.functions.emplace(name, fn).first;
  }
  *f = it->second;
  return DSH_OK;
}

}  // namespace dsh

extern "C" {

int dsh_model_compile(const char* source, int form, int64_t n, int64_t nparams, int64_t nroots, int64_t nout, int has_mass, int* model_id) {
  DSH_REQUIRE(source != nullptr && model_id != nullptr, "null argument");
  DSH_REQUIRE(form == DSH_JIT_FORM_STATIC || form == DSH_JIT_FORM_DYNAMIC || form == DSH_JIT_FORM_STATIC_BANDED, "unknown model form");
  DSH_REQUIRE(n >= 1 && nparams >= 1 && nroots >= 0 && nout >= 0, "bad model dimensions");
  if (form == DSH_JIT_FORM_STATIC_BANDED) DSH_REQUIRE(n <= 512 && nroots <= 8, "the lane-per-member banded form needs n <= 512 and at most 8 stop conditions (a mass matrix must be diagonal)");
  if (form == DSH_JIT_FORM_STATIC) {
    DSH_REQUIRE(n <= 8, "the register-resident form needs n <= 8");
    DSH_REQUIRE(nroots <= 1, "the register-resident form supports at most one root function; use the dynamic form");
  }
  auto rec = std::make_unique<JitModelRec>();
  rec->info.form = form; rec->info.n = n; rec->info.np = nparams; rec->info.nroots = nroots; rec->info.nout = nout; rec->info.has_mass = has_mass ? 1 : 0;
  rec->source = source;
  rec->info.has_sens = rec->source.find("DSH_JIT_HAS_SENS") != std::string::npos ? 1 : 0;
  rec->info.has_reset = rec->source.find("DSH_JIT_HAS_RESET") != std::string::npos ? 1 : 0;
  if (rec->source.find("DSH_JIT_JAC_NNZ") != std::string::npos) {
    const size_t at = rec->source.find("constexpr int kJitJacNnz = ");
    if (at != std::string::npos) rec->info.jac_nnz = std::atoll(rec->source.c_str() + at + 27);
  }
  // the stated dimensions must be the ones the source was generated with
  if (form != DSH_JIT_FORM_DYNAMIC)
    rec->source += "\nstatic_assert(dsh::JitModel::N == " + std::to_string(n) + " && dsh::JitModel::NP == " + std::to_string(nparams) + " && dsh::JitModel::NROOTS == " +
                   std::to_string(nroots) + " && dsh::JitModel::NOUT == " + std::to_string(nout) + " && dsh::JitModel::HAS_MASS == " + (has_mass ? "true" : "false") +
                   ", \"dsh_model_compile: dimensions do not match the model source\");\n";
  else
    rec->source += "\nstatic_assert(dsh::kJitN == " + std::to_string(n) + " && dsh::kJitNP == " + std::to_string(nparams) + " && dsh::kJitNRoots == " + std::to_string(nroots) +
                   " && dsh::kJitNOut == " + std::to_string(nout) + " && dsh::kJitHasMass == " + (has_mass ? "true" : "false") +
                   ", \"dsh_model_compile: dimensions do not match the model source\");\n";
  // compile the operator kernels now: a model that does not compile is rejected here, not at the first launch.  (The banded lane-per-member form only
  // exists for the device-resident BDF, compiled on first use: its operators are those of the run-time-sized twin it belongs to.)
  if (form != DSH_JIT_FORM_STATIC_BANDED) {
    auto m = std::make_unique<JitModule>();
    const std::vector<std::string> none;
    int rc = compile_module(*rec, ops_header(form), form == DSH_JIT_FORM_STATIC ? jit_static_op_names() : none, m.get());
    if (rc != DSH_OK) return rc;
    rec->modules[std::string(ops_header(form)) + "|ops"] = std::move(m);
  }
  std::lock_guard<std::mutex> lk(g_mu);
  *model_id = g_next_id++;
  g_models[*model_id] = std::move(rec);
  return DSH_OK;
}

int dsh_model_set_band(int model_id, int jac_kl, int jac_ku, int mass_kl, int mass_ku) {
  std::lock_guard<std::mutex> lk(g_mu);
  JitModelRec* rec = find_model(model_id);
  if (!rec) { set_error("dsh_model_set_band: unknown model id"); return DSH_E_INVALID; }
  rec->info.jac_kl = jac_kl; rec->info.jac_ku = jac_ku; rec->info.mass_kl = mass_kl; rec->info.mass_ku = mass_ku;
  return DSH_OK;
}

// The banded lane-per-member form of a built-in run-time-sized model (dsh_models_lane.hpp), created on first request and kept for the life of the process.
// -1 if the model has none (not a registry model with n <= 64, identity mass, declared bandwidth <= 4, at most 2 stop conditions and parameters that fit).
// largest built-in banded model that gets a lane-per-member twin: 512 (the kernels keep their arrays in per-lane scratch, ~230 bytes per state for BDF, and
// the hardware's scratch wave size allows 128 KB per lane); DSH_LANE_TWIN_MAX_N lowers it (64: the round-1 limit).  hiprtc needs ~3 minutes for the
// BDF kernel at n = 512 (seconds at n = 100), once per size: the code object is cached.
static int64_t lane_twin_max_n() {
  const char* e = std::getenv("DSH_LANE_TWIN_MAX_N");
  const int64_t v = e && *e ? std::atoll(e) : 512;
  return v < 8 ? 8 : (v > 512 ? 512 : v);
}
int dsh_model_lane_twin(int model, int64_t size) {
  if (is_jit_model(model)) return dsh_model_twin(model);
  static std::map<std::pair<int, int64_t>, int> twins;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  auto it = twins.find({model, size});
  if (it != twins.end()) return it->second;
  int id = -1;
  int64_t n = 0, np = 0, nroots = 0;
  int has_mass = 0, jl = -1, ju = -1, ml = -1, mu2 = -1;
  if (dsh_model_info(model, size, &n, &np, &has_mass, &nroots) == DSH_OK && dsh_model_band(model, size, &jl, &ju, &ml, &mu2) == DSH_OK && !dsh_model_has_fused(model, size) &&
      n > 8 && n <= lane_twin_max_n() && !has_mass && jl >= 0 && ju >= 0 && std::max(jl, ju) <= 4 && nroots <= 2 && np >= 1 && np <= 8) {
    const int k = std::max(1, std::max(jl, ju));
    const std::string src = "#include \"dsh_models_lane.hpp\"\nnamespace dsh { using JitModel = DynLane<" + std::to_string(model) + ", " + std::to_string(n) + ", " + std::to_string(np) +
                            ", " + std::to_string(nroots) + ", " + std::to_string(k) + ">; }\n";
    if (dsh_model_compile(src.c_str(), DSH_JIT_FORM_STATIC_BANDED, n, np, nroots, 0, 0, &id) != DSH_OK) id = -1;
  }
  twins[{model, size}] = id;
  return id;
}

int dsh_model_set_twin(int model_id, int twin_id) {
  std::lock_guard<std::mutex> lk(g_mu);
  JitModelRec* rec = find_model(model_id);
  if (!rec || (twin_id >= 0 && !find_model(twin_id))) { set_error("dsh_model_set_twin: unknown model id"); return DSH_E_INVALID; }
  rec->info.twin = twin_id;
  return DSH_OK;
}
int dsh_model_twin(int model_id) {
  if (!is_jit_model(model_id)) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  JitModelRec* rec = find_model(model_id);
  return rec ? rec->info.twin : -1;
}

// The run-time-sized form of a static model, for per-member device-resident solves (the static form has them up to n = 4 only): the source is kept, the compilation
// happens at the first request.  DiffSL front ends (diffsol_amd/diffsl.py, host/diffsol_c.cpp, rust/diffsol-hip/src/diffsl.rs) call this for static models with n >= 5.
int dsh_model_set_member_twin_source(int model_id, const char* source, int64_t n, int64_t nparams, int64_t nroots, int64_t nout) {
  DSH_REQUIRE(source != nullptr && n >= 1 && nparams >= 1 && nroots >= 0 && nout >= 0, "bad arguments");
  std::lock_guard<std::mutex> lk(g_mu);
  JitModelRec* rec = find_model(model_id);
  if (!rec) { set_error("dsh_model_set_member_twin_source: unknown model id"); return DSH_E_INVALID; }
  DSH_REQUIRE(rec->info.form == DSH_JIT_FORM_STATIC && rec->info.n == n && rec->info.np == nparams, "the twin must be the same model (static form, same dimensions)");
  const int old_twin = rec->member_twin;  // a twin compiled from the previous source is released, not forgotten
  rec->member_twin_source = source;
  rec->member_twin_dims[0] = n; rec->member_twin_dims[1] = nparams; rec->member_twin_dims[2] = nroots; rec->member_twin_dims[3] = nout;
  rec->member_twin = -1; rec->member_twin_failed = false;
  if (old_twin >= 0) {
    auto it = g_models.find(old_twin);
    if (it != g_models.end()) {
      for (auto& kv : it->second->modules)
        if (kv.second && kv.second->module) (void)hipModuleUnload(kv.second->module);
      g_models.erase(it);
    }
  }
  return DSH_OK;
}
// -1: the model has no such twin (not a static run-time-compiled model with a registered source, or the source did not compile: dsh_last_error says why)
int dsh_model_member_twin(int model_id) {
  if (!is_jit_model(model_id)) return -1;
  std::string src;
  int64_t d[4];
  int has_mass = 0, band[4];
  {
    std::lock_guard<std::mutex> lk(g_mu);
    JitModelRec* rec = find_model(model_id);
    if (!rec || rec->member_twin_source.empty() || rec->member_twin_failed) return -1;
    if (rec->member_twin >= 0) return rec->member_twin;
    src = rec->member_twin_source;
    for (int k = 0; k < 4; ++k) d[k] = rec->member_twin_dims[k];
    has_mass = rec->info.has_mass;
    band[0] = rec->info.jac_kl; band[1] = rec->info.jac_ku; band[2] = rec->info.mass_kl; band[3] = rec->info.mass_ku;
  }
  int id = -1;
  const int rc = dsh_model_compile(src.c_str(), DSH_JIT_FORM_DYNAMIC, d[0], d[1], d[2], d[3], has_mass, &id);  // takes the registry lock itself
  // the registry lock was dropped for the compilation: the parent may have vanished, another first request may have won, the source may have been replaced —
  // the twin compiled here is then released instead of leaked
  int loser = -1, ret = -1;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    JitModelRec* rec = find_model(model_id);
    if (rc != DSH_OK) { if (rec && rec->member_twin < 0 && rec->member_twin_source == src) rec->member_twin_failed = true; return rec && rec->member_twin >= 0 ? rec->member_twin : -1; }
    if (!rec || rec->member_twin_source != src) { loser = id; ret = -1; }
    else if (rec->member_twin >= 0) { loser = id; ret = rec->member_twin; }
    else {
      if (JitModelRec* tw = find_model(id)) { tw->info.jac_kl = band[0]; tw->info.jac_ku = band[1]; tw->info.mass_kl = band[2]; tw->info.mass_ku = band[3]; }
      rec->member_twin = id;
      ret = id;
    }
  }
  if (loser >= 0) (void)dsh_model_release(loser);
  return ret;
}

int dsh_model_release(int model_id) {
  {
    int tw = -1;
    {
      std::lock_guard<std::mutex> lk(g_mu);
      JitModelRec* rec = find_model(model_id);
      if (rec) { tw = rec->member_twin; rec->member_twin = -1; }
    }
    if (tw >= 0) (void)dsh_model_release(tw);
  }
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_models.find(model_id);
  if (it == g_models.end()) { set_error("dsh_model_release: unknown model id"); return DSH_E_INVALID; }
  for (auto& kv : it->second->modules)
    if (kv.second && kv.second->module) (void)hipModuleUnload(kv.second->module);
  g_models.erase(it);
  return DSH_OK;
}

// compile (not load) one kernel family of a run-time-compiled model: 0 operators, 1 fused Newton kernels, 2 resident BDF, 3 resident SDIRK.
// Needs no GPU; used by the build check and by callers that want to pay the compilation before the first solve.
int64_t dsh_jit_compile_count(void) { return (int64_t)g_jit_compiles.load(); }

// Compile the requests of a manifest written under DSH_JIT_RECORD into the on-disk cache (no GPU needed, nothing is loaded).  Requests i with i % nparts == part are this
// call's share (the build step runs one process per core).  *requests = records seen, *compiled = modules this call had to compile (the rest were in the cache already).
int dsh_jit_replay(const char* manifest_path, int part, int nparts, int64_t* requests, int64_t* compiled) {
  DSH_REQUIRE(manifest_path != nullptr && nparts >= 1 && part >= 0 && part < nparts, "dsh_jit_replay: bad arguments");
  FILE* f = fopen(manifest_path, "rb");
  if (!f) { set_error(std::string("dsh_jit_replay: cannot open ") + manifest_path); return DSH_E_INVALID; }
  auto get_str = [&](std::string* out) {
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1 || n > (1ull << 30)) return false;
    out->resize(n);
    return n == 0 || fread(out->data(), 1, n, f) == n;
  };
  int64_t seen = 0, done = 0;
  int rc = DSH_OK;
  const long before = g_jit_compiles.load();
  std::set<uint64_t> uniq;
  for (;;) {
    char magic[4];
    if (fread(magic, 1, 4, f) != 4) break;  // end of file
    uint32_t ver = 0;
    std::string header, tu;
    uint64_t ng = 0;
    bool ok = memcmp(magic, "DSHJ", 4) == 0 && fread(&ver, 4, 1, f) == 1 && ver == 1 && get_str(&header) && fread(&ng, 8, 1, f) == 1 && ng < 4096;
    std::vector<std::string> group(ok ? ng : 0);
    for (size_t i = 0; ok && i < group.size(); ++i) ok = get_str(&group[i]);
    ok = ok && get_str(&tu);
    if (!ok) { set_error(std::string("dsh_jit_replay: malformed record in ") + manifest_path); rc = DSH_E_INVALID; break; }
    uint64_t h = fnv1a(tu.data(), tu.size(), 1469598103934665603ull);
    h = fnv1a(header.data(), header.size() + 1, h);
    for (const std::string& e : group) h = fnv1a(e.data(), e.size() + 1, h);
    if (!uniq.insert(h).second) continue;  // several processes recorded the same request
    const int64_t idx = seen++;
    if (idx % nparts != part) continue;
    JitModelRec dummy;
    JitModule m;
    // a request that does not compile is skipped: the recorded runs contain the tests' deliberately malformed models (rejected by dsh_model_compile at run time too)
    if (compile_tu(dummy, tu, header.c_str(), group, &m) == DSH_OK) ++done;
  }
  fclose(f);
  (void)done;
  if (requests) *requests = seen;
  if (compiled) *compiled = (int64_t)(g_jit_compiles.load() - before);
  return rc;
}

int dsh_model_precompile(int model_id, int family) {
  std::lock_guard<std::mutex> lk(g_mu);
  JitModelRec* rec = find_model(model_id);
  if (!rec) { set_error("dsh_model_precompile: unknown model id"); return DSH_E_INVALID; }
  struct Unit { const char* header; std::string key; std::vector<std::string> group; };
  std::vector<Unit> units;
  const bool st = rec->info.form == DSH_JIT_FORM_STATIC;
  if (rec->info.form == DSH_JIT_FORM_STATIC_BANDED) {
    if (family != 2 && family != 3) { set_error("dsh_model_precompile: the banded lane-per-member form only has the device-resident integrators (families 2, 3)"); return DSH_E_UNSUPPORTED; }
    // the variants with shared tolerances and per-member control (what the host-side problem uses) are enough to pay the cost up front
    std::vector<std::pair<const char*, std::string>> units;
    if (family == 2) units.push_back({"dsh_lane_banded_kernel.hpp", "dsh::k_bdf_lane_banded<dsh::JitModel, true, false>"});
    else for (int s = 3; s <= 4; ++s) units.push_back({"dsh_sdirk_kernel.hpp", "dsh::k_sdirk_resident<dsh::JitModel, true, false, " + std::to_string(s) + ">"});
    for (const auto& u : units) {
      for (int variant = 0; variant < (family == 2 ? 2 : 1); ++variant) {  // the BDF has a second code object for small ensembles (compile_module: "#small")
        const std::string gkey = u.second + (variant ? "#small" : "");
        const std::string key = std::string(u.first) + "|" + gkey;
        if (rec->modules.count(key) && rec->modules[key]) continue;
        auto m = std::make_unique<JitModule>();
        int rc = compile_module(*rec, u.first, {u.second}, m.get(), gkey);
        if (rc != DSH_OK) return rc;
        rec->modules[key] = std::move(m);
      }
    }
    return DSH_OK;
  }
  auto tf = [](bool b) { return b ? "true" : "false"; };
  if (family == 0) units.push_back({ops_header(rec->info.form), "ops", st ? jit_static_op_names() : std::vector<std::string>()});
  else if (!st && family == 2 && rec->info.n <= (rec->info.has_mass ? 48 : 64) && rec->info.nroots <= 2) {  // wavefront-per-member BDF
    const int64_t n = rec->info.n;
    const std::string name = std::string("dsh::k_bdf_wave_member<") + (n <= 16 ? "16" : n <= 32 ? "32" : n <= 48 ? "48" : "64") + ">";
    units.push_back({"dsh_jit_wave_member.hpp", name, {name}});
    if (rec->info.has_sens && !rec->info.has_mass && rec->info.nroots == 0 && rec->info.np <= 16) {  // with forward sensitivities (dsh_bdf_solve_wave_member_sens)
      const std::string sname = name.substr(0, name.size() - 1) + ", true>";
      units.push_back({"dsh_jit_wave_member.hpp", sname, {sname}});
    }
    if (!rec->info.has_mass && n > 32) {  // what dsh_bdf_solve_wave_member launches for an identity-mass model of this size (dsh_wave_member.hip: small_team): the workgroup form, LU in registers
      const std::string rname = "dsh::k_bdf_team_member_rl<" + std::to_string((n + 7) / 8 * 8) + ">";
      units.push_back({"dsh_jit_team_member.hpp", rname, {rname}});
    }
  }
  else if (!st && family == 3 && rec->info.n <= (rec->info.has_mass ? 48 : 64) && rec->info.nroots <= 2) {  // wavefront-per-member TR-BDF2 / ESDIRK34
    const int64_t n = rec->info.n;
    for (int S = 3; S <= 4; ++S) {
      const std::string name = std::string("dsh::k_sdirk_wave_member<") + (n <= 16 ? "16" : n <= 32 ? "32" : n <= 48 ? "48" : "64") + ", " + std::to_string(S) + ">";
      units.push_back({"dsh_jit_sdirk_wave_member.hpp", name, {name}});
      if (rec->info.has_sens && !rec->info.has_mass && rec->info.nroots == 0 && rec->info.np <= 16) {  // with forward sensitivities (dsh_sdirk_solve_wave_member_sens)
        const std::string sname = name.substr(0, name.size() - 1) + ", true>";
        units.push_back({"dsh_jit_sdirk_wave_member.hpp", sname, {sname}});
      }
    }
  }
  else if (!st) { set_error("dsh_model_precompile: run-time-sized models have the operator kernels and (n <= 64) the wavefront-per-member integrators"); return DSH_E_UNSUPPORTED; }
  else if (family == 1) {
    units.push_back({"dsh_fused_kernels.hpp", "jac_factor", {"dsh::k_jac_factor<dsh::JitModel>"}});
    for (int sd = 0; sd < 2; ++sd)
      for (int ba = 0; ba < 2; ++ba)
        for (int we = 0; we < 2; ++we) {
          if (sd && we) continue;
          Unit u{"dsh_fused_kernels.hpp", std::string("newton") + tf(sd) + tf(ba) + tf(we), {}};
          for (int nit = 1; nit <= 4; ++nit)
            u.group.push_back(std::string("dsh::k_newton_iter<dsh::JitModel, ") + tf(sd) + ", " + tf(ba) + ", " + tf(we) + ", " + std::to_string(nit) + ">");
          units.push_back(u);
        }
    for (int ba = 0; ba < 2; ++ba) {
      Unit u{"dsh_fused_kernels.hpp", std::string("accept") + tf(ba), {}};
      for (int nit = 1; nit <= 4; ++nit) u.group.push_back(std::string("dsh::k_accept_newton<dsh::JitModel, ") + tf(ba) + ", " + std::to_string(nit) + ">");
      units.push_back(u);
    }
  } else if (family == 2 || family == 3) {
    if (rec->info.n > 4) { set_error("dsh_model_precompile: the device-resident integrators need n <= 4"); return DSH_E_UNSUPPORTED; }
    for (int ba = 0; ba < 2; ++ba)
      for (int wave = 0; wave < 2; ++wave) {
        if (family == 2) {
          const std::string name = std::string("dsh::k_bdf_adaptive<dsh::JitModel, ") + tf(ba) + ", " + tf(wave) + ">";
          units.push_back({"dsh_adaptive_kernel.hpp", name, {name}});
        } else {
          for (int s = 3; s <= 4; ++s) {
            const std::string name = std::string("dsh::k_sdirk_resident<dsh::JitModel, ") + tf(ba) + ", " + tf(wave) + ", " + std::to_string(s) + ">";
            units.push_back({"dsh_sdirk_kernel.hpp", name, {name}});
          }
        }
      }
  } else { set_error("dsh_model_precompile: unknown kernel family"); return DSH_E_INVALID; }
  for (const Unit& u : units) {
    const std::string key = std::string(u.header) + "|" + u.key;
    if (rec->modules.count(key) && rec->modules[key]) continue;
    auto m = std::make_unique<JitModule>();
    int rc = compile_module(*rec, u.header, u.group, m.get());
    if (rc != DSH_OK) return rc;
    rec->modules[key] = std::move(m);
  }
  return DSH_OK;
}

}  // extern "C"
