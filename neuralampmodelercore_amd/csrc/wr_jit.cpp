// wr_jit.cpp — per-model compile of nam_wn_reg_kernel (wr_jit.h).
//
// The reference compiles its fast WaveNet path for the channel counts it knows (template <int C> A2FastModel,
// NAM/wavenet/a2_fast.cpp:57) and every other model takes the generic path. Here a layer is a fully unrolled function
// of eleven compile-time parameters; the shipped example models' shapes are instantiated ahead of time, and any other
// model gets the same source compiled for exactly its own shapes the first time it is loaded (seconds; cached by a hash
// of the shape tables, the kernel sources and the compiler, so the second load of any model with the same shapes — in
// any process — is a file lookup).
#include "wr_jit.h"

#include <atomic>
#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <utility>
#include <vector>

extern char** environ;

namespace namhip
{
namespace
{
std::string lib_dir()
{
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&wr_jit_enabled), &info) && info.dli_fname)
  {
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    return k == std::string::npos ? "." : p.substr(0, k);
  }
  return ".";
}
bool readable(const std::string& p)
{
  return access(p.c_str(), R_OK) == 0;
}
std::string read_file(const std::string& p)
{
  std::ifstream f(p, std::ios::binary);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
// FNV-1a, 64 bit: a cache key, not a security boundary
unsigned long long fnv(const std::string& s, unsigned long long h = 1469598103934665603ull)
{
  for (unsigned char c : s)
  {
    h ^= c;
    h *= 1099511628211ull;
  }
  return h;
}
// A cache directory holds code objects this process will load and run on the GPU: it must be a real directory (not a
// symbolic link) that nobody else can write into — ours (or, for the install directory, root's) and neither group- nor
// world-writable. `create`: make it (mode 0700) when it does not exist.
bool cache_dir_ok(const std::string& d, bool create, bool must_own)
{
  struct stat st;
  if (lstat(d.c_str(), &st) != 0)
  {
    if (!create || mkdir(d.c_str(), 0700) != 0 || lstat(d.c_str(), &st) != 0)
      return false;
  }
  if (!S_ISDIR(st.st_mode) || (st.st_mode & (S_IWGRP | S_IWOTH)) != 0)
    return false;
  if (st.st_uid != geteuid() && (must_own || st.st_uid != 0))
    return false;
  return access(d.c_str(), W_OK | X_OK) == 0;
}
// hipcc with an argument vector (no shell: the paths come from the environment and from dladdr), output into `log`
int run_compiler(const std::vector<std::string>& argv, const std::string& log)
{
  std::vector<char*> av;
  for (const std::string& a : argv)
    av.push_back(const_cast<char*>(a.c_str()));
  av.push_back(nullptr);
  posix_spawn_file_actions_t fa;
  if (posix_spawn_file_actions_init(&fa) != 0)
    return -1;
  posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
  posix_spawn_file_actions_adddup2(&fa, 1, 2);
  pid_t pid = 0;
  const int rc = posix_spawn(&pid, av[0], &fa, nullptr, av.data(), environ);
  posix_spawn_file_actions_destroy(&fa);
  if (rc != 0)
    return -1;
  int status = 0;
  while (waitpid(pid, &status, 0) < 0)
    if (errno != EINTR)
      return -1;
  return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}
const char* const kSources[] = {"kernel_wn_reg.hip", "device_common.h", "kernels.h", "plan.h", "model_spec.h", "persist_wave.h"};
} // namespace

bool wr_jit_enabled()
{
  const char* e = std::getenv("NAM_HIP_JIT");
  return !(e && e[0] == '0');
}

std::string wr_jit_build(const WrShapeSet& shapes, std::string& why)
{
  if (!wr_jit_enabled())
  {
    why = "the per-model compile is switched off (NAM_HIP_JIT=0)";
    return "";
  }
  const std::string lib = lib_dir();
  const char* e_src = std::getenv("NAM_HIP_JIT_SRC");
  const std::string src = e_src ? e_src : lib + "/../csrc";
  const char* e_cc = std::getenv("NAM_HIP_HIPCC");
  const std::string hipcc = e_cc ? e_cc : "/opt/rocm/bin/hipcc";
  for (const char* f : kSources)
    if (!readable(src + "/" + f))
    {
      why = "per-model compile: kernel source " + src + "/" + f + " not found (NAM_HIP_JIT_SRC)";
      return "";
    }
  if (access(hipcc.c_str(), X_OK) != 0)
  {
    why = "per-model compile: no compiler at " + hipcc + " (NAM_HIP_HIPCC)";
    return "";
  }
  const std::string text = shapes.header_text();
  unsigned long long h = fnv(text);
  for (const char* f : kSources)
    h = fnv(read_file(src + "/" + f), h);
  h = fnv(hipcc, h);
  // (developer switch: extra compiler arguments, e.g. -DNAM_WR_... knobs of kernel_wn_reg.hip, for A/B runs; part of the key)
  std::vector<std::string> extra;
  if (const char* e = std::getenv("NAM_HIP_JIT_FLAGS"))
  {
    std::istringstream is(e);
    for (std::string t; is >> t;)
      extra.push_back(t);
    h = fnv(std::string(e), h);
  }
  char key[32];
  std::snprintf(key, sizeof(key), "%016llx", h);

  // the cache: NAM_HIP_JIT_CACHE, else jit/ next to the library (the install), else the user's own cache directory
  // ($XDG_CACHE_HOME or ~/.cache), else a 0700 directory of this user under /tmp — never a directory someone else can write
  std::string cache;
  {
    std::vector<std::pair<std::string, bool>> cand; // {path, must be owned by this user}
    if (const char* e = std::getenv("NAM_HIP_JIT_CACHE"))
      cand.push_back({e, false});
    else
    {
      cand.push_back({lib + "/jit", false});
      const char* xdg = std::getenv("XDG_CACHE_HOME");
      const char* home = std::getenv("HOME");
      if (xdg && xdg[0] == '/')
        cand.push_back({std::string(xdg) + "/nam_hip_jit", true});
      else if (home && home[0] == '/')
      {
        (void)mkdir((std::string(home) + "/.cache").c_str(), 0700);
        cand.push_back({std::string(home) + "/.cache/nam_hip_jit", true});
      }
      cand.push_back({"/tmp/nam_hip_jit_" + std::to_string((long)geteuid()), true});
    }
    for (const auto& c : cand)
      if (cache_dir_ok(c.first, true, c.second))
      {
        cache = c.first;
        break;
      }
    if (cache.empty())
    {
      why = "per-model compile: no cache directory that only this user can write (NAM_HIP_JIT_CACHE)";
      return "";
    }
  }
  const std::string out = cache + "/wn_reg_" + key + ".hsaco";
  if (readable(out))
  {
    (void)utimensat(AT_FDCWD, out.c_str(), nullptr, 0); // last use (tools/warm_jit_cache.py prunes what nothing asks for any more)
    return out;
  }

  // compile into a private name, publish with a rename (two processes — or two host threads of one, e.g. a batch spread
  // over devices — loading the same new model race benignly)
  static std::atomic<unsigned> serial{0};
  const std::string tag = std::to_string((long)getpid()) + "_" + std::to_string(serial.fetch_add(1));
  const std::string hdr = cache + "/wn_reg_" + key + "." + tag + ".shapes.h", tmp = cache + "/wn_reg_" + key + "." + tag + ".tmp",
                    log = cache + "/wn_reg_" + key + "." + tag + ".log";
  {
    std::ofstream f(hdr);
    f << "// generated by libnam_hip.so (wr_jit.cpp): the layer shapes of one model\n" << text;
  }
  std::vector<std::string> cmd = {hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-mllvm", "-amdgpu-mfma-vgpr-form",
                                  "-include", hdr, "-I" + src, "-c", "-o", tmp, src + "/kernel_wn_reg.hip"};
  cmd.insert(cmd.end(), extra.begin(), extra.end());
  int rc = run_compiler(cmd, log);
  if (rc != 0 || !readable(tmp))
  {
    // once more with the compiler's own choice of where matrix-instruction results live: forcing them into vector registers
    // (-amdgpu-mfma-vgpr-form: fewer moves) crashes ROCm 7.2's "Rewrite AGPR-Copy-MFMA" pass on some register-hungry layer
    // shapes (a gated 16-row layer with every FiLM on a 4-value condition: tools/fuzz_models.py 160 7707, model 123)
    std::remove(tmp.c_str());
    cmd = {hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-include", hdr, "-I" + src, "-c", "-o", tmp,
           src + "/kernel_wn_reg.hip"};
    cmd.insert(cmd.end(), extra.begin(), extra.end());
    rc = run_compiler(cmd, log);
  }
  if (!std::getenv("NAM_HIP_JIT_KEEP")) // (developer switch: the generated shapes header stays next to the code object)
    std::remove(hdr.c_str());
  if (rc != 0 || !readable(tmp))
  {
    std::string tail = read_file(log);
    if (tail.size() > 600)
      tail = tail.substr(tail.size() - 600);
    std::remove(tmp.c_str());
    std::remove(log.c_str());
    why = "per-model compile failed: " + tail;
    return "";
  }
  std::remove(log.c_str());
  if (std::rename(tmp.c_str(), out.c_str()) != 0)
  {
    std::remove(tmp.c_str());
    if (!readable(out))
    {
      why = "per-model compile: could not publish " + out;
      return "";
    }
  }
  return out;
}

} // namespace namhip
