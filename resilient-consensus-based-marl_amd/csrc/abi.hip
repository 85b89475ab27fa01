// ABI version of the rcmarl C interface (include/rcmarl.h), and the two pieces of process-wide host state the library has:
// the operand form of the lattice path and the record of which form each packed buffer was last written in.
#include "rcmarl_lattice.h"
#include <stdlib.h>
#include <mutex>

RCMARL_EXPORT int rcmarl_abi_version(void) { return 4; }

// sizeof(rcmarl_mb_job) (what = 0) or the offset of its field number `what` (1 x_seed_stride, 2 theta, 3 agents, 4 n_adv, 5 in_dim, 6 ldp,
// 7 y, 8 perm, 9 loss_out, 10 ovf_flags), -1 otherwise: lets a binding check its own declaration of the structure against the library's
RCMARL_EXPORT int rcmarl_mb_job_layout(int what) {
  switch (what) {
    case 0: return (int)sizeof(rcmarl_mb_job);
    case 1: return (int)offsetof(rcmarl_mb_job, x_seed_stride);
    case 2: return (int)offsetof(rcmarl_mb_job, theta);
    case 3: return (int)offsetof(rcmarl_mb_job, agents);
    case 4: return (int)offsetof(rcmarl_mb_job, n_adv);
    case 5: return (int)offsetof(rcmarl_mb_job, in_dim);
    case 6: return (int)offsetof(rcmarl_mb_job, ldp);
    case 7: return (int)offsetof(rcmarl_mb_job, y);
    case 8: return (int)offsetof(rcmarl_mb_job, perm);
    case 9: return (int)offsetof(rcmarl_mb_job, loss_out);
    case 10: return (int)offsetof(rcmarl_mb_job, ovf_flags);
    default: return -1;
  }
}

namespace {
std::mutex g_mu;
int g_mode = -1;                                         // -1: not read yet
int mode_from_env() {
  const char* e = getenv("RCMARL_LAT_F16");
  return e ? (atoi(e) & 3) : RC_LAT_F16_DEFAULT;
}
// which operand form (0 = bf16 pieces, 1 = f16 pieces) a packed buffer was last written in, by base pointer.  A small table:
// producers overwrite their entry, consumers that find one check it; unknown pointers (a window into a larger buffer) pass.
constexpr int TAGS = 512;
struct Tag { const void* p; int form; };
Tag g_tags[TAGS];
int g_ntags = 0;
}  // namespace

// the operand form: read from RCMARL_LAT_F16 ONCE (at the first call), afterwards changed only through
// rcmarl_lattice_set_f16_mode (callers that switch forms inside one process: tests, bench.py's exact-form workload)
int rc_lat_f16_mode() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_mode < 0) g_mode = mode_from_env();
  return g_mode;
}
RCMARL_EXPORT int rcmarl_lattice_f16_mode(void) { return rc_lat_f16_mode(); }
RCMARL_EXPORT int rcmarl_lattice_set_f16_mode(int mode) {
  if (mode > 3) return RCMARL_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  g_mode = mode < 0 ? mode_from_env() : mode;
  return RCMARL_OK;
}

void rc_form_set(const void* p, int form) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < g_ntags; ++i)
    if (g_tags[i].p == p) { g_tags[i].form = form; return; }
  if (g_ntags < TAGS) { g_tags[g_ntags].p = p; g_tags[g_ntags].form = form; ++g_ntags; return; }
  g_tags[(reinterpret_cast<uintptr_t>(p) >> 8) % TAGS] = Tag{p, form};      // full: evict
}
// a caller that frees or re-purposes a packed buffer says so: a recycled address must not inherit the old buffer's form
RCMARL_EXPORT int rcmarl_lattice_forget(const void* p) {
  if (!p) return RCMARL_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < g_ntags; ++i)
    if (g_tags[i].p == p) { g_tags[i] = g_tags[--g_ntags]; break; }
  return RCMARL_OK;
}
bool rc_form_ok(const void* p, int form) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < g_ntags; ++i)
    if (g_tags[i].p == p) return g_tags[i].form == form;
  return true;
}
