// Library-wide state: last-error string (thread local) and cached device properties.
#include "common.h"
#include <mutex>
#include <stdlib.h>

static thread_local char g_err[512] = "";
static int g_num_cus = 0;
static std::once_flag g_once;

void sk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static void query_device() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    g_num_cus = prop.multiProcessorCount;
  if (g_num_cus <= 0) g_num_cus = 256;  // MI355X
}

int sk_num_cus() {
  std::call_once(g_once, query_device);
  return g_num_cus;
}

// CUs the GEMM schedulers plan their rounds for: all of them, minus those the caller has said are held by other
// work while the GEMMs run (skhip_set_reserved_cus; SKHIP_RESERVE_CUS as the start value).
static int g_reserved_cus = -1;
int sk_sched_cus() {
  if (g_reserved_cus < 0) {
    const char* e = getenv("SKHIP_RESERVE_CUS");
    const int v = e ? atoi(e) : 0;
    g_reserved_cus = v > 0 ? v : 0;
  }
  const int n = sk_num_cus() - g_reserved_cus;
  return n >= 8 ? n : 8;
}

extern "C" int skhip_set_reserved_cus(int n) {
  if (n < 0 || n > sk_num_cus() - 8) {
    sk_set_error("skhip_set_reserved_cus: %d not in 0..%d", n, sk_num_cus() - 8);
    return SK_ERR_ARG;
  }
  g_reserved_cus = n;
  return SK_OK;
}

extern "C" int skhip_init(void) {
  sk_num_cus();
  float* scratch = nullptr;
  return sk_colsum_scratch(&scratch);  // the current device's copy of skhip_colsum_noalloc's scratch array
}
extern "C" const char* skhip_version(void) { return "specialk_hip 0.1 (gfx950)"; }
extern "C" const char* skhip_last_error(void) { return g_err; }
