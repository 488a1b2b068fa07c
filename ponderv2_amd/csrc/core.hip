// Library-wide state of libponderv2_hip.so: ABI version and the per-thread error string.
#include <string.h>

#include "common.h"

namespace pv2 {
static thread_local char g_error[256] = "";
void set_error(const char* msg) {
  strncpy(g_error, msg ? msg : "", sizeof(g_error) - 1);
  g_error[sizeof(g_error) - 1] = 0;
}
}  // namespace pv2

extern "C" {
int pv2_abi_version(void) { return 1; }
const char* pv2_last_error(void) { return pv2::g_error; }
}
