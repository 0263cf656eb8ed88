// Error slot + ABI version of libdip_hip.so.
#include "dip_common.h"
#include <string.h>

static char g_err[256] = "";

extern "C" void dip_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* dip_last_error(void) { return g_err; }
extern "C" int dip_abi_version(void) { return DIP_ABI_VERSION; }
