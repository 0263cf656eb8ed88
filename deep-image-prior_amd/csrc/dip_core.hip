// Error slot, ABI version and device identity of libdip_hip.so.
#include "dip_common.h"
#include <string.h>

static char g_err[256] = "";

extern "C" void dip_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* dip_last_error(void) { return g_err; }
extern "C" int dip_abi_version(void) { return DIP_ABI_VERSION; }

// PCI address ("0000:d9:00.0") of HIP device `device` as THIS library's HIP runtime enumerates it: what a host-side
// monitor needs to find the GPU's sysfs directory (/sys/bus/pci/devices/<address>: hwmon power / clocks, numa_node) --
// the position among /sys/class/drm/card* says nothing in a container that sees every card of the host.
extern "C" int dip_device_pci_bus_id(int device, char* buf, int len) {
    if (buf == nullptr || len < 13) DIP_FAIL("device_pci_bus_id: buffer of >= 13 bytes required");
    hipError_t e = hipDeviceGetPCIBusId(buf, len, device);
    if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
    return 0;
}
