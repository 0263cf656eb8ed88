// Error slot, ABI version and device identity of libdip_hip.so.
#include "dip_common.h"
#include "dip_group.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[256] = "";      // per thread, like the group context below

extern "C" void dip_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* dip_last_error(void) { return g_err; }
extern "C" int dip_abi_version(void) { return DIP_ABI_VERSION; }

// sha256 (first 16 hex digits) over the library's sources -- every csrc/*.hip, csrc/*.h and include/dip_hip.h, names and
// contents in sorted order -- handed in by the one build recipe (__graft_entry__.py: -DDIP_BUILD_ID=...).  build() rebuilds
// when the id in the binary differs from the id of the sources on disk (not on mtimes), and bench.py prints it in its JSON
// line: a record of a run names the sources that produced the binary that ran.
#ifndef DIP_BUILD_ID
#define DIP_BUILD_ID "unknown"
#endif
static const char g_build_id[] = "DIP_BUILD_ID=" DIP_BUILD_ID;
extern "C" const char* dip_build_id(void) { return g_build_id + 13; }

// PCI address ("0000:d9:00.0") of HIP device `device` as THIS library's HIP runtime enumerates it: what a host-side
// monitor needs to find the GPU's sysfs directory (/sys/bus/pci/devices/<address>: hwmon power / clocks, numa_node) --
// the position among /sys/class/drm/card* says nothing in a container that sees every card of the host.
extern "C" int dip_device_pci_bus_id(int device, char* buf, int len) {
    if (buf == nullptr || len < 13) DIP_FAIL("device_pci_bus_id: buffer of >= 13 bytes required");
    hipError_t e = hipDeviceGetPCIBusId(buf, len, device);
    if (e != hipSuccess) {
        dip_set_error(hipGetErrorString(e));
        (void)hipGetLastError();        // a query must not leave HIP's sticky last-error set: the next torch call would raise it
        return (int)e;                   // (round 5: rank r asking for GPU r + 1 on a 1-GPU box made `net.to(dev)` fail later)
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Grouped multi-instance execution (dip_group.h; include/dip_hip.h "grouped execution").  Host-side state only: which
// launches are grouped is decided where they are issued, so a grouped launch list is hipGraph-capturable like a solo one.
// One process per GPU; the context is THREAD-LOCAL: a group opened by one host thread replicates that thread's launches only
// (round 4 had a process-wide global: a launch from another thread between dip_group_begin and dip_group_end -- a solo net's
// forward, a monitor -- was replicated B times or refused, and a fault raised there was consumed by whoever checked next).
// ---------------------------------------------------------------------------------------------
static unsigned native_default() {
    const char* e = getenv("DIP_GROUP_NATIVE");
    return e ? (unsigned)strtoul(e, nullptr, 0) : ~0u;
}
static thread_local DipGroupCtx g_grp = {1, 0, nullptr, 0, ~0u};
static thread_local bool g_grp_native_init = false;
static thread_local bool g_grp_fault = false;

extern "C" const DipGroupCtx* dip_group_ctx(void) { return &g_grp; }
extern "C" void dip_group_fault(const char* what) {
    char buf[240];
    snprintf(buf, sizeof(buf), "grouped launch refused: %s", what ? what : "?");
    dip_set_error(buf);
    g_grp_fault = true;
}
extern "C" int dip_group_take_fault(void) {
    const bool f = g_grp_fault;
    g_grp_fault = false;
    return f ? 1 : 0;
}

extern "C" int dip_group_begin(int ninst, long long stride_bytes, const void* base, long long row_bytes) {
    if (g_grp.ninst != 1) DIP_FAIL("group_begin: a group is already open");
    if (ninst < 1 || ninst > 4096) DIP_FAIL("group_begin: 1..4096 instances");
    if (base == nullptr || row_bytes <= 0) DIP_FAIL("group_begin: slab of instance 0 required");
    if (ninst > 1 && (stride_bytes < row_bytes || (stride_bytes & 255) != 0 || (reinterpret_cast<uintptr_t>(base) & 255) != 0))
        DIP_FAIL("group_begin: slabs must not overlap and must be 256-byte aligned");
    if (!g_grp_native_init) { g_grp.native = native_default(); g_grp_native_init = true; }
    g_grp.ninst = ninst;
    g_grp.stride = stride_bytes;
    g_grp.base = static_cast<const char*>(base);
    g_grp.row = row_bytes;
    g_grp_fault = false;
    return 0;
}
extern "C" int dip_group_end(void) {
    g_grp.ninst = 1;
    g_grp.stride = 0;
    g_grp.base = nullptr;
    g_grp.row = 0;
    return 0;
}
extern "C" int dip_group_size(void) { return g_grp.ninst; }
// mask >= 0: which kernel families run their native grouped kernels (DIP_FAM_* bits; the others fall back to a host loop
// of B solo launches); mask < 0: query.  Returns the mask in force before the call.  Default: DIP_GROUP_NATIVE or all.
extern "C" int dip_group_native(int mask) {
    if (!g_grp_native_init) { g_grp.native = native_default(); g_grp_native_init = true; }
    const unsigned prev = g_grp.native;
    if (mask >= 0) g_grp.native = (unsigned)mask;
    return (int)(prev & 0x7fffffffu);
}
