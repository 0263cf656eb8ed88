// Command lists: the static launch list of one direction of the skip-net (forward / backward: 80..170 launches on up to four
// HIP streams, with the event records / waits that order them) issued by ONE call into the library instead of one ctypes
// call per launch.  The reference has no counterpart (its launches are issued by PyTorch's dispatcher, one Python call per
// op: utils/common_utils.py:223-230 drives nn.Module.forward / autograd); this is the host half of "an iteration is a static
// launch list" (DESIGN.md section 3.3): the Python thread of a rank spent ~10 us per launch (2.4 ms of a 5.5 ms iteration at
// 512 x 512, 2.0 of 2.7 ms for the 'library' net), the loop below spends what hipLaunchKernel costs.
//
// A command is LAUNCH (an entry point of this library, its arguments as 8-byte slots, a stream index), RECORD (event, stream)
// or WAIT (stream, event).  The caller owns the slot arrays, the streams and the events (dip_events_create); nothing is
// allocated or synchronised here, so a list is hipGraph-capturable exactly like the launches it contains.
#include "dip_common.h"
#include <string.h>
#include <utility>

namespace {

template <class T>
inline T slot_as(const uint64_t* s) {
    T v;
    memcpy(&v, s, sizeof(T));          // little-endian: an int / float lives in the low bytes of its slot
    return v;
}
template <class... A, size_t... I>
inline int call_slots(int (*fn)(A...), const uint64_t* s, std::index_sequence<I...>) {
    return fn(slot_as<A>(s + I)...);
}
template <auto Fn> struct Thunk;
template <class... A, int (*Fn)(A...)>
struct Thunk<Fn> {
    static constexpr int nargs = (int)sizeof...(A);
    static int call(const uint64_t* s) { return call_slots(Fn, s, std::index_sequence_for<A...>{}); }
};

struct Entry {
    const char* name;
    int (*call)(const uint64_t*);
    int nargs;                          // including the trailing stream argument
};
#define DIP_REG(f) {#f, &Thunk<&f>::call, Thunk<&f>::nargs}
// every entry point of include/dip_hip.h that launches on a stream (last parameter: void* stream)
const Entry g_fns[] = {
    DIP_REG(dip_nchw_to_nhwc), DIP_REG(dip_nhwc_to_nchw), DIP_REG(dip_head_fwd), DIP_REG(dip_head_bwd),
    DIP_REG(dip_pack_weights), DIP_REG(dip_pack_weights_bf3),
    DIP_REG(dip_conv_igemm), DIP_REG(dip_conv_thin4), DIP_REG(dip_conv_igemm_dma_cols), DIP_REG(dip_conv_small), DIP_REG(dip_conv_thin),
    DIP_REG(dip_conv_dgrad_ring), DIP_REG(dip_conv_bf3_cols), DIP_REG(dip_conv_splitk_finish),
    DIP_REG(dip_conv_wgrad), DIP_REG(dip_wgrad_bf3), DIP_REG(dip_wgrad_thin), DIP_REG(dip_conv_wgrad_tail), DIP_REG(dip_wgrad_tail_stream), DIP_REG(dip_wgrad_reduce),
    DIP_REG(dip_bn_finalize), DIP_REG(dip_bn_bwd_stats), DIP_REG(dip_bn_bwd_stats_fin), DIP_REG(dip_bn_bwd_finalize),
    DIP_REG(dip_bn_bwd_finalize2), DIP_REG(dip_bn_bwd_apply), DIP_REG(dip_bn_bwd_apply_src), DIP_REG(dip_bn_bwd_apply_fin),
    DIP_REG(dip_bn_bwd_apply_src_fin), DIP_REG(dip_bn_bwd_one), DIP_REG(dip_fold_to_nchw), DIP_REG(dip_fold_to_nhwc),
    DIP_REG(dip_upcat_fwd), DIP_REG(dip_upcat_fwd_fin), DIP_REG(dip_avgpool2_fwd), DIP_REG(dip_avgpool2_bwd),
    DIP_REG(dip_maxpool2_fwd), DIP_REG(dip_maxpool2_bwd), DIP_REG(dip_upsample_bwd_stats), DIP_REG(dip_upsample_bwd_stats_crop),
    DIP_REG(dip_upsample_bwd_stats_crop_fin), DIP_REG(dip_upsample_bwd_one),
    DIP_REG(dip_adam_step), DIP_REG(dip_noise_axpy), DIP_REG(dip_adam_tick), DIP_REG(dip_adam_step_dev),
    DIP_REG(dip_noise_axpy_dev), DIP_REG(dip_noise_axpy_dev2), DIP_REG(dip_counter_add),
    DIP_REG(dip_loss_head_fwd), DIP_REG(dip_loss_head_bwd), DIP_REG(dip_fit_monitor), DIP_REG(dip_arena_backtrack),
    DIP_REG(dip_lanczos_down_fwd), DIP_REG(dip_lanczos_down_bwd), DIP_REG(dip_down_dense_fwd), DIP_REG(dip_down_dense_bwd_data),
    DIP_REG(dip_down_dense_bwd_weight),
};
constexpr int NFN = (int)(sizeof(g_fns) / sizeof(g_fns[0]));

}  // namespace

extern "C" int dip_list_fn_id(const char* name) {
    if (name == nullptr) return -1;
    for (int i = 0; i < NFN; ++i)
        if (strcmp(g_fns[i].name, name) == 0) return i;
    return -1;
}

extern "C" int dip_list_fn_nargs(int fn) { return (fn >= 0 && fn < NFN) ? g_fns[fn].nargs : -1; }

extern "C" int dip_list_run(const DipCmd* cmds, int n, void* const* streams, int nstreams, void* const* events, int nevents,
                            int* failed_at) {
    if (failed_at != nullptr) *failed_at = -1;
    for (int i = 0; i < n; ++i) {
        const DipCmd& c = cmds[i];
        int rc = 0;
        if (c.stream < 0 || c.stream >= nstreams) rc = -1;
        else if (c.kind == DIP_CMD_LAUNCH) {
            if (c.fn < 0 || c.fn >= NFN || c.slots == nullptr || c.nslots != g_fns[c.fn].nargs) rc = -1;
            else {
                c.slots[c.nslots - 1] = (uint64_t)reinterpret_cast<uintptr_t>(streams[c.stream]);
                rc = g_fns[c.fn].call(c.slots);
            }
        } else if (c.event < 0 || c.event >= nevents) rc = -1;
        else if (c.kind == DIP_CMD_RECORD)
            rc = (int)hipEventRecord(reinterpret_cast<hipEvent_t>(events[c.event]), reinterpret_cast<hipStream_t>(streams[c.stream]));
        else if (c.kind == DIP_CMD_WAIT)
            rc = (int)hipStreamWaitEvent(reinterpret_cast<hipStream_t>(streams[c.stream]), reinterpret_cast<hipEvent_t>(events[c.event]), 0);
        else rc = -1;
        if (rc != 0) {
            if (failed_at != nullptr) *failed_at = i;
            if (rc == -1 && c.kind != DIP_CMD_LAUNCH) dip_set_error("list_run: malformed command (stream / event index, kind)");
            else if (rc == -1 && (c.fn < 0 || c.fn >= NFN || c.slots == nullptr || c.nslots != g_fns[c.fn < 0 || c.fn >= NFN ? 0 : c.fn].nargs))
                dip_set_error("list_run: malformed LAUNCH command (function id, slot count)");
            else if (c.kind != DIP_CMD_LAUNCH) dip_set_error(hipGetErrorString((hipError_t)rc));
            return rc;
        }
    }
    return 0;
}

extern "C" int dip_events_create(void** events, int n) {
    for (int i = 0; i < n; ++i) {
        hipEvent_t e;
        hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (rc != hipSuccess) {
            dip_set_error(hipGetErrorString(rc));
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(events[j]));
            return (int)rc;
        }
        events[i] = reinterpret_cast<void*>(e);
    }
    return 0;
}

extern "C" int dip_events_destroy(void** events, int n) {
    int rc = 0;
    for (int i = 0; i < n; ++i)
        if (events[i] != nullptr) {
            hipError_t e = hipEventDestroy(reinterpret_cast<hipEvent_t>(events[i]));
            if (e != hipSuccess) rc = (int)e;
            events[i] = nullptr;
        }
    return rc;
}
