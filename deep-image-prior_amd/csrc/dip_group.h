// Grouped multi-instance execution: B independent skip-nets -- own weights, own BatchNorm
// statistics, own Adam state; B copies of models/skip.py:45-100 of the reference -- advance through ONE launch list.
//
// Memory model.  Every buffer of a fit (parameters, gradients, Adam moments, BatchNorm state, packed weights, activations,
// scratch, tables, input, target, loss) is carved from ONE slab; the B slabs are the rows of a [B][stride] allocation, laid
// out identically.  The caller compiles the launch list for instance 0 and brackets it with dip_group_begin / dip_group_end;
// while a group is open every launch of the library serves all B instances: instance b sees every non-NULL pointer argument
// and descriptor field advanced by b * stride bytes.  Nothing else changes: same plans, same tile walk, same summation order
// per instance, so an instance's results are bit-identical to those of the same fit run on its own.
//
// Two forms per kernel, chosen at the launch site:
//   * native (dip_launch_pair): the kernel carries a `bool GRP` template parameter.  GRP = false is the solo kernel,
//     instruction for instruction what it was before the parameter existed (tools/isa_diff.py; the trailing group argument
//     is an empty struct and the descriptor is used in place); GRP = true is launched with gridDim.z x B workgroups,
//     instance = blockIdx.z / gz, and shifts a copy of its descriptor / its pointers in SGPRs before the same body runs --
//     ONE dispatch for B nets;
//   * host loop (dip_launch, or a family switched off in the native mask): B dispatches of the solo kernel with host-shifted
//     arguments.  Correct for every kernel without touching it; used for the kernels that fill the chip on their own
//     (bf16-pipe conv / weight gradient, the persistent 1x1 kernel) and the rarely used ones.  DIP_GROUP_NATIVE=<mask>
//     (dip_group_native) turns native families into host loops for A/B runs and for bisecting a parity failure.
// Which pointers a descriptor holds is written down ONCE per struct (dip_ptrs); the shift, and the host-side check that
// every pointer of a grouped launch lies inside instance 0's slab, are visitors over it.
#pragma once
#include "dip_common.h"
#include <stdint.h>
#include <type_traits>

struct DipGroupCtx {
    int ninst;            // 1: no group open
    long long stride;     // bytes between the slabs of consecutive instances
    const char* base;     // slab of instance 0 ...
    long long row;        // ... and its size in bytes
    unsigned native;      // DIP_FAM_* families whose native grouped kernels are enabled
};
extern "C" const DipGroupCtx* dip_group_ctx(void);
extern "C" void dip_group_fault(const char* what);      // sticky; DIP_CHECK_LAUNCH reports it

enum : unsigned {
    DIP_FAM_BN = 1u,          // BatchNorm finalisation / backward phases
    DIP_FAM_CONV = 2u,        // register-staged implicit GEMM + split-K finish
    DIP_FAM_DMA = 4u,         // LDS-DMA implicit GEMM (all modes)
    DIP_FAM_SMALL = 8u,       // conv_small (+ ring data gradients)
    DIP_FAM_THIN = 16u,       // conv_thin4, thin weight gradients
    DIP_FAM_WGRAD = 32u,      // MFMA weight gradient + slab reduction
    DIP_FAM_LOSS = 64u,       // loss head, Adam, reg-noise, iteration state
    DIP_FAM_MISC = 128u,      // layout conversion, weight packing
    DIP_FAM_UPCAT = 256u,     // up-sample + concat and its adjoint
};

// what a grouped kernel learns about the group: the slab stride and the gridDim.z of the solo launch
struct DipGrp {
    long long stride;
    int gz;
};
struct DipNoGrp {};
template <bool GRP> using DipGrpArg = std::conditional_t<GRP, DipGrp, DipNoGrp>;

// ---------------------------------------------------------------------------------------------
// pointer fields of the descriptor structs (include/dip_hip.h).  A struct without an overload does not compile as a
// launch argument: nothing with a pointer inside can slip through unshifted.
// ---------------------------------------------------------------------------------------------
template <class F> __host__ __device__ inline void dip_ptrs(DipNoGrp&, F&) {}
template <class F> __host__ __device__ inline void dip_ptrs(DipTransform& t, F& f) { f(t.a); f(t.b); }
template <class F> __host__ __device__ inline void dip_ptrs(DipConvDesc& d, F& f) {
    f(d.x); dip_ptrs(d.tr, f); f(d.wp); f(d.bias); f(d.y); f(d.stats); f(d.ws);
    f(d.bnb_y); f(d.bnb_state); f(d.bnb_partials); f(d.bnb_partials_thin); f(d.wp3);
}
template <class F> __host__ __device__ inline void dip_ptrs(DipWgradDesc& d, F& f) {
    f(d.x); dip_ptrs(d.tr, f); f(d.dy); f(d.partial); f(d.bias_partial);
}
template <class F> __host__ __device__ inline void dip_ptrs(DipGradSrc& s, F& f) { f(s.g); f(s.tw); }
template <class F> __host__ __device__ inline void dip_ptrs(DipBnFin& b, F& f) {
    f(b.gamma); f(b.beta); f(b.state); f(b.running_mean); f(b.running_var); f(b.ticket);
}
template <class F> __host__ __device__ inline void dip_ptrs(DipBnbFin& b, F& f) { f(b.dgamma); f(b.dbeta); f(b.coef); f(b.ticket); }
template <class F> __host__ __device__ inline void dip_ptrs(DipUpcatDesc& d, F& f) {
    f(d.s); dip_ptrs(d.ts, f); f(d.d); dip_ptrs(d.td, f); f(d.cat); f(d.stats);
}
template <class F> __host__ __device__ inline void dip_ptrs(DipLossHeadDesc& d, F& f) {
    f(d.u); dip_ptrs(d.tr, f); f(d.w); f(d.bias); f(d.target); f(d.mask); f(d.out); f(d.partials); f(d.loss);
}

struct DipShiftF {
    long long off;
    template <class T> __host__ __device__ void operator()(T*& p) const {
        if (p != nullptr) p = reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + (uintptr_t)off);
    }
};
struct DipRangeF {                     // host: is every non-NULL pointer inside [lo, hi)?
    uintptr_t lo, hi;
    bool ok;
    template <class T> void operator()(T*& p) {
        if (p != nullptr) {
            const uintptr_t v = reinterpret_cast<uintptr_t>(p);
            if (v < lo || v >= hi) ok = false;
        }
    }
};

template <class T, class F>
__host__ __device__ inline void dip_visit(T& v, F& f) {
    using U = std::remove_cv_t<T>;
    if constexpr (std::is_pointer_v<U>) f(v);
    else if constexpr (std::is_arithmetic_v<U> || std::is_enum_v<U> || std::is_same_v<U, std::nullptr_t>) { (void)f; }
    else dip_ptrs(v, f);
}
template <class T>
__host__ __device__ inline T dip_gshift(T v, long long off) {
    DipShiftF f{off};
    dip_visit(v, f);
    return v;
}
template <class... A>
inline bool dip_group_in_range(const DipGroupCtx& g, A... a) {
    DipRangeF f{reinterpret_cast<uintptr_t>(g.base), reinterpret_cast<uintptr_t>(g.base) + (uintptr_t)g.row, true};
    (dip_visit(a, f), ...);
    if (!f.ok) dip_group_fault("a pointer of a grouped launch lies outside instance 0's slab");
    return f.ok;
}

// ---------------------------------------------------------------------------------------------
// kernel side of a `bool GRP` kernel:
//     template <..., bool GRP = false>
//     __global__ void k(const DipConvDesc d_, const float* __restrict__ p_, int n, const DipGrpArg<GRP> grp) {
//         DIP_GRP_DESC(DipConvDesc, d);          // const DipConvDesc& d: d_ itself (solo) or a shifted copy (grouped)
//         DIP_GRP_PTR(const float*, p);          // p = p_ (solo) or p_ advanced to this workgroup's instance
//         ... body as before; blockIdx.z -> dip_grp_z<GRP>(grp) ...
// ---------------------------------------------------------------------------------------------
template <bool GRP, class T>
__device__ __forceinline__ const T& dip_pick(const T& shifted, const T& solo) {
    if constexpr (GRP) return shifted;
    else return solo;
}
template <bool GRP>
__device__ __forceinline__ long long dip_grp_off(const DipGrpArg<GRP>& grp) {
    if constexpr (GRP) return (long long)(blockIdx.z / (unsigned)grp.gz) * grp.stride;
    else return 0;
}
template <bool GRP>
__device__ __forceinline__ unsigned dip_grp_z(const DipGrpArg<GRP>& grp) {        // blockIdx.z of the solo launch (unsigned, as it is)
    if constexpr (GRP) return blockIdx.z % (unsigned)grp.gz;
    else return blockIdx.z;
}
#define DIP_GRP_DESC(T, d)                                                   \
    T d##sh_;                                                                \
    if constexpr (GRP) d##sh_ = dip_gshift(d##_, dip_grp_off<GRP>(grp));     \
    const T& d = dip_pick<GRP>(d##sh_, d##_)
#define DIP_GRP_PTR(T, p) T p = GRP ? dip_gshift(p##_, dip_grp_off<GRP>(grp)) : p##_

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// a kernel without a grouped form: host loop
template <class K, class... A>
inline void dip_launch(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t st, A... a) {
    const DipGroupCtx& g = *dip_group_ctx();
    if (g.ninst <= 1) { kern<<<grid, block, lds, st>>>(a...); return; }
    if (!dip_group_in_range(g, a...)) return;
    for (int b = 0; b < g.ninst; ++b) kern<<<grid, block, lds, st>>>(dip_gshift(a, (long long)b * g.stride)...);
}

// a `bool GRP` kernel pair: kern = the GRP = false instantiation, kern_g = GRP = true; `a` = the arguments in front of
// the trailing group argument.  A kernel that needs hipFuncAttributeMaxDynamicSharedMemorySize needs it for BOTH (the
// caller's once-per-device block; K / KG are pointer TYPES here, shared by every kernel of the same signature).
template <unsigned FAM, class K, class KG, class... A>
inline void dip_launch_pair(K kern, KG kern_g, dim3 grid, dim3 block, size_t lds, hipStream_t st, A... a) {
    const DipGroupCtx& g = *dip_group_ctx();
    if (g.ninst <= 1) { kern<<<grid, block, lds, st>>>(a..., DipNoGrp{}); return; }
    if (!dip_group_in_range(g, a...)) return;
    if (g.native & FAM) {
        const int gz = (int)grid.z;
        grid.z = (unsigned)(gz * g.ninst);
        kern_g<<<grid, block, lds, st>>>(a..., DipGrp{g.stride, gz});
        return;
    }
    for (int b = 0; b < g.ninst; ++b) kern<<<grid, block, lds, st>>>(dip_gshift(a, (long long)b * g.stride)..., DipNoGrp{});
}

// hipFuncAttributeMaxDynamicSharedMemorySize for both kernels of a pair
template <class K, class KG>
inline hipError_t dip_pair_lds_attr(K kern, KG kern_g, int lds_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern_g), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}
