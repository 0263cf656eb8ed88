// Device-side bookkeeping of the notebooks' closures.
//
// The reference closure (denoising.ipynb:214-248) does, every iteration, on the host:
//   out_avg = out_avg * exp_weight + out * (1 - exp_weight)                  (:214-217)
//   psrn_noisy / psrn_gt / psrn_gt_sm = compare_psnr(...) on .cpu().numpy()  (:223-225)  -> 3 D2H + sync
//   if i % show_every: fall back to the last checkpoint when psrn_noisy dropped by > 5 dB,
//                      else checkpoint all parameters on the CPU               (:238-248) -> 8.9 MB D2H
// Here the same arithmetic stays on the GPU: one streaming pass updates the EMA and the three
// squared-error sums, a one-block kernel turns them into a record {loss, 3 MSEs, 3 PSNRs, fell_back}
// and takes the back-tracking decision, and dip_arena_backtrack applies it to the flat parameter
// arena against a device-resident snapshot.  Nothing synchronises; the host reads the records when
// it wants to print.
#include "dip_common.h"
#include "dip_group.h"

namespace {

__global__ __launch_bounds__(256) void fit_monitor_partials_kernel(const float* __restrict__ out,
                                                                   const float* __restrict__ noisy,
                                                                   const float* __restrict__ gt, float* __restrict__ avg,
                                                                   int64_t n, float w, int first, float* __restrict__ partial) {
    __shared__ float sh[3][256];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float o = out[i];
        const float a = first ? o : fmaf(avg[i], w, o * (1.f - w));
        avg[i] = a;
        const float dn = o - noisy[i];
        s0 = fmaf(dn, dn, s0);
        if (gt != nullptr) {
            const float g = gt[i];
            s1 = fmaf(o - g, o - g, s1);
            s2 = fmaf(a - g, a - g, s2);
        }
    }
    sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1; sh[2][threadIdx.x] = s2;
    for (int s = 128; s >= 1; s >>= 1) {          // fixed pairing order: deterministic
        __syncthreads();
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + s];
        }
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 4 + 0] = sh[0][0];
        partial[blockIdx.x * 4 + 1] = sh[1][0];
        partial[blockIdx.x * 4 + 2] = sh[2][0];
    }
}

// record: [loss, mse_noisy, mse_gt, mse_gt_sm, psnr_noisy, psnr_gt, psnr_gt_sm, fell_back]
// state:  [psnr_noisy_last, restore_flag, have_last, snapshot_flag]
__global__ __launch_bounds__(64) void fit_monitor_finalize_kernel(const float* __restrict__ partial, int nblk, int64_t n,
                                                                  int have_gt, const float* __restrict__ loss,
                                                                  float* __restrict__ record, float* __restrict__ state,
                                                                  int check, float thresh_db) {
    if (threadIdx.x != 0) return;
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = 0; b < nblk; ++b)
        for (int k = 0; k < 3; ++k) s[k] += (double)partial[b * 4 + k];
    float mse[3], psnr[3];
    for (int k = 0; k < 3; ++k) {
        mse[k] = (float)(s[k] / (double)n);
        psnr[k] = (float)(-10.0 * log10(s[k] / (double)n));      // data_range = 1
    }
    record[0] = loss != nullptr ? loss[0] : 0.f;
    record[1] = mse[0]; record[2] = have_gt ? mse[1] : 0.f; record[3] = have_gt ? mse[2] : 0.f;
    record[4] = psnr[0]; record[5] = have_gt ? psnr[1] : 0.f; record[6] = have_gt ? psnr[2] : 0.f;
    float restore = 0.f, snap = 0.f;
    if (check) {
        if (state[2] != 0.f && psnr[0] - state[0] < -thresh_db) {
            restore = 1.f;                                        // "Falling back to previous checkpoint."
        } else {
            snap = 1.f;
            state[0] = psnr[0];
            state[2] = 1.f;
        }
    }
    state[1] = restore;
    state[3] = snap;
    record[7] = restore;
}

__global__ __launch_bounds__(256) void arena_backtrack_kernel(float* __restrict__ params, float* __restrict__ snapshot,
                                                              int64_t n, const float* __restrict__ state) {
    const float restore = state[1], snap = state[3];
    if (restore == 0.f && snap == 0.f) return;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        if (restore != 0.f) *reinterpret_cast<f32x4*>(params + i) = *reinterpret_cast<const f32x4*>(snapshot + i);
        else *reinterpret_cast<f32x4*>(snapshot + i) = *reinterpret_cast<const f32x4*>(params + i);
    } else {
        for (int64_t j = i; j < n; ++j) {
            if (restore != 0.f) params[j] = snapshot[j];
            else snapshot[j] = params[j];
        }
    }
}

}  // namespace

extern "C" int dip_fit_monitor_nblk(int64_t n) {
    int64_t b = (n + 1023) / 1024;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int dip_fit_monitor(const float* out, const float* noisy, const float* gt, float* out_avg, int64_t n,
                               float exp_weight, int first, const float* loss, float* partial, float* record,
                               float* state, int check_backtrack, float backtrack_db, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n <= 0 || out == nullptr || noisy == nullptr || out_avg == nullptr) DIP_FAIL("fit_monitor: bad arguments");
    const int nblk = dip_fit_monitor_nblk(n);
    dip_launch(fit_monitor_partials_kernel, dim3(nblk), dim3(256), 0, st, out, noisy, gt, out_avg, n, exp_weight,
                       first, partial);
    DIP_CHECK_LAUNCH();
    dip_launch(fit_monitor_finalize_kernel, dim3(1), dim3(64), 0, st, partial, nblk, n, gt != nullptr ? 1 : 0,
                       loss, record, state, check_backtrack, backtrack_db);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_arena_backtrack(float* params, float* snapshot, int64_t n, const float* state, void* stream) {
    if (n <= 0) return 0;
    if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(snapshot)) & 15)
        DIP_FAIL("arena_backtrack: arenas must be 16-byte aligned");
    const int64_t quads = (n + 3) / 4;
    dip_launch(arena_backtrack_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), params, snapshot, n, state);
    DIP_CHECK_LAUNCH();
    return 0;
}
