/*
 * dip_hip.h -- C ABI of libdip_hip.so: the MI355X (gfx950) kernels behind the
 * deep-image-prior optimisation loop (skip-net forward + backward + Adam).
 *
 * The reference (DmitryUlyanov/deep-image-prior) has NO native code and no FFI: its arithmetic
 * is PyTorch `nn` modules.  Each entry point below therefore cites the reference *call site* of
 * the PyTorch op(s) it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every tensor is fp32, device memory, NHWC ("pixel-major": [H][W][Cs]) unless stated;
 *     Cs (channel stride) is a multiple of 4, channels [C, Cs) are zero;
 *   - the library allocates nothing and never synchronises: all buffers are caller-owned
 *     (PyTorch caching allocator), every launch goes to the `stream` argument (a hipStream_t),
 *     so the launch sequence is hipGraph-capturable;
 *   - return value: 0 on success, otherwise a hipError_t (or -1 for an unsupported
 *     configuration); dip_last_error() returns a description of the CALLING THREAD's last failure (thread-local).
 *   - one process per GPU, calls come from the thread that owns the stream; the error slot, the open group
 *     (dip_group_begin) and the native-family mask (dip_group_native) are per thread.
 */
#ifndef DIP_HIP_H
#define DIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIP_ABI_VERSION 8

#define DIP_PAD_ZERO 0
#define DIP_PAD_REFLECT 1
#define DIP_PAD_REPLICATE 2   /* nn.ReplicationPad2d: the Downsampler inside conv(), models/downsampler.py:56-61 */

/* DipTransform.slope codes for activations other than LeakyReLU (act_fun='Swish' | 'ELU') */
#define DIP_ACT_SWISH (-1.0f)
#define DIP_ACT_ELU (-2.0f)
#define DIP_ACT_RELU (-3.0f)  /* act_fun given as the module CLASS nn.ReLU (models/common.py:90-91: `return act_fun()`) */

#define DIP_UP_NEAREST 0
#define DIP_UP_BILINEAR 1

int dip_abi_version(void);
/* First 16 hex digits of the sha256 over the sources this binary was built from (csrc/*.hip, csrc/*.h, this header; see
 * __graft_entry__.source_id()); "unknown" for a build outside the recipe.  No reference counterpart: provenance of a
 * measured binary (VERDICT r04 next #9c). */
const char* dip_build_id(void);
const char* dip_last_error(void);
/* PCI address ("0000:d9:00.0", lower case hex) of HIP device `device`, for locating its sysfs directory
 * (/sys/bus/pci/devices/<address>: hwmon power / clock sensors, numa_node).  No reference counterpart: bench.py's
 * `device` / power blocks and its rank -> NUMA-node pinning use it.  buf: >= 13 bytes. */
int dip_device_pci_bus_id(int device, char* buf, int len);


/* ---------------------------------------------------------------- command lists ------------- */
/* The static launch list of one direction of the skip-net issued by ONE call (csrc/dip_list.hip) instead of one foreign
 * call per launch -- the host half of "an iteration is a static launch list".  No reference counterpart: the reference's
 * launches are issued op by op by PyTorch's dispatcher from nn.Module.forward / autograd (utils/common_utils.py:223-230).
 * A command is LAUNCH (fn = dip_list_fn_id("dip_conv_igemm") ..., the call's arguments in `slots`: one 8-byte slot per
 * parameter in declaration order -- ints sign-extended to 64 bits, floats in the low 4 bytes, pointers as addresses --
 * INCLUDING a last slot for the trailing `stream` parameter, which dip_list_run fills with streams[stream]), RECORD
 * (hipEventRecord(events[event], streams[stream])) or WAIT (hipStreamWaitEvent(streams[stream], events[event])).
 * The caller owns slots, streams and events; dip_list_run allocates nothing and never synchronises (hipGraph-capturable).
 * Returns 0, or the rc of the first failing command with its index in *failed_at (dip_last_error() describes it). */
#define DIP_CMD_LAUNCH 0
#define DIP_CMD_RECORD 1
#define DIP_CMD_WAIT 2
typedef struct DipCmd {
    int32_t kind, fn, stream, event;
    uint64_t* slots;
    int32_t nslots, reserved;
} DipCmd;
int dip_list_fn_id(const char* name);          /* -1: not a stream-launching entry point */
int dip_list_fn_nargs(int fn);                 /* parameters of that entry point, the trailing stream included */
int dip_list_run(const DipCmd* cmds, int n, void* const* streams, int nstreams, void* const* events, int nevents,
                 int* failed_at);
/* n HIP events without timing (hipEventDisableTiming) for RECORD / WAIT commands */
int dip_events_create(void** events, int n);
int dip_events_destroy(void** events, int n);

/* ---------------------------------------------------------------- grouped execution --------- */
/* B independent fits in ONE launch list: B copies of the skip-net of models/skip.py:45-100 --
 * own weights, own BatchNorm statistics, own Adam state, own input and target -- with identical architecture and sizes.
 * The caller carves EVERY buffer of a fit (arenas, activations, scratch, the DipPackRec tables, input, target, loss)
 * from one slab per instance; the slabs are the rows of one [ninst][stride_bytes] allocation, laid out identically.  It
 * builds the descriptors of instance 0 and issues the launch list once between dip_group_begin and dip_group_end: every
 * launch of this library then serves all instances, instance b with each non-NULL pointer argument and descriptor field
 * advanced by b * stride_bytes (base / row_bytes = slab of instance 0: a pointer outside it fails the call, rc -1).  Plans,
 * tile walks and summation orders are those of a solo launch, so each instance's results are bit-identical to the same fit
 * run on its own.  Kernels with a grouped form run B instances in one dispatch (gridDim.z x B); the others are dispatched B
 * times by the library (csrc/dip_group.h).  Host-side state, thread-local: a group replicates the launches of the thread
 * that opened it only; hipGraph-capturable like a solo list.
 * dip_group_native: bit mask of the kernel families whose one-dispatch form is in use (default all; DIP_GROUP_NATIVE);
 * mask < 0 only queries.  Returns the previous mask.  dip_group_size: instances of the open group (1: none). */
int dip_group_begin(int ninst, long long stride_bytes, const void* base, long long row_bytes);
int dip_group_end(void);
int dip_group_size(void);
int dip_group_native(int mask);

/* Per-channel input transform fused into a consumer's loader:
 *   u = act(t),  t = a[c]*x + b[c];  act = max(t, slope*t) for slope in (0, 1] (slope = 1 -> affine
 *   only), t*sigmoid(t) for slope == DIP_ACT_SWISH, ELU(alpha=1) for slope == DIP_ACT_ELU, max(t, 0) for DIP_ACT_RELU
 * This is BatchNorm2d(train)-apply (models/common.py:95-96) + act() (models/common.py:76-92:
 * LeakyReLU(0.2), Swish, ELU, none) with a = gamma*rstd, b = beta - mean*a.  a == NULL -> identity. */
typedef struct DipTransform {
    const float* a;
    const float* b;
    float slope;
} DipTransform;

/* ---------------------------------------------------------------- layout / head ------------ */
/* NCHW [C][H*W] -> NHWC [H*W][Cs] (pad channels zeroed).  Boundary of `net(net_input)`:
 * get_noise() makes [1,C,H,W] NCHW (utils/common_utils.py:140). */
int dip_nchw_to_nhwc(const float* src, float* dst, int C, int HW, int Cs, void* stream);
/* NHWC [H*W][Cs] -> NCHW [C][H*W], optionally adding into dst (grad wrt net_input). */
int dip_nhwc_to_nchw(const float* src, float* dst, int C, int HW, int Cs, int accumulate, void* stream);
/* out[c][p] = sigmoid(y[p][c]) (or copy): nn.Sigmoid, models/skip.py:97-98; NHWC -> NCHW. */
int dip_head_fwd(const float* y, float* out, int C, int HW, int Cs, int sigmoid, void* stream);
/* dy[p][c] = gout[c][p] * out[c][p]*(1-out[c][p]) (sigmoid backward); NCHW -> NHWC. */
int dip_head_bwd(const float* gout, const float* out, float* dy, int C, int HW, int Cs, int sigmoid,
                 void* stream);

/* ---------------------------------------------------------------- weights ----------------- */
/* One record per Conv2d: repacks OIHW weights (nn.Conv2d, models/common.py:120) from the flat
 * parameter arena into the MFMA B-operand layouts used by dip_conv_igemm:
 *   forward : Wf[tap][c/4][o (CoutP32)][c%4]          = W[o][c][tap]
 *   dgrad   : Wd[tap][o/4][c (CinP32)][o%4]           = W[o][c][KS*KS-1-tap]
 * offsets are in floats; dgrad_off < 0 skips the dgrad pack. */
typedef struct DipPackRec {
    int64_t w_off;      /* into `params` */
    int64_t fwd_off;    /* into `packed` */
    int64_t dgrad_off;  /* into `packed`, or -1 */
    int32_t Cout, Cin, KS;
    int32_t CinP4, CoutP32, CoutP4, CinP32;
} DipPackRec;
int dip_pack_weights(const float* params, float* packed, const DipPackRec* recs_dev, int nrec,
                     int max_elems, void* stream);
/* The same weights as THREE bf16 planes for the bf16-pipe convolution (conv_bf3.hip: w == w1 + w2 + w3 exactly, 8 + 8 + 8
 * significand bits), [tap][k / 16][plane][n][16 k] bf16:  forward: k = input channel, n = output channel (CoutP32 of them);
 * data gradient: k = output channel, n = input channel (CinP32), taps flipped.  Offsets in bf16 elements, -1 = skip. */
typedef struct DipPackRec3 {
    int64_t w_off;      /* into `params` (floats) */
    int64_t fwd_off;    /* into `packed3` (bf16 elements), or -1 */
    int64_t dgrad_off;  /* into `packed3`, or -1 */
    int32_t Cout, Cin, KS;
    int32_t nchF, CoutP32, nchD, CinP32;   /* nchF = ceil(CinP4 / 16), nchD = ceil(CoutP4 / 16) */
} DipPackRec3;
int dip_pack_weights_bf3(const float* params, void* packed3, const DipPackRec3* recs_dev, int nrec, long long max_elems,
                         void* stream);

/* ---------------------------------------------------------------- convolution ------------- */
/* In-launch finalisation of the BatchNorm2d that follows a launch (dip_upcat_fwd_fin; opt-in, DIP_TICKET_FIN=1): the last
 * workgroup to arrive (fence-free ticket, one counter per 32-channel block) reduces the partial rows in fp64 and
 * writes the state block + running statistics -- exactly what dip_bn_finalize computes in a launch of its own
 * (nn.BatchNorm2d training forward, models/common.py:95-96).  state == NULL: off (the caller runs dip_bn_finalize).
 * `ticket` points at zero-initialised counters (ceil(Cout / 32) of them) that the launch leaves at zero. */
typedef struct DipBnFin {
    const float* gamma;
    const float* beta;
    float eps, momentum;
    float* state;                /* [4][Cs]: mean, rstd, a, b */
    int32_t Cs, C;
    float* running_mean;         /* or NULL */
    float* running_var;
    uint32_t* ticket;
} DipBnFin;
/* The same for phase 2 of a BatchNorm backward (dip_bn_bwd_stats_fin, dip_upsample_bwd_stats_crop_fin): dgamma, dbeta (may be NULL) and coef [2][bnb_Cs] = {k1 = S1 / npix, k2 = S2 / npix}, what
 * dip_bn_bwd_finalize computes.  coef == NULL: off. */
typedef struct DipBnbFin {
    float* dgamma;
    float* dbeta;
    float* coef;
    int32_t C, npix;
    uint32_t* ticket;
} DipBnbFin;

/* Implicit-GEMM convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32):
 *   y[q][o] (+)= bias[o] + sum_{tap,c} u[src(q,tap)][c] * Wp[tap][c][o],  u = transform(x)
 * per spatial dim  v = q*stride + k - off ; pad_mode reflect mirrors v into [0,Hv), zero drops
 * it;  Hv = (Hin-1)*dil + 1 ; dil == 2 reads x[v/2] for even v and 0 for odd v.
 * Replaces ReflectionPad2d + Conv2d (models/common.py:114-124) for the forward
 * (off = pad, dil = 1) and autograd's ConvolutionBackward data-gradient (transposed conv for
 * the stride-2 layers, models/skip.py:64) with flipped weights (off = KS-1-zero_pad, stride 1,
 * dil = forward stride; with reflection padding the result is the gradient on the PADDED
 * domain, folded back by dip_bn_bwd_stats / dip_fold_add).
 * stats != NULL: per-workgroup BatchNorm partials {count, mean, M2}[CoutP32] per tile
 * (Chan/Welford form) for the BatchNorm2d that follows the conv (models/common.py:95-96). */
typedef struct DipConvDesc {
    const float* x;
    int32_t Hin, Win, Cx, Cin;   /* Cx = channel stride of x, Cin = channels convolved (mult of 4) */
    DipTransform tr;
    const float* wp;             /* packed weights (see DipPackRec) */
    const float* bias;           /* [Cout] or NULL */
    float* y;
    int32_t Hout, Wout, Cy, Cout; /* Cy = channel stride of y */
    int32_t y_pitch;             /* pixels per output row in memory (0 -> Wout); lets a 1x1 dgrad
                                    accumulate into the interior of a padded gradient buffer */
    int32_t ks, stride, pad_mode, off, dil, accumulate;
    float* stats;                /* [stats_rows][3][CoutP32] or NULL (rows: dip_conv_plan) */
    int32_t ksplit;              /* <= 1: one pass.  > 1: split-K over (channel chunk, tap) units into
                                    `ws`, then a fixed-order reduction (small images: too few tiles
                                    to fill 256 CUs otherwise); take the value from dip_conv_plan */
    float* ws;                   /* split-K workspace, ksplit*Hout*Wout*Cy floats, or NULL */
    /* Optional (data-gradient launches): phase 1 of the backward of the BatchNorm(+activation) that PRODUCED this
     * convolution's input in the forward pass, fused into the epilogue.  The launch's output is the gradient g wrt
     * that BatchNorm's activated output on a domain padded by bnb_pad (reflection padding: the fold of the ring is
     * linear, so every padded position contributes with the activation of its mirror pixel):
     *   {sum g*act'(z), sum g*act'(z)*xhat},  z = a*y + b,  per output tile and channel
     *   -> bnb_partials [dip_conv_ntiles(Hout, Wout)][2][bnb_Cs]   (8x16-pixel tiles; columns >= 128 of a
     *      129..132-column gradient come from conv_thin4: bnb_partials_thin [dip_conv_thin4_ntiles()][2][bnb_Cs])
     * i.e. what dip_bn_bwd_stats(dz = NULL) computes in a pass of its own over g and y (autograd
     * NativeBatchNormBackward + LeakyReluBackward of models/common.py:82,96); dip_bn_bwd_finalize2 reduces the
     * rows.  bnb_y == NULL: off.  Only one-pass launches (dip_conv_bnb_fusable). */
    const float* bnb_y;          /* raw output y of the conv in front of that BatchNorm, [H][W][bnb_Cy], H = Hout - 2*bnb_pad */
    const float* bnb_state;      /* its state block [4][bnb_Cs]: mean, rstd, a, b */
    float* bnb_partials;
    float* bnb_partials_thin;
    int32_t bnb_Cy, bnb_Cs, bnb_pad;
    float bnb_slope;
    /* Optional: the same weights as three bf16 planes (DipPackRec3).  When set, dip_conv_igemm runs the 3x3 stride-1 layers
     * with >= 256 tiles on the bf16 matrix pipe: every fp32 operand split EXACTLY into three bf16 terms (8 + 8 + 8
     * significand bits), the cross products -- each exact in fp32 -- accumulated in fp32 by v_mfma_f32_32x32x16_bf16: the
     * same roundings as an fp32 fmaf chain (measured error vs fp64 below the fp32 MFMA's).  Default: eight of the nine
     * (lo x lo, < 2^-32 of a*b, is left out; half the fp32 matrix-pipe time, conv_bf3.hip); DIP_CONV_BF3=9: all nine;
     * =0: fp32 MFMA everywhere; =6: without the three smallest products (each < 2^-24 of a*b).  NULL: fp32 MFMA. */
    const void* wp3;
} DipConvDesc;
int dip_conv_igemm(const DipConvDesc* d, void* stream);
/* number of 8x16 output tiles */
int dip_conv_ntiles(int Hout, int Wout);
/* number of 16x16 output tiles of dip_conv_thin4 (rows of bnb_partials_thin) */
int dip_conv_thin4_ntiles(int Hout, int Wout);
/* 1 when dip_conv_igemm runs `d` in one pass through a kernel whose epilogue can emit the fused BatchNorm-backward
 * partials (bnb_* fields; evaluated with bnb_y ignored): no split-K, not the phase mode, not the N = 160 variant */
int dip_conv_bnb_fusable(const DipConvDesc* d);
/* launch plan of one convolution: split-K factor, rows of the statistics partial buffer and the
 * split-K workspace size in floats (0 when *ksplit == 1) */
int dip_conv_plan(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int* ksplit, int* stats_rows,
                  int64_t* ws_floats);
/* the same for a layer that is NOT taken by the bf16-pipe kernel although its shape is (no DipConvDesc.wp3, a transform over
 * more than 512 input channels, fused bnb_* partials): without the "96..255 tiles run the 64-column bf16 form in one pass"
 * rule, i.e. split-K as the fp32 kernels want it.  The engine re-plans with it when dip_conv_bf3_eligible(d) == 0. */
int dip_conv_plan_fp32(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int* ksplit, int* stats_rows,
                       int64_t* ws_floats);
/* launch plan of the data gradient of a stride-2 3x3 convolution (a dil == 2 descriptor; Hout x Wout is
 * the gradient's domain, Cin/Cout the descriptor's): the LDS-DMA kernel's phase mode evaluates it as 4
 * dense sub-filter convolutions, one per output-pixel parity (9 taps per 4 pixels instead of 36), with its
 * own split-K bound.  Other filter sizes / column counts get dip_conv_plan's answer. */
int dip_conv_plan_dil2(int Hout, int Wout, int Cin, int Cout, int ks, int* ksplit, int* stats_rows,
                       int64_t* ws_floats);
/* which kernel dip_conv_igemm launches for `d` (diagnostic; bench.py attributes its HIP-event times
 * with it): 0 = conv_igemm_kernel (operands staged through registers: stride 2, 5x5),
 * 1 = conv_igemm_dma_kernel (LDS-DMA staging: stride-1 1x1 / 3x3), 2 = conv_igemm_kernel N=160 variant
 * (3x3, 129..160 output channels, split-K), 3 = conv_thin4_mfma_kernel (columns 0..Cout-129: the tap sum pulled out of the
 * contraction, v_mfma_f32_16x16x4_f32 fed straight from global memory; the vector-ALU conv_thin4_kernel above 128 input channels
 * and with fused BatchNorm-backward partials) + conv_igemm_dma_kernel (the other 128 columns): 3x3 data gradients towards a
 * 132-channel tensor,
 * 4 = conv_igemm_dma_kernel in phase mode (dil == 2: data gradient of a stride-2 3x3 convolution),
 * 5 = conv_igemm_dma_kernel in strided-forward mode (3x3, stride 2: the input split by pixel parity),
 * 6 = conv1x1_res_kernel (1x1, 128 -> 97..128 channels, >= 256x256 pixels: persistent workgroups, weights in registers),
 * 7 = conv_bf3_kernel (3x3 stride 1 on the bf16 matrix pipe; with DIP_CONV_BF3_1X1=1 also 1x1 layers from 256 tiles:
 * conv_bf3_k1_kernel, measured slower per iteration and off by default), 8 = conv_thin_kernel (<= 64 channels in and out, 3x3 / 5x5,
 * 16x16x4 MFMA tiles, all taps' weights LDS-resident: the high-resolution layers of the narrow nets) */
int dip_conv_variant(const DipConvDesc* d);
/* The two launches behind variant 3, exported so that a caller can put them on DIFFERENT streams (they
 * write disjoint columns of the same output): columns [0, ncols) (ncols = Cout - 128 <= 4) of a 3x3
 * stride-1 transform-free convolution (matrix-pipe form up to 128 input channels, vector ALU beyond), and one 128-column block starting at column
 * n_base on the LDS-DMA kernel.  Same descriptor as dip_conv_igemm. */
int dip_conv_thin4(const DipConvDesc* d, int ncols, void* stream);
int dip_conv_igemm_dma_cols(const DipConvDesc* d, int n_base, void* stream);
/* Thin layers (round 6, csrc/conv_thin.hip): 3x3 / 5x5, stride 1 / 2 (forward) or the stride-1 data gradient, <= 64 input
 * and output channels, more than DIP_THIN_MIN_PIXELS (4625) output pixels -- the 16 / 32 / 64-channel layers of the
 * 'library' inpainting net (inpainting.ipynb:222-232) and the snail net (denoising.ipynb:143-150) at their high
 * resolutions; same contract as dip_conv_igemm (which dispatches to it: dip_conv_variant == 8), one pass, stats rows =
 * dip_conv_ntiles.  dip_conv_thin_shape_ok: the shape part of the eligibility (what dip_conv_plan knows). */
int dip_conv_thin(const DipConvDesc* d, void* stream);
int dip_conv_thin_eligible(const DipConvDesc* d);
int dip_conv_thin_shape_ok(int Hout, int Wout, int Cin, int Cout, int ks, int stride);
/* Low-resolution layers (models/skip.py:57-91 at depth >= 2: <= 64x64 outputs in the notebooks' nets): ONE launch per
 * convolution instead of conv + split-K finish.  A workgroup computes one 32-pixel x 32-channel tile; its 4 / 8 / 16 waves
 * split K, each with its whole slice of operand loads in flight at once, straight from L2 into the MFMA registers (no LDS
 * staging: such a layer is latency-bound, not bandwidth-bound), and are summed through LDS in a fixed order; the epilogue
 * emits the BatchNorm partials (`stats`: [rows][3][CoutP32], for dip_bn_finalize) or -- data gradients -- the d->bnb_*
 * partials ([rows][2][bnb_Cs], for dip_bn_bwd_finalize2 with nblk_lo = 0); any 1..160 output columns (no conv_thin4 side
 * launch).  1x1 / 3x3, stride 1 / 2, dil 1 / 2 (dil == 2: per output-parity class, only the taps that hit non-zero
 * positions), every padding mode and activation; d->ksplit / d->ws are ignored.
 * dip_conv_small_eligible: the shape is served AND Hout * Wout <= DIP_SMALL_MAX_PIXELS (default 4624 = 68 x 68);
 * dip_conv_small_rows: rows of the partial buffers (0: shape not served). */
int dip_conv_small(const DipConvDesc* d, void* stream);
int dip_conv_small_eligible(const DipConvDesc* d);
int dip_conv_small_rows(const DipConvDesc* d);
/* Data gradient of a reflection-padded 3x3 stride-1 convolution WITHOUT the padded domain (models/common.py:116-121 +
 * autograd's ReflectionPad2dBackward): the caller computes the interior H x W positions as a plain zero-padded correlation
 * (dip_conv_igemm with off = 1: 512 tiles at 256^2 instead of the 561 of the (H+2) x (W+2) domain) and this launch adds
 * what the ring of the padded domain folds onto the frame rows 1, H-2 / columns 1, W-2 -- per frame pixel the gradient of
 * the <= 3 ring positions that mirror onto it, accumulated into y.  `d` = the interior descriptor (x = dy [H][W],
 * y = gradient [H][W][Cy], ks 3, stride 1, dil 1, off 1, DIP_PAD_ZERO, no transform / bias / stats).  The gradient buffer
 * then needs no fold (DipGradSrc.pad = 0).  dip_conv_dgrad_ring_ok: shape served and H * W <= DIP_DGRAD_RING_MAX
 * (default 300000: up to 512 x 512). */
int dip_conv_dgrad_ring(const DipConvDesc* d, void* stream);
int dip_conv_dgrad_ring_ok(const DipConvDesc* d);
/* bf16-pipe convolution (see DipConvDesc.wp3): eligibility of `d`, the launch of columns [n_base, n_base + ncols) and the
 * number of cross products in use (0 = off, 9, 6) */
int dip_conv_bf3_eligible(const DipConvDesc* d);
int dip_conv_bf3_cols(const DipConvDesc* d, int n_base, int ncols, void* stream);
int dip_conv_bf3_terms(void);
/* overrides DIP_CONV_BF3 for this process: 0 (off), 6, 8 (without lo x lo: < 2^-32 of a product), 9; -1 = back to the environment */
int dip_conv_bf3_set_terms(int terms);
/* second half of a split-K dispatch (d->ksplit > 1): fixed-order sum of the workspace slices, bias,
 * store, BatchNorm partials.  dip_conv_igemm calls it itself; exported for per-kernel timing. */
int dip_conv_splitk_finish(const DipConvDesc* d, void* stream);

/* Weight gradient (autograd ConvolutionBackward, weight + bias part):
 *   dW[o][c][tap] = sum_q dy[q][o] * u[src(q,tap)][c],  db[o] = sum_q dy[q][o]
 * Two stages, deterministic: dip_conv_wgrad writes `nsplit` partial slabs
 * [nsplit][tap][CinP32][CoutP32] (+ [nsplit][CoutP32] bias partials), dip_wgrad_reduce sums the
 * slabs in a fixed order and writes OIHW gradients into the grad arena. */
typedef struct DipWgradDesc {
    const float* x;
    int32_t Hin, Win, Cx, Cin;
    DipTransform tr;
    const float* dy;
    int32_t Hout, Wout, Cdy, Cout;
    int32_t ks, stride, pad_mode, off;
    float* partial;
    float* bias_partial;          /* or NULL */
    int32_t nsplit;
    int32_t tap_groups;           /* 3x3 MFMA kernel: the 9 taps are spread over 1, 3 or 9 workgroups (more, lighter
                                     workgroups for low-resolution layers); 0 = 1.  From dip_wgrad_plan2. */
    int32_t chan_block;           /* 1x1 MFMA kernel: input channels per workgroup / 32: 4 (default, 0) or 1 */
} DipWgradDesc;
int dip_conv_wgrad(const DipWgradDesc* d, void* stream);
/* dip_conv_wgrad routes the 3x3 stride-1 layers with >= 32 input channels and >= 256 x 256 outputs to the bf16 matrix pipe
 * (wgrad_bf3.hip: the exact three-way operand split of DipConvDesc.wp3's kernel, here applied to BOTH activations while
 * they are staged transposed in LDS; same slabs, same dip_wgrad_reduce): eligibility, the launch, and the
 * (tap, channel)-packed <= 4-channel tail of a 132-channel layer that stays on the fp32 MFMA (phase 2 of the fp32 kernel on
 * its own).  DIP_WGRAD_NO_BF3=1 / DIP_CONV_BF3=0 switch it off. */
int dip_wgrad_bf3_eligible(const DipWgradDesc* d);
int dip_wgrad_bf3(const DipWgradDesc* d, void* stream);
int dip_conv_wgrad_tail(const DipWgradDesc* d, void* stream);
/* The same rows as a streaming kernel (round 6, wgrad_tail.hip; dip_conv_wgrad_tail dispatches to it): (tap, channel) packed
 * into the rows of the fp32 MFMA, K = pairs of output pixels straight from global memory, one pass over dy, the slabs of
 * `d->nsplit` walkers written like the bf16-pipe kernel's.  _ok: 3x3, stride 1, 1..4 channels behind whole 32-channel chunks
 * (DIP_WGRAD_TAIL_OLD=1 switches it off). */
int dip_wgrad_tail_stream_ok(const DipWgradDesc* d);
int dip_wgrad_tail_stream(const DipWgradDesc* d, void* stream);
/* Thin layers (round 6, csrc/wgrad_thin.hip): 3x3 / 5x5, stride 1 / 2, 8..64 input channels, <= 64 output channels, >= 4096
 * output pixels -- the 16 / 32 / 64-channel convs of the 'library' and snail nets (inpainting.ipynb:222-232,
 * denoising.ipynb:143-150): 16-channel x 16-column x 4-pixel MFMA tiles, the taps spread over the workgroup's waves; same
 * slabs, same dip_wgrad_reduce.  dip_conv_wgrad dispatches to it; nsplit from dip_wgrad_plan / dip_wgrad_plan2
 * (= dip_wgrad_thin_nsplit for these shapes). */
int dip_wgrad_thin(const DipWgradDesc* d, void* stream);
int dip_wgrad_thin_eligible(const DipWgradDesc* d);
int dip_wgrad_thin_shape_ok(int Hout, int Wout, int Cin, int Cout, int ks, int stride);
int dip_wgrad_thin_nsplit(int Hout, int Wout, int Cin, int Cout, int ks, int stride);
/* number of 4x16 output tiles walked by the wgrad workgroups (upper bound for nsplit) */
int dip_conv_wgrad_ntiles(int Hout, int Wout);
/* nsplit (number of partial slabs) to run dip_conv_wgrad with; mandatory for 1x1 convs with
 * Cout <= 8 and for 3x3 / 5x5 / 7x7 convs with <= 4 INPUT channels (the first conv of a net), which take thin
 * vector-ALU streaming kernels with one slab per block */
int dip_wgrad_plan(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int* nsplit);
/* the full launch plan: nsplit plus the tap_groups / chan_block fields of DipWgradDesc.  Large layers
 * get (nsplit as dip_wgrad_plan, 1, 4); layers with <= 256 pixel tiles (<= 128x128 outputs) trade
 * taps-per-workgroup and slabs against workgroup count with a small cost model so that a 16x16 layer
 * runs ~150 light workgroups instead of 4 heavy ones. */
int dip_wgrad_plan2(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int* nsplit, int* tap_groups,
                    int* chan_block);
int dip_wgrad_reduce(const float* partial, const float* bias_partial, int nsplit, int ks, int Cin,
                     int Cout, float* dw /*OIHW*/, float* dbias /*or NULL*/, void* stream);

/* ---------------------------------------------------------------- BatchNorm (train mode) -- */
/* state block per BatchNorm: 4 rows of Cs floats: mean, rstd, a, b  (a = gamma*rstd, b = beta - mean*a) */
/* Combine per-tile partials -> batch statistics, update running stats (momentum 0.1, unbiased
 * var), nn.BatchNorm2d training forward (models/common.py:95-96). */
int dip_bn_finalize(const float* partials, int ntiles, int Cstride, int C, const float* gamma,
                    const float* beta, float eps, float momentum, float* state, int Cs,
                    float* running_mean, float* running_var, void* stream);

/* Source of an incoming activation gradient `du` for pixel (r,c), channel ch:
 *   du = sum over folded positions of  g[((r+pad)*Wg + (c+pad)) * Cg + choff + ch]
 * fold == 1 adds the mirror images of the reflection-padded border (adjoint of
 * nn.ReflectionPad2d, models/common.py:116-118), fold == 2 the ring positions that nn.ReplicationPad2d clamps onto a
 * border pixel (models/downsampler.py:56-61); Hg = H + 2*pad, Wg = W + 2*pad. */
typedef struct DipGradSrc {
    const float* g;
    int32_t pad, fold, Cg, choff;
    /* Optional crop window (win_h > 0, then pad == 0): g is a [win_h][win_w] tensor that covers rows win_y..win_y+win_h-1,
     * columns win_x..win_x+win_w-1 of the [H][W] activation; du = 0 outside (adjoint of Concat's centre crop,
     * models/common.py:29-37, for the branch that was cropped). */
    int32_t win_y, win_x, win_h, win_w;
    /* Optional thin 1x1 convolution in front of the activation (tw != NULL; then pad == 0, no window, choff == 0): g is the
     * gradient [H][W][Cg] of the OUTPUT of a bias + 1x1 conv with tn <= 4 output channels (the net's last nn.Conv2d,
     * models/skip.py:98: num_channels_up[0] -> num_output_channels) and
     *   du[ch] = sum_{j < tn} g[(r*W + c) * Cg + j] * tw[j * tcw + ch]          (tw = that conv's OIHW weight, tcw = its Cin)
     * is evaluated where it is consumed -- the conv's data gradient (autograd ConvolutionBackward, input part) is never
     * written to memory: one launch and three passes over an activation-sized tensor less per iteration. */
    const float* tw;
    int32_t tn, tcw;
} DipGradSrc;

/* BatchNorm+LeakyReLU backward, phase 1:  dz = du * (a*y+b > 0 ? 1 : slope); writes dz (unless
 * dz == NULL: then phase 3 is dip_bn_bwd_apply_src, which recomputes it) and the per-block partial
 * sums {sum dz, sum dz*xhat}.  (autograd NativeBatchNormBackward + LeakyReluBackward of
 * models/common.py:82,96.) */
int dip_bn_bwd_stats(const DipGradSrc* src, const float* y, int H, int W, int Cy, int C,
                     const float* state, int Cs, float slope, float* dz, int Cdz,
                     float* partials /*[nblk][2][Cs]*/, int nblk, void* stream);
int dip_bn_bwd_nblk(int H, int W, int C);
/* phase 1 + in-launch phase 2 (fin->coef != NULL; needs dip_fin_rows_ok(nblk, C)): the last block to arrive writes
 * dgamma, dbeta and coef, no dip_bn_bwd_finalize launch */
int dip_bn_bwd_stats_fin(const DipGradSrc* src, const float* y, int H, int W, int Cy, int C,
                         const float* state, int Cs, float slope, float* dz, int Cdz,
                         float* partials /*[nblk][2][Cs]*/, int nblk, const DipBnbFin* fin, void* stream);
/* phase 2: reduce partials -> dgamma, dbeta (grad arena) and k1 = S1/N, k2 = S2/N in `coef` [2][Cs] */
int dip_bn_bwd_finalize(const float* partials, int nblk, int Cs, int C, int npix, float* dgamma,
                        float* dbeta, float* coef, void* stream);
/* phase 2 over two partial buffers: channels < c_lo from partials_lo [nblk_lo][2][Cs] (the conv_thin4 columns of a
 * 129..132-column data gradient with fused statistics), the others from partials [nblk][2][Cs]; c_lo % 4 == 0 */
int dip_bn_bwd_finalize2(const float* partials, int nblk, const float* partials_lo, int nblk_lo, int c_lo, int Cs,
                         int C, int npix, float* dgamma, float* dbeta, float* coef, void* stream);
/* phase 3 (in place): dy = a * (dz - k1 - xhat*k2) */
int dip_bn_bwd_apply(float* dz, int Cdz, const float* y, int Cy, int npix, int C, const float* state,
                     int Cs, const float* coef, void* stream);
/* phase 3 straight from the gradient source (one tensor pass less than stats-with-dz + apply):
 *   dy = a * (du * lrelu'(a*y+b) - k1 - xhat*k2) */
int dip_bn_bwd_apply_src(const DipGradSrc* src, const float* y, int H, int W, int Cy, int C,
                         const float* state, int Cs, float slope, const float* coef, float* dy, int Cdy,
                         void* stream);
/* Phases 2 + 3 in one launch (round 6): every block of the apply launch reduces the `nrows` partial rows [nrows][2][Cs] of
 * phase 1 for itself in its prologue (fp64, fixed order: the same k1, k2 in every block) and block 0 writes dgamma, dbeta
 * (may be NULL) and coef -- no dip_bn_bwd_finalize launch in the dependent chain.  Same reference ops as above.
 * dip_bn_bwd_fin_rows_ok: 1 when the engine should use the form (nrows <= DIP_BNB_FIN_MAX_ROWS, default 320: every block
 * reads nrows * C * 8 bytes from L2). */
int dip_bn_bwd_fin_rows_ok(int nrows, int C);
int dip_bn_bwd_apply_fin(float* dz, int Cdz, const float* y, int Cy, int npix, int C, const float* state, int Cs,
                         const float* partials, int nrows, float* dgamma, float* dbeta, float* coef, void* stream);
int dip_bn_bwd_apply_src_fin(const DipGradSrc* src, const float* y, int H, int W, int Cy, int C, const float* state, int Cs,
                             float slope, const float* partials, int nrows, float* dgamma, float* dbeta, float* coef,
                             float* dy, int Cdy, void* stream);
/* The three phases in ONE launch for low-resolution activations (round 6; csrc/bn_bwd_one.hip): a workgroup owns four
 * channels of the whole H x W plane, so S1 = sum dz and S2 = sum dz * xhat are workgroup-local (per-thread fp32 sums, a
 * fixed-order fp64 reduction) and dy = a * (du * act'(a*y+b) - S1/N - xhat * S2/N) follows in the same launch; dgamma,
 * dbeta and coef [2][Cs] are written as dip_bn_bwd_finalize writes them (each may be NULL).  Same reference ops as
 * dip_bn_bwd_stats / _finalize / _apply_src (autograd NativeBatchNormBackward + LeakyReluBackward of
 * models/common.py:82,96).  dip_bn_bwd_one_ok: 1 when the engine should use this form (npix <= DIP_BNB_ONE_MAX_PIXELS,
 * default 1024 = 32 x 32: above that a 4-channel slice of the plane costs 8x its bytes in L1 fills; 0 switches it off). */
int dip_bn_bwd_one_ok(int npix, int C);
int dip_bn_bwd_one(const DipGradSrc* src, const float* y, int H, int W, int Cy, int C, const float* state, int Cs,
                   float slope, float* dy, int Cdy, float* dgamma, float* dbeta, float* coef, void* stream);
/* Fold a (reflection-)padded gradient back onto the image and emit it NCHW: gradient wrt
 * `net_input` for get_params('net,input') (utils/common_utils.py:47-49). */
int dip_fold_to_nchw(const DipGradSrc* src, int H, int W, int C, float* dst, void* stream);
/* The same, NHWC out ([H][W][Cd]): dy of a conv whose output feeds a padded consumer directly, without a BatchNorm in
 * between (the stride-1 conv in front of the Lanczos Downsampler of conv(..., downsample_mode='lanczos2')). */
int dip_fold_to_nhwc(const DipGradSrc* src, int H, int W, int C, float* dst, int Cd, void* stream);

/* ---------------------------------------------------------------- upsample + concat ------- */
/* cat[p][0:ns]      = T_s(s[p])                      (skip branch, Concat child "0")
 * cat[p][ns:ns+nd]  = upsample2x(T_d(d))[p]          (deeper branch, nn.Upsample models/skip.py:81)
 * and the {count, mean, M2} partials of the BatchNorm2d(ns+nd) that follows (models/skip.py:55).
 * Concat: models/common.py:11-42.  Default geometry (Hs == 0): s is [H][W], d is [(H+1)/2][(W+1)/2] (an odd size drops
 * the last up-sampled row / column: the centre crop with offset 0).  General centre crop (models/common.py:29-37; pooling
 * nets or skip-less scales at sizes that 2^scales does not divide): Hs, Ws = size of s, (os_y, os_x) = its crop offset;
 * Hd, Wd = size of d, (od_y, od_x) = crop offset inside its [2*Hd][2*Wd] up-sampled image; H, W = the common (min) size. */
typedef struct DipUpcatDesc {
    const float* s; int32_t Cs_s, ns; DipTransform ts;      /* ns may be 0 (s == NULL) */
    const float* d; int32_t Cs_d, nd; DipTransform td;      /* low-res [H/2][W/2][Cs_d] */
    int32_t H, W, mode;                                      /* output (high-res) size */
    float* cat; int32_t Cs_cat;
    float* stats; int32_t nblk;                              /* [nblk][3][Cs_cat] */
    int32_t Hs, Ws, os_y, os_x;                              /* 0: default geometry */
    int32_t Hd, Wd, od_y, od_x;
} DipUpcatDesc;
int dip_upcat_fwd(const DipUpcatDesc* d, void* stream);
/* the same launch + in-launch finalisation of the concat BatchNorm (fin->state != NULL; needs dip_fin_rows_ok(d->nblk, ns + nd)) */
int dip_upcat_fwd_fin(const DipUpcatDesc* d, const DipBnFin* fin, void* stream);
/* 1 when a launch that writes `rows` partial rows of C channels may finalise them itself (<= 256 rows, <= 256 channels;
 * DIP_NO_TICKET_FIN=1 switches every in-launch finalisation off) */
int dip_fin_rows_ok(int rows, int C);
int dip_upcat_nblk(int H, int W, int C);

/* nn.AvgPool2d(2,2) behind a stride-1 conv (conv(..., downsample_mode='avg'), models/common.py:101-104):
 * x [H][W][Cx] -> y [H/2][W/2][Cy] (H, W even or floored) + the {count, mean, M2} partials
 * [nblk][3][Cy] of the BatchNorm that follows (nblk = dip_upcat_nblk(H/2, W/2, C); stats may be
 * NULL); the adjoint writes dx[r][c] = dy[r/2][c/2] / 4 (0 on a floored odd border). */
int dip_avgpool2_fwd(const float* x, int H, int W, int Cx, int C, float* y, int Cy, float* stats, int nblk,
                     void* stream);
int dip_avgpool2_bwd(const float* dy, int H, int W, int Cdy, int C, float* dx, int Cdx, void* stream);
/* nn.MaxPool2d(2,2) behind a stride-1 conv (conv(..., downsample_mode='max'), models/common.py:105-106):
 * same contract as dip_avgpool2_fwd; the adjoint routes dy to the first maximal element of each 2x2
 * window of x (ATen's tie rule), recomputing the arg-max from x, zeros elsewhere. */
int dip_maxpool2_fwd(const float* x, int H, int W, int Cx, int C, float* y, int Cy, float* stats, int nblk,
                     void* stream);
int dip_maxpool2_bwd(const float* dy, const float* x, int H, int W, int Cdy, int Cx, int C, float* dx, int Cdx,
                     void* stream);

/* Adjoint of the 2x upsample fused with the LeakyReLU/BatchNorm backward phase 1 of the
 * deeper branch: du_d = upsample2x^T(dcat[:, choff:choff+nd]); dz = du_d * lrelu'(a*y+b);
 * partial sums as in dip_bn_bwd_stats.  (autograd UpsampleBilinear2DBackward / Nearest.) */
int dip_upsample_bwd_stats(const float* dcat, int Cs_cat, int choff, int H, int W, int mode,
                           const float* y, int Cy, int C, const float* state, int Cs, float slope,
                           float* dz, int Cdz, float* partials, int nblk, void* stream);
/* the same with the general crop geometry of DipUpcatDesc: dcat is [H][W], the deeper branch [Hd][Wd] (y, dz), cropped at
 * (od_y, od_x) of its up-sampled image; nblk = dip_bn_bwd_nblk(Hd, Wd, C) */
int dip_upsample_bwd_stats_crop(const float* dcat, int Cs_cat, int choff, int H, int W, int Hd, int Wd, int od_y,
                                int od_x, int mode, const float* y, int Cy, int C, const float* state, int Cs,
                                float slope, float* dz, int Cdz, float* partials, int nblk, void* stream);

/* dip_upsample_bwd_stats_crop + in-launch phase 2 of the deeper branch's BatchNorm backward (fin as in dip_bn_bwd_stats_fin) */
int dip_upsample_bwd_stats_crop_fin(const float* dcat, int Cs_cat, int choff, int H, int W, int Hd, int Wd, int od_y,
                                    int od_x, int mode, const float* y, int Cy, int C, const float* state, int Cs,
                                    float slope, float* dz, int Cdz, float* partials, int nblk, const DipBnbFin* fin,
                                    void* stream);

/* Adjoint of the up-sampling + all three BatchNorm-backward phases of the deeper branch in ONE launch (the form of
 * dip_bn_bwd_one with dip_upsample_bwd_stats_crop's gradient source; autograd UpsampleBilinear2DBackward / Nearest +
 * NativeBatchNormBackward + LeakyReluBackward, models/skip.py:81, models/common.py:82,96): dy [Hd][Wd][Cdy] = gradient wrt
 * the raw output y of the deeper branch's last conv.  Use when dip_bn_bwd_one_ok(Hd * Wd, C). */
int dip_upsample_bwd_one(const float* dcat, int Cs_cat, int choff, int H, int W, int Hd, int Wd, int od_y, int od_x,
                         int mode, const float* y, int Cy, int C, const float* state, int Cs, float slope, float* dy,
                         int Cdy, float* dgamma, float* dbeta, float* coef, void* stream);

/* ---------------------------------------------------------------- optimiser --------------- */
/* torch.optim.Adam(lr) defaults, one fused launch over a flat arena
 * (utils/common_utils.py:225-230):  m += (1-b1)(g-m); v = b2 v + (1-b2) g^2;
 * p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps),  bc = 1 - beta^step. */
int dip_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                  double beta2, double eps, int step, void* stream);

/* net_input = z + sigma * N(0,1): the closure's reg-noise line (denoising.ipynb:208-209),
 * counter-based Philox4x32-10 + Box-Muller, NCHW in / NCHW out (elementwise). */
int dip_noise_axpy(const float* z, float* out, int64_t n, float sigma, uint64_t seed, uint64_t offset,
                   void* stream);

/* ---------------------------------------------------------------- device-side iteration state */
/* Everything that changes from one optimisation iteration to the next on the HOST side of the
 * reference loop (utils/common_utils.py:226-230: the step count inside torch.optim.Adam; the
 * device RNG position behind `noise.normal_()`, denoising.ipynb:209) kept in DEVICE memory, so the
 * launch list of one iteration is static and can be captured once and replayed as a hipGraph. */
typedef struct DipIterState {
    uint64_t step;        /* Adam step count t (0 before the first step) */
    float step_size;      /* lr / (1 - beta1^t), written by dip_adam_tick */
    float bc2_sqrt;       /* sqrt(1 - beta2^t) */
} DipIterState;
/* t <- t + 1 and the two bias-correction scalars of torch.optim.Adam, computed in double. */
int dip_adam_tick(DipIterState* st, double lr, double beta1, double beta2, void* stream);
/* dip_adam_step with step_size / bc2_sqrt read from `st` (call dip_adam_tick first). */
int dip_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, double beta1, double beta2,
                      double eps, const DipIterState* st, void* stream);
/* dip_noise_axpy whose Philox offset lives at *offset_dev and is advanced by ceil(n/4) afterwards. */
int dip_noise_axpy_dev(const float* z, float* out, int64_t n, float sigma, uint64_t seed, uint64_t* offset_dev,
                       void* stream);
/* dip_noise_axpy_dev with the seed in device memory too: state_dev = {offset, seed} (two uint64).  In a group every
 * instance reads its own pair, i.e. draws its own stream. */
int dip_noise_axpy_dev2(const float* z, float* out, int64_t n, float sigma, uint64_t* state_dev, void* stream);
/* *counter += inc (one thread; ordering by the stream). */
int dip_counter_add(uint64_t* counter, uint64_t inc, void* stream);

/* ---------------------------------------------------------------- fused loss head ------------ */
/* The tail of the closure in two launches: output conv (1x1, Cout <= 4, weights straight from the
 * OIHW parameter arena) + nn.Sigmoid (models/skip.py:96-98) + optional mask + torch.nn.MSELoss
 * (denoising.ipynb:177,219: mse(out, img_noisy); inpainting.ipynb:310: mse(out * mask, img * mask),
 * the mean runs over ALL Cout*HW elements in both cases).
 *   forward : out[o][p] = sigmoid(sum_c w[o][c] * tr(u[p][c]) + bias[o])            (NCHW)
 *             *loss = 1/(Cout*HW) * sum (out*m - target*m)^2
 *             one lane per pixel (conv on the 4x4x1 MFMA), LDS tree per block, one partial per block,
 *             then a one-block fp64 sum of the partials in a fixed order (second launch inside the
 *             call): deterministic, no float atomics.
 *   backward: dy[p][o] = *gscale * 2/(Cout*HW) * (out*m - target*m) * m * out*(1-out)  (NHWC, stride Cy)
 *             = grad wrt the output conv's result, consumed by dip_conv_wgrad / dip_conv_igemm. */
typedef struct DipLossHeadDesc {
    const float* u;       /* [HW][Cu] input activation of the output conv (raw conv output of the producer) */
    int Cu, Cin;
    DipTransform tr;      /* the producer's BatchNorm + LeakyReLU */
    const float* w;       /* [Cout][Cin] 1x1 weights (OIHW) */
    const float* bias;    /* [Cout] or NULL */
    int Cout, HW, sigmoid;
    const float* target;  /* [Cout][HW] NCHW */
    const float* mask;    /* [mask_c][HW] NCHW or NULL; mask_c in {1, Cout} */
    int mask_c;
    float* out;           /* [Cout][HW] NCHW: the network output */
    float* partials;      /* nblk floats */
    int nblk;             /* dip_loss_head_nblk(HW, Cin) */
    float* loss;          /* 1 float */
} DipLossHeadDesc;
int dip_loss_head_nblk(int HW, int Cin);
int dip_loss_head_fwd(const DipLossHeadDesc* d, void* stream);
int dip_loss_head_bwd(const DipLossHeadDesc* d, const float* gscale, float* dy, int Cy, void* stream);

/* ---------------------------------------------------------------- closure bookkeeping ---- */
/* The per-iteration bookkeeping of the notebooks' closures without host round trips
 * (denoising.ipynb:214-248): EMA of the output, PSNR against the noisy / clean image, and the
 * "fall back to the last checkpoint if psrn_noisy dropped by more than 5 dB" rule.
 *   out_avg <- first ? out : out_avg*exp_weight + out*(1-exp_weight)
 *   record  <- {loss, mse_noisy, mse_gt, mse_gt_sm, psnr_noisy, psnr_gt, psnr_gt_sm, fell_back}
 *              (psnr = -10 log10(mse), data range 1; the gt entries are 0 when gt == NULL;
 *               `loss` is a device scalar or NULL)
 *   state   =  {psnr_noisy_last, restore, have_last, snapshot}; zero it before the first call.
 *              With check_backtrack != 0 (the notebook's `if i % show_every:`): restore = 1 when
 *              psnr_noisy - psnr_noisy_last < -backtrack_db, otherwise snapshot = 1 and
 *              psnr_noisy_last <- psnr_noisy.
 * `partial` is scratch of 4*dip_fit_monitor_nblk(n) floats.  All tensors NCHW / flat, n elements. */
int dip_fit_monitor_nblk(int64_t n);
int dip_fit_monitor(const float* out, const float* noisy, const float* gt, float* out_avg, int64_t n,
                    float exp_weight, int first, const float* loss, float* partial, float* record,
                    float* state, int check_backtrack, float backtrack_db, void* stream);
/* applies the decision in `state` to the flat parameter arena: restore -> params = snapshot,
 * snapshot -> snapshot = params, neither -> nothing (the notebook's net_param.data.copy_(...) /
 * last_net = [x.detach().cpu() ...], :242-247) */
int dip_arena_backtrack(float* params, float* snapshot, int64_t n, const float* state, void* stream);

/* ---------------------------------------------------------------- Lanczos down-sampler ---- */
/* Downsampler.forward (models/downsampler.py:65-71): ReplicationPad2d(pad) + depth-wise
 * k x k stride-`factor` correlation with the fixed taps; NCHW [C][H][W] -> [C][H/f][W/f]. */
int dip_lanczos_down_fwd(const float* x, const float* taps, float* y, int C, int H, int W, int k,
                         int factor, int pad, void* stream);
int dip_lanczos_down_bwd(const float* gy, const float* taps, float* gx, int C, int H, int W, int k,
                         int factor, int pad, void* stream);
/* The same module with its dense weight being optimised -- get_params('down', ...), utils/common_utils.py:44-46:
 * ReplicationPad2d(pad) + Conv2d(C, C, k, stride=factor) with weight w [C][C][k][k] (OIHW) and bias [C] (may be NULL),
 * models/downsampler.py:88-101; NCHW.  _bwd_data: autograd's gradient wrt the input, _bwd_weight: wrt weight (dw, same
 * layout as w) and bias (db, may be NULL); x is the UNPADDED input of the forward. */
int dip_down_dense_fwd(const float* x, const float* w, const float* bias, float* y, int C, int H, int W, int k,
                       int factor, int pad, void* stream);
int dip_down_dense_bwd_data(const float* gy, const float* w, float* gx, int C, int H, int W, int k, int factor,
                            int pad, void* stream);
int dip_down_dense_bwd_weight(const float* gy, const float* x, float* dw, float* db, int C, int H, int W, int k,
                              int factor, int pad, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIP_HIP_H */
