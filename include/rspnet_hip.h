/*
 * rspnet_hip.h — C ABI of librspnet_hip.so: the MI355X (gfx950) kernels behind RSPNet's pretext hot path.
 *
 * The reference (PeihaoChen/RSPNet) has NO native interface: every op on the path is an ATen call made from
 * Python.  Each entry point below therefore cites the *reference call site* whose ATen op
 * it replaces; the Python binding a maintainer would add is the ctypes stub shown in INTEGRATION.md (our own
 * host side, rspnet_amd/_lib.py, is exactly that stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless its type says otherwise; tensors are dense NDHWC
 *     ("rows" = N*D*H*W positions, channels contiguous, row pitch given by an `ld` argument in floats);
 *   - weights cross the boundary in the reference's own layout (Cout,Cin,kT,kH,kW) (nn.Conv3d.weight,
 *     SURVEY.md §A.5) and are re-packed on device by rsp_conv3d_pack_*;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work (no sync, no allocation: graph-capturable);
 *   - return value: 0 = ok, negative = RSP_E* (see rsp_strerror); nothing is written on a negative return.
 */
#ifndef RSPNET_HIP_H
#define RSPNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSP_OK 0
#define RSP_EINVAL (-1)      /* bad geometry / null pointer / misaligned pointer */
#define RSP_EWORKSPACE (-2)  /* workspace too small */
#define RSP_ELAUNCH (-3)     /* hipLaunch error (hipGetLastError text via rsp_last_error) */
#define RSP_EUNSUPPORTED (-4)

const char* rsp_strerror(int code);
const char* rsp_last_error(void);
/* Demangled template instance of the FIRST matrix kernel the calling thread's most recent rsp_conv3d_{fwd,dgrad,dgrad_packed,
 * wgrad} call launched (written by the launcher itself; "" if the call launched none).  Cross-checks rsp_conv3d_kernel_name. */
const char* rsp_last_conv_kernel(void);
int rsp_version(void);

/* ---------------------------------------------------------------------------------------------------------
 * 3-D convolution (nn.Conv3d, groups=1, dilation=1): models/c3d.py:21-52, models/resnet.py:20-28,131-138,
 * models/s3dg.py:19-20,48-51,60, models/r2plus1d_vcop.py:54-55,66-67.
 * x: [N,Di,Hi,Wi,(in_ld)]  y: [N,Do,Ho,Wo,(out_ld)]  with Do = (Di+2pT-kT)/sT+1 ...
 * ------------------------------------------------------------------------------------------------------- */
typedef struct rsp_conv3d_desc {
  int32_t N, Di, Hi, Wi, Cin;
  int32_t Do, Ho, Wo, Cout;
  int32_t kT, kH, kW;
  int32_t sT, sH, sW;
  int32_t pT, pH, pW;
  int32_t in_ld;   /* floats between consecutive positions of x (>= Cin; lets x be a channel slice) */
  int32_t out_ld;  /* same for y */
} rsp_conv3d_desc;

/* Number of floats of the packed forward weight.  The layout is private to the library and chosen per descriptor
 * ([Cout][taps][Cin] with K padded to a multiple of 4 for the implicit-GEMM kernel, [taps][64][4] for the 4-channel stem
 * kernel); pack and forward must be called with the same descriptor. */
size_t rsp_conv3d_packed_fwd_elems(const rsp_conv3d_desc* d);
/* (Cout,Cin,kT,kH,kW) -> forward-packed. */
int rsp_conv3d_pack_fwd(const rsp_conv3d_desc* d, const float* w_ref, float* w_packed, void* stream);

/* Number of stat tiles of the output grid (128-row tiles, or 8x16 output patches on the stem path): stat partials are
 * [tiles][Cout][2] (sum, sum of squares of the bias-free conv output) — consumed by rsp_bn_finalize.  */
int32_t rsp_conv3d_stat_tiles(const rsp_conv3d_desc* d);
size_t rsp_conv3d_fwd_workspace(const rsp_conv3d_desc* d);
/* y = conv(x, w) + bias.  bias and stat_partials may be NULL. */
int rsp_conv3d_fwd(const rsp_conv3d_desc* d, const float* x, const float* w_packed, const float* bias, float* y,
                   float* stat_partials, void* workspace, size_t workspace_bytes, void* stream);

/* dgrad: dx = conv_transpose(dy, w).  Replaces autograd's conv backward-input for every conv on the path
 * (loss.backward(), pretrain.py:164).  w_ref in reference layout; packs per stride-parity class into workspace. */
size_t rsp_conv3d_dgrad_workspace(const rsp_conv3d_desc* d);
int rsp_conv3d_dgrad(const rsp_conv3d_desc* d, const float* dy, const float* w_ref, float* dx, void* workspace,
                     size_t workspace_bytes, void* stream);

/* dgrad over weights already re-packed into the per-stride-class layouts (rsp_conv3d_pack_jobs(which = 1) + rsp_pack_run):
 * the workspace then only holds the K-split partials (rsp_conv3d_dgrad_workspace(d) is always enough). */
size_t rsp_conv3d_packed_dgrad_elems(const rsp_conv3d_desc* d);
int rsp_conv3d_dgrad_packed(const rsp_conv3d_desc* d, const float* dy, const float* w_packed, float* dx, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Batched weight re-pack.  The weights change once per step (torch.optim.SGD step, pretrain.py:165; momentum update of the key
 * encoder, builder_diffspeed_diffloss.py:337-343), so every convolution's packed copy is rebuilt once per step: instead of one
 * small launch per convolution (and per dgrad stride class) the host describes the re-packs once — the weight and packed
 * buffers do not move — uploads the array, and replays it with ONE launch per step and encoder.
 *   rsp_conv3d_pack_jobs: fills (host memory) the jobs that re-pack w_ref (reference layout (Cout_src, Cin_src, kT, kH, kW)) into
 *     w_packed for descriptor d; d->Cout / d->Cin may exceed the source dims, the excess is zero (channel padding, e.g. the
 *     3 -> 4 channel stems; for those the library notes, per w_packed address, that the filters have three real input channels, and
 *     rsp_conv3d_fwd on that buffer skips the zero fourth channel's multiply-adds).  which = 0: forward layout (1 job,
 *     rsp_conv3d_packed_fwd_elems floats); which = 1: all dgrad
 *     stride-class layouts (<= sT*sH*sW jobs, rsp_conv3d_packed_dgrad_elems floats).  Returns the number of jobs or RSP_E*.
 *   rsp_pack_run: executes n_jobs jobs stored in DEVICE memory; max_blocks = the largest `blocks` field among them (grid.x).
 *   rsp_conv3d_pack_forget: the owner of a packed buffer calls this before releasing it: the library drops what it noted about
 *     that address (the three-channel mark of a stem), so that an unrelated buffer which later lands on the same address never
 *     inherits it (an unmarked stem buffer runs all four channel steps: always correct). */
typedef struct rsp_pack_job {
  const void* src;
  void* dst;
  int64_t total;
  int32_t kind, Cout_src, Cin_src, kT, kH, kW;
  int32_t transpose, O, C, Kld, nTd, nTh, nTw, k0d, k0h, k0w, kstepd, ksteph, kstepw, ntaps;
  int32_t blocks, reserved;   /* workgroups this job needs (one per packed row) */
} rsp_pack_job;
int32_t rsp_conv3d_pack_jobs(const rsp_conv3d_desc* d, int32_t which, int32_t Cout_src, int32_t Cin_src, const float* w_ref,
                             float* w_packed, rsp_pack_job* jobs, int32_t max_jobs);
int rsp_pack_run(const rsp_pack_job* jobs_device, int32_t n_jobs, int32_t max_blocks, void* stream);
void rsp_conv3d_pack_forget(const void* w_packed);

/* wgrad: dw (reference layout, overwritten) = sum over positions of dy ⊗ im2col(x); dbias (nullable) = sum dy. */
size_t rsp_conv3d_wgrad_workspace(const rsp_conv3d_desc* d);
int rsp_conv3d_wgrad(const rsp_conv3d_desc* d, const float* x, const float* dy, float* dw_ref, float* dbias,
                     void* workspace, size_t workspace_bytes, void* stream);
/* The same for a convolution whose channels are zero padded in the descriptor (d->Cout / d->Cin) but not in the parameter:
 * dw_ref is (cout_valid, cin_valid, kT, kH, kW), the gradients of the padding channels are dropped.  dbias must be NULL. */
int rsp_conv3d_wgrad_v(const rsp_conv3d_desc* d, const float* x, const float* dy, float* dw_ref, int32_t cout_valid,
                       int32_t cin_valid, void* workspace, size_t workspace_bytes, void* stream);

/* The weight-gradient kernels stream a per-row geometry table (input byte offset + padding-validity bits of each output position:
 * a function of the geometry alone).  rsp_conv3d_wgrad / _v compute it in a pre-pass of every call; a caller that keeps one table
 * per geometry (rsp_conv3d_rowgeom_bytes, rsp_conv3d_rowgeom: one launch, once) passes it to rsp_conv3d_wgrad_t and saves that
 * launch — 59 per S3D-G step.  rowgeom_table may be NULL (then as rsp_conv3d_wgrad_v; dbias only with unpadded channels). */
size_t rsp_conv3d_rowgeom_bytes(const rsp_conv3d_desc* d);
int rsp_conv3d_rowgeom(const rsp_conv3d_desc* d, void* table, void* stream);
int rsp_conv3d_wgrad_t(const rsp_conv3d_desc* d, const float* x, const float* dy, float* dw_ref, float* dbias, int32_t cout_valid,
                       int32_t cin_valid, const void* rowgeom_table, void* workspace, size_t workspace_bytes, void* stream);

/* Name of the kernel template instance the library launches for this descriptor (which: 0 forward, 1 dgrad, 2 wgrad;
 * 16-byte aligned tensors assumed) -- lets bench.py label its roofline block with the kernel that actually dominates a
 * backbone and match it to the rocprofv3 kernel-trace row.  Static string, never NULL. */
const char* rsp_conv3d_kernel_name(const rsp_conv3d_desc* d, int which);

/* Share (0..1] of the convolution's algorithmic multiply-adds — 2*MACs counted with the taps that read the zero padding, as
 * cuDNN / oneDNN under the reference's nn.Conv3d count them (models/c3d.py:21-52) — that the launched kernels actually execute:
 * the implicit-GEMM walks skip K chunks (forward / dgrad) and row chunks (wgrad) that are padding for a whole tile.  Host
 * arithmetic only (the kernels' own planning code), no GPU.  which: 0 forward, 1 dgrad, 2 wgrad (frame-granular estimate).
 * bench.py prices roofline.executed_frac with it next to the algorithmic rate. */
double rsp_conv3d_executed_fraction(const rsp_conv3d_desc* d, int which);

/* Planning options of the convolution launchers, settable at run time (process-wide; host arithmetic only: which kernel instance and
 * which K split a descriptor gets — every choice computes the same convolution, in another summation order).  Returns the previous
 * value, or RSP_EINVAL for an unknown name.
 *   "narrow_max_tiles"    launches of fewer than this many 128-wide tiles (and 97..128 or > 160 columns) run on the 64-wide tile
 *                         (default 512, environment RSP_NARROW_MAX_TILES; 0: never; < 0: back to the default).
 *   "narrow32_max_units"  launches of at most this many 128 x 64 tiles (more than 32 columns, K of at least 8 chunks) run on the
 *                         32-wide tile with the whole K per unit instead of a K split + reduce launch (default 256, environment
 *                         RSP_NARROW32_MAX_UNITS; 0: never; < 0: back to the default).
 *   "tall_min_tiles"      33..64-column launches of at least this many 256-row tiles run on the 256 x 64 instance of the persistent
 *                         kernel (default 0 = never: measured neutral-to-negative per step in round 6; environment
 *                         RSP_TALL_MIN_TILES; < 0: back to the default).
 *   "two_level_min_chunks" slice-major 128-wide launches of at least this many 32-deep K chunks sum K in panels of 512 products
 *                         (a second accumulator set, two waves per SIMD): the CPU convolution's error level on the long-K layers
 *                         (default 0 = never: priced in round 6; environment RSP_TWO_LEVEL_MIN_CHUNKS).
 * Used by the kernel tests (an instance at sizes the checker finishes in seconds) and by tools/geom_bench.py (A/B of a plan in one
 * process).  The whole-step parity tests run under the DEFAULT plan only. */
int rsp_conv3d_set_option(const char* name, int32_t value);

/* Host evaluation of the constant division the conv kernels use to decode GEMM rows and k positions (multiply-high by a
 * host-computed magic number + shift, exact for 0 <= n < 2^31): returns n / d computed that way.  No GPU needed; exists so
 * the CPU test suite can check the derivation over the full range. */
int rsp_fastdiv_check(int d, int n);

/* ---------------------------------------------------------------------------------------------------------
 * BatchNorm3d (train mode) fused with ReLU / residual add / MaxPool3d:
 * models/c3d.py:22-24 (bn+relu+pool), models/resnet.py:61-77, models/s3dg.py:23,28-33, r2plus1d_vcop.py:59-60,116-123.
 * ------------------------------------------------------------------------------------------------------- */
/* Reduce stat partials -> mean (incl. conv bias), invstd; update running stats
 * (running_var with the unbiased n/(n-1) estimate), as F.batch_norm(training=True) does.
 * count = number of positions per channel.  scale_shift out: [2][C] = (gamma*invstd, beta - mean*gamma*invstd).
 * stat_ld (>= C) = channels per partial row: the C channels may be a slice of a wider convolution's partials (several
 * BasicConv3d that share their input run as ONE GEMM over the concatenated filters, models/s3dg.py:80-88). */
size_t rsp_bn_finalize_workspace(int32_t tiles, int32_t C);
int rsp_bn_finalize(const float* stat_partials, int32_t tiles, int32_t C, int32_t stat_ld, int64_t count, const float* conv_bias,
                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                    float* running_var, float* mean_invstd /*[2][C]*/, float* scale_shift /*[2][C]*/, void* workspace,
                    size_t workspace_bytes, void* stream);
/* The same over a convolution that ran with its output channels zero-padded from c_valid to C (models/r2plus1d_vcop.py:35-38: mid
 * channel counts such as 83 / 230 / 921 run as 84 / 232 / 924 so that rows are 16 bytes aligned): conv_bias / gamma / beta /
 * running_mean / running_var hold c_valid entries; channels [c_valid, C) get scale = shift = 0 and move no running statistic. */
int rsp_bn_finalize_v(const float* stat_partials, int32_t tiles, int32_t C, int32_t c_valid, int32_t stat_ld, int64_t count,
                      const float* conv_bias, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* mean_invstd /*[2][C]*/, float* scale_shift /*[2][C]*/, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Deferred running statistics.  With batch_stats_out != NULL rsp_bn_finalize_x does NOT move running_mean / running_var (both
 * ignored) but writes this pass's batch moments — [2][c_valid]: mean (conv bias included), unbiased variance; rsp_bn_running_update
 * later applies  r = (1 - momentum) r + momentum * moment  for a whole list of layers in one launch (jobs in device memory), the
 * update nn.BatchNorm3d does inside forward.  The two key-encoder passes of a step (builder_diffspeed_diffloss.py:445,512) go through
 * the SAME BatchNorm buffers; deferring the second pass's update lets the passes run side by side in a captured graph and still
 * leaves the buffers as two consecutive forwards would. */
typedef struct rsp_bn_ema_job {
  float* running_mean;
  float* running_var;
  const float* batch_stats; /* [2][C] */
  int32_t C;
  float momentum;
} rsp_bn_ema_job;
int rsp_bn_finalize_x(const float* stat_partials, int32_t tiles, int32_t C, int32_t c_valid, int32_t stat_ld, int64_t count,
                      const float* conv_bias, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* batch_stats_out, float* mean_invstd, float* scale_shift, void* workspace,
                      size_t workspace_bytes, void* stream);
int rsp_bn_running_update(const rsp_bn_ema_job* jobs_device, int32_t n_jobs, int32_t max_c, void* stream);

/* Standalone per-channel statistics of y (for convs whose epilogue did not produce partials): writes
 * [tiles][C][2] partials with tiles = rsp_bn_stat_tiles(rows). */
int32_t rsp_bn_stat_tiles(int64_t rows);
int rsp_bn_stats(const float* y, int64_t rows, int32_t C, int32_t ld, float* stat_partials, void* stream);

typedef struct rsp_pool3d_desc {
  int32_t N, Di, Hi, Wi, C;
  int32_t Do, Ho, Wo;
  int32_t kT, kH, kW, sT, sH, sW, pT, pH, pW; /* k=s=1,p=0: no pooling */
  int32_t in_ld, out_ld, res_ld;
} rsp_pool3d_desc;

/* out = maxpool(act(scale*y + shift (+ residual))) ; act = ReLU if relu!=0.  residual (nullable) has y's shape. */
int rsp_bn_act_pool_fwd(const rsp_pool3d_desc* d, const float* y, const float* scale_shift, const float* residual,
                        int relu, float* out, void* stream);
/* ... with per-sample channel gates ([N][C], nullable) multiplied in after the activation and before the max (see rsp_bn_gate_sums) */
int rsp_bn_act_pool_gate_fwd(const rsp_pool3d_desc* d, const float* y, const float* scale_shift, const float* residual,
                             int relu, const float* gate, float* out, void* stream);

/* The ResNet stems' bn1 -> relu -> MaxPool3d(3, 2, 1) (models/resnet.py:139,203-207) — BatchNorm apply and an OVERLAPPING 3x3x3 or
 * 1x3x3 max-pool (any stride / padding) in one pass over the convolution output: out = maxpool(act(scale*y + shift)); argmax
 * (nullable) as rsp_maxpool3d_fwd writes it (first maximum in scan order, linear input position per sample), so that
 * rsp_maxpool3d_bwd and then rsp_bn_act_pool_bwd (unit window) form its backward.  Bit-identical to rsp_bn_act_pool_fwd with a unit
 * window followed by rsp_maxpool3d_fwd; the activated tensor is neither written nor read.  rsp_bn_act_pool_fwd takes this path
 * itself for such windows (no residual).  Needs C, in_ld, out_ld multiples of 4 and 16-byte aligned pointers
 * (rsp_bn_act_maxpool_applicable). */
int rsp_bn_act_maxpool_applicable(const rsp_pool3d_desc* d);
int rsp_bn_act_maxpool_fwd(const rsp_pool3d_desc* d, const float* y, const float* scale_shift, int relu, float* out, int32_t* argmax,
                           void* stream);
/* ... with S3D-G's per-sample channel gates ([N][C], nullable) multiplied in between the activation and the pool (models/s3dg.py:105-108:
 * the two front-end sep_conv units are followed by (1,3,3) / (1,2,2) max-pools): rsp_bn_gate_sums -> this, instead of
 * rsp_bn_act_pool_gate_fwd (unit window) -> rsp_maxpool3d_fwd; same bits, and rsp_bn_act_pool_gate_fwd takes this path itself. */
int rsp_bn_act_maxpool_gate_fwd(const rsp_pool3d_desc* d, const float* y, const float* scale_shift, int relu, const float* gate, float* out,
                                int32_t* argmax, void* stream);

/* Backward of the fused block, two launches:
 *  reduce: per-channel partial sums of dz and dz*xhat (dz = grad at the BN output after pool routing + ReLU mask)
 *  apply : dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)); also d(residual) = dz (if dres != NULL),
 *          dgamma = sum(dz*xhat), dbeta = sum(dz).
 * The pooled activation is recomputed from y (saved conv output), never stored. */
size_t rsp_bn_bwd_workspace(const rsp_pool3d_desc* d);
int rsp_bn_act_pool_bwd(const rsp_pool3d_desc* d, const float* y, const float* residual, const float* dout,
                        const float* gamma, const float* mean_invstd, const float* scale_shift, int relu, float* dy,
                        float* dres, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                        void* stream);
/* channel-padded variant (see rsp_bn_finalize_v): gamma / dgamma / dbeta hold c_valid entries; padding channels have gamma = 0 */
int rsp_bn_act_pool_bwd_v(const rsp_pool3d_desc* d, const float* y, const float* residual, const float* dout,
                          const float* gamma, const float* mean_invstd, const float* scale_shift, int relu, float* dy,
                          float* dres, float* dgamma, float* dbeta, int32_t c_valid, void* workspace, size_t workspace_bytes,
                          void* stream);

/* ... behind an S3D-G self-gating unit whose forward kept no activation (rsp_bn_gate_sums, act == NULL): dout is the gradient of
 * the GATED output; gate [N][C] from the forward, dmean [N][C] from rsp_gate_bwd_params.  Unit windows, no residual. */
int rsp_bn_act_pool_bwd_g(const rsp_pool3d_desc* d, const float* y, const float* residual, const float* dout,
                          const float* gamma, const float* mean_invstd, const float* scale_shift, int relu, float* dy,
                          float* dres, float* dgamma, float* dbeta, int32_t c_valid, const float* gate, const float* dmean,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Stand-alone MaxPool3d, any window/stride/padding (models/resnet.py:139, models/s3dg.py:90,107-119).
 * argmax (nullable in forward when no backward is needed): [N,Do,Ho,Wo,C] int32, linear input position per sample. */
int rsp_maxpool3d_fwd(const rsp_pool3d_desc* d, const float* x, float* out, int32_t* argmax, void* stream);
int rsp_maxpool3d_bwd(const rsp_pool3d_desc* d, const float* dout, const int32_t* argmax, float* dx, void* stream);

/* S3D-G self-gating (models/s3dg.py:63-72): out = x * sigmoid(W·mean_p(x) + b); x:[N][P][C] (pitch in_ld), W (C,C)
 * = excitation.weight viewed (C,C,1,1,1), b (C).  Saves mean [N][C] and gate [N][C] for backward. */
size_t rsp_gate_fwd_workspace(int32_t N, int32_t P, int32_t C);
int rsp_gate_fwd(const float* x, int32_t N, int32_t P, int32_t C, int32_t in_ld, const float* w, const float* b,
                 float* out, int32_t out_ld, float* mean, float* gate, void* workspace, size_t workspace_bytes,
                 void* stream);
/* The same unit with the gate's spatial mean taken by the kernel that applies the BatchNorm in front of it
 * (sep_conv = ... BasicConv3d -> excitation, models/s3dg.py:52-72): rsp_bn_gate_sums computes a = relu(y*scale + shift), its
 * per-(sample, channel) mean and the gate in one pass over y — storing a ([N][P][C], pitch act_ld) only when `act` is given (a
 * backward will read it) — and the gated output is then either rsp_gate_apply(a) or, without the stored activation,
 * rsp_bn_act_pool_gate_fwd straight from y (optionally through the max-pool that follows the front-end units, s3dg.py:105-109).
 * Bit-identical to rsp_bn_act_pool_fwd + rsp_gate_fwd (+ rsp_maxpool3d_fwd).  Workspace: rsp_gate_fwd_workspace(N, P, C). */
int rsp_bn_gate_sums(const float* y, int32_t N, int32_t P, int32_t C, int32_t y_ld, const float* scale_shift, int relu,
                     float* act, int32_t act_ld, const float* w, const float* b, float* mean, float* gate, void* workspace,
                     size_t workspace_bytes, void* stream);
int rsp_gate_apply(const float* x, int32_t N, int32_t P, int32_t C, int32_t in_ld, const float* gate, float* out, int32_t out_ld,
                   void* stream);
/* Parameter half of the gating backward with the activation recomputed from y: dw (C,C), db (C) and dmean [N][C] (the data half
 * runs inside rsp_bn_act_pool_bwd_g).  Workspace: rsp_gate_bwd_workspace(N, P, C). */
int rsp_gate_bwd_params(const float* y, const float* scale_shift, int relu, const float* dout, int32_t N, int32_t P, int32_t C,
                        int32_t y_ld, int32_t dout_ld, const float* w, const float* mean, const float* gate, float* dw, float* db,
                        float* dmean, void* workspace, size_t workspace_bytes, void* stream);
size_t rsp_gate_bwd_workspace(int32_t N, int32_t P, int32_t C);
int rsp_gate_bwd(const float* x, const float* dout, int32_t N, int32_t P, int32_t C, int32_t x_ld, int32_t dout_ld,
                 const float* w, const float* mean, const float* gate, float* dx, int32_t dx_ld, float* dw, float* db,
                 void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Projection heads: AdaptiveAvgPool3d(1) -> Flatten -> Linear x2 -> F.normalize (moco/split_wrapper.py:138-152,163-169)
 * feat: [B][P][C] (ld = feat_ld).  w1,w2: (dim, C) row-major (nn.Linear.weight), b1,b2: (dim).
 * out: qa,qm [B][dim] unit-norm; saves pooled [B][C] and raw (pre-normalize) [2][B][dim] for backward.
 * ------------------------------------------------------------------------------------------------------- */
int rsp_head_fwd(const float* feat, int32_t B, int32_t P, int32_t C, int32_t feat_ld, const float* w1,
                 const float* b1, const float* w2, const float* b2, int32_t dim, float* out1, float* out2,
                 float* pooled, float* raw, void* stream);
size_t rsp_head_bwd_workspace(int32_t B, int32_t dim);
int rsp_head_bwd(const float* dout1, const float* dout2, const float* pooled, const float* raw, const float* w1,
                 const float* w2, int32_t B, int32_t P, int32_t C, int32_t feat_ld, int32_t dim, float* dw1,
                 float* db1, float* dw2, float* db2, float* dfeat, void* workspace, size_t workspace_bytes,
                 void* stream);

/* 'mlp' head variant (moco/split_wrapper.py:171-179: pool -> Linear(C,C) -> ReLU -> Linear(C,dim) -> F.normalize),
 * built from generic pieces.  x:[N][P][C] pitch ld; linear weights (Cout,Cin) row-major like nn.Linear.weight. */
int rsp_spatial_mean_fwd(const float* x, int32_t N, int32_t P, int32_t C, int32_t ld, float* mean, void* stream);
int rsp_spatial_mean_bwd(const float* dmean, int32_t N, int32_t P, int32_t C, int32_t ld, float* dx, void* stream);
int rsp_linear_fwd(const float* x, int32_t B, int32_t Cin, const float* w, const float* bias, int32_t Cout, int relu,
                   float* y, void* stream);
size_t rsp_linear_bwd_workspace(int32_t B, int32_t Cout);
/* y = forward output (for the ReLU mask); dx may be NULL. */
int rsp_linear_bwd(const float* x, const float* y, const float* dy, const float* w, int32_t B, int32_t Cin, int32_t Cout,
                   int relu, float* dx, float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream);
int rsp_l2norm_fwd(const float* x, int32_t B, int32_t dim, float* y, void* stream);
int rsp_l2norm_bwd(const float* x, const float* dy, int32_t B, int32_t dim, float* dx, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Contrastive block (moco/builder_diffspeed_diffloss.py:521-538) and Loss (:263-283).
 * logits1 = [qA.kA | qA@queue]/T, logits2 = [qA.knegA | qA@queue]/T  each [B][1+K];  lposM = qM.kM/T, lnegM = qM.knegM/T
 * queue: [dim][K] (register_buffer "queue", :329-330).  qA/kA/knegA are [B][dim]; qM/kM/knegM are [B][dim_m]
 * (dim_m == dim except for fc_type 'speednet', whose second head is 1-dimensional: moco/split_wrapper.py:124-126).
 * ------------------------------------------------------------------------------------------------------- */
int rsp_logits_fwd(const float* qA, const float* qM, const float* kA, const float* kM, const float* knegA,
                   const float* knegM, const float* queue, int32_t B, int32_t dim, int32_t dim_m, int32_t K, float inv_T,
                   float* logits1, float* logits2, float* lposM, float* lnegM, void* stream);
size_t rsp_logits_bwd_workspace(int32_t B, int32_t dim, int32_t K);
int rsp_logits_bwd(const float* dlogits1, const float* dlogits2, const float* dlposM, const float* dlnegM,
                   const float* kA, const float* kM, const float* knegA, const float* knegM, const float* queue,
                   int32_t B, int32_t dim, int32_t dim_m, int32_t K, float inv_T, float* dqA, float* dqM, void* workspace,
                   size_t workspace_bytes, void* stream);

/* losses[3] = (A*(ce1+ce2)+M*ranking, ce1+ce2, ranking); CE targets are class 0 (labels_A), ranking target +1.
 * Also writes the gradients for d(losses[0]) = 1: dlogits1/2 [B][1+K], dlpos/dlneg [B]. */
int rsp_loss_fwd_bwd(const float* logits1, const float* logits2, const float* lposM, const float* lnegM, int32_t B,
                     int32_t K1 /* = 1+K */, float margin, float A, float M, float* losses, float* dlogits1,
                     float* dlogits2, float* dlposM, float* dlnegM, float* row_scratch /*[2*B]*/, void* stream);

/* queue[:, ptr:ptr+n] = keys.T  (_dequeue_and_enqueue, :345-359); keys [n][dim]. */
int rsp_queue_enqueue(float* queue, int32_t dim, int32_t K, int32_t ptr, const float* keys, int32_t n, void* stream);
/* The same with the pointer in device memory: reads *ptr_dev (the module's int64 queue_ptr buffer, :332), writes the slab and
 * advances *ptr_dev by n modulo K (:356-359).  No host-side pointer: a step captured in a HIP graph replays correctly. */
int rsp_queue_enqueue_dev(float* queue, int32_t dim, int32_t K, int64_t* ptr_dev, const float* keys, int32_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Step glue
 * ------------------------------------------------------------------------------------------------------- */
/* _diff_speed frame gather (:421-443) fused with the shuffle-BN sample permutation (:384-387) and the
 * NCDHW -> NDHWC layout change:  out[j] = im[src[j]][:, frames(step[j])], frames(s) = 0, s, 2s, ... (T_out of them).
 * im: (B_in, C, T_in, H, W) NCDHW as the reference's data loader hands it over; out: [B_out][T_out][H][W][C_out],
 * C_out >= C, extra channels zero (C_out = 4 lets a 3-channel stem convolution use 16-byte gathers). */
int rsp_clip_gather(const float* im, int32_t B_in, int32_t C, int32_t T_in, int32_t H, int32_t W, const int32_t* src,
                    const int32_t* step, int32_t B_out, int32_t T_out, int32_t C_out, float* out, void* stream);
/* The step's gathers in ONE launch: n_jobs (<= 4) gathers of identical geometry — the k_negative clips, the k clips (both in the
 * send order of their shuffle-BN exchange, :361-387) and the query clips (:421-447) — each with its own source batch, index and
 * speed vectors and output.  A launch that moves 0.5 GB streams at HBM rate; three launches of 0.18 GB each spend a fifth of their
 * 48 us on ramp-up and tail. */
int rsp_clip_gather_multi(int32_t n_jobs, const float* const* ims, const int32_t* const* srcs, const int32_t* const* steps,
                          float* const* outs, int32_t B_in, int32_t C, int32_t T_in, int32_t H, int32_t W, int32_t B_out, int32_t T_out,
                          int32_t C_out, void* stream);

/* _momentum_update_key_encoder (:337-343) on flat parameter buffers: k = k*m + q*(1-m). */
int rsp_momentum_update(float* k, const float* q, int64_t n, float m, void* stream);

/* torch.optim.SGD step (pretrain.py:65-72,165) on flat buffers: d = g*gscale + wd*p; buf = first ? d : mu*buf + d;
 * p -= lr*buf. */
int rsp_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float mu, float wd, float gscale,
                 int first, void* stream);

/* Small element-wise ops on dense fp32 vectors (y may alias a or b):
 *   RSP_ELT_RELU_FWD     y = max(a, 0)              nn.ReLU between the two convs of ConvFc (moco/split_wrapper.py:31-34)
 *   RSP_ELT_RELU_BWD     y = a > 0 ? b : 0          its backward (a = forward output, b = incoming gradient)
 *   RSP_ELT_SIGMOID_FWD  y = 1 / (1 + exp(-a))      torch.sigmoid of the 'speednet' head (moco/split_wrapper.py:146-147)
 *   RSP_ELT_SIGMOID_BWD  y = b * a * (1 - a)        its backward (a = forward output)
 *   RSP_ELT_ADD          y = a + b                  autograd's gradient accumulation where a tensor has several consumers
 *                                                   (residual adds models/resnet.py:72-75, inception fan-out models/s3dg.py:93-99) */
#define RSP_ELT_RELU_FWD 0
#define RSP_ELT_RELU_BWD 1
#define RSP_ELT_SIGMOID_FWD 2
#define RSP_ELT_SIGMOID_BWD 3
#define RSP_ELT_ADD 4
int rsp_eltwise(int32_t op, const float* a, const float* b /* nullable for the *_FWD ops */, float* y, int64_t n, void* stream);

/* rows gather: out[j][:] = in[idx[j]][:]  (feature un-shuffle, :389-406). */
int rsp_rows_gather(const float* in, const int32_t* idx, int32_t n, int32_t width, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Per-clip GPU augmentation of the pretext data path, fused into one pass:
 *   ToTensorVideo -> Resize(size, bilinear, align_corners=False) -> RandomGrayScale -> ColorJitter (brightness, contrast,
 *   saturation, hue in a per-clip random order) -> RandomHorizontalFlipVideo -> NormalizeVideo
 * and the `moco.aug_plus` variant  ... -> RandomApply(ColorJitter) -> RandomGrayScale -> RandomApply(GaussianBlur 3x3) -> ...
 * (datasets/classification/__init__.py:189-218; transforms_spatial.py:16-25; transforms_tensor.py:12-34,52-143;
 *  functional_tensor.py:89-162,254-417; applied clip by clip in SequentialGPUCollateFn, transforms_tensor.py:207-233).
 * The random draws stay on the host (Python's `random`, same order as the reference); each clip is described by one
 * rsp_augment_clip_desc in DEVICE memory.  src is the uint8 (T,h,w,3) crop produced by RawVideoRandomCrop
 * (transforms_spatial.py:28-80), addressed through pitches so it may also point into a larger decoded frame.
 * out: float32 (n_clips, 3, T, size, size) -- the model's NCDHW input -- with `out_clip_stride` floats between clips.
 * ------------------------------------------------------------------------------------------------------- */
#define RSP_AUG_BRIGHTNESS 0
#define RSP_AUG_CONTRAST 1
#define RSP_AUG_SATURATION 2
#define RSP_AUG_HUE 3
typedef struct rsp_augment_clip_desc {
  const uint8_t* src;
  int64_t frame_pitch;   /* bytes between frames */
  int32_t row_pitch;     /* bytes between rows (pixels are 3 packed bytes) */
  int32_t h, w;          /* region height / width */
  int32_t gray, flip;    /* gray: 0 no, 1 RandomGrayScale hit BEFORE the colour ops (default chain), 2 AFTER them (aug_plus chain),
                            +4 GaussianBlur hit (aug_plus: 3x3 zero-padded blur of the intermediate image, before the flip);
                            flip: RandomHorizontalFlipVideo hit */
  int32_t n_ops;         /* 0..4 colour ops, applied in this order */
  int32_t op[4];         /* RSP_AUG_* */
  float factor[4];       /* (float)ratio, or the hue shift */
  float one_minus[4];    /* (float)(1.0 - ratio), evaluated in double as Python does (functional_tensor.py:103-106) */
} rsp_augment_clip_desc;
size_t rsp_augment_workspace(int32_t n_clips, int32_t T, int32_t size);
/* mean3 / std3 / blur9: HOST pointers to 3 / 3 / 9 floats (read during the call); blur9 = row-major 3x3 kernel of
 * transforms_tensor.py:GaussianBlur (:146-204), may be NULL when no descriptor has the blur bit. */
int rsp_augment_batch(const rsp_augment_clip_desc* descs, int32_t n_clips, int32_t T, int32_t size, const float* mean3,
                      const float* std3, const float* blur9, float* out, int64_t out_clip_stride, void* workspace,
                      size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RSPNET_HIP_H */
