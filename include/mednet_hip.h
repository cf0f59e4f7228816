/* mednet_hip.h -- C ABI of libmednet_hip.so: the MI355X (gfx950) 3D U-Net training hot path.
 *
 * The reference (tobiashepp/torch-mednet) has no FFI of its own: its hot path is torch.nn modules calling
 * ATen.  Each entry point below therefore names the reference call site whose ATen op it replaces
 * (paths relative to the reference root).  The Python host (mednet_hip/_lib.py, ctypes) is the only caller.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator); nothing is allocated,
 *     freed or synchronised inside; launches go on `stream` (a hipStream_t passed as void*).
 *   - activations are NDHWC ("channels_last_3d") unless a layout argument says otherwise; dtype arguments name the STORAGE
 *     type of a tensor: MEDNET_F32, MEDNET_BF16 or MEDNET_F16 for activations and their gradients (arithmetic is fp32
 *     accumulation in every mode), MEDNET_U8 / MEDNET_I64 for labels and MEDNET_U8 / MEDNET_F16 for resident volumes where
 *     an entry point says so; parameters and their gradients are always fp32 in PyTorch layout.
 *   - no global state: everything a launch depends on is an argument.  The one exception is mednet_set_option, a table of
 *     integer A/B knobs for measurements; it never carries pointers, results never depend on it, and the product path
 *     (the mednet_hip Python package) sets none.
 *   - return value: MEDNET_OK or a negative MEDNET_E_*; mednet_last_error() gives the text.  The Python shim
 *     raises RuntimeError, matching the reference's assert/exception convention (components.py:30-31,56,65;
 *     loss.py:28,66).
 *   - workspace: *_ws_bytes() gives the scratch size a call needs; the caller passes a buffer at least that big.
 *     Workspaces hold deterministic per-workgroup partials (no float atomics anywhere => bitwise reproducible).
 */
#ifndef MEDNET_HIP_H
#define MEDNET_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* mednet_stream; /* hipStream_t */

enum { MEDNET_F32 = 0, MEDNET_BF16 = 1,
       MEDNET_F16 = 2, /* third activation storage type (fp16, with loss scaling: train.LossScaler); resident image volumes */
       MEDNET_U8 = 3,  /* label / heat-map volumes (mednet_crop_patches), labels of the fused head + Dice kernels */
       MEDNET_I64 = 4  /* labels as the reference's callers pass them (`.long()`, segmentation.py:60) */ };
enum { MEDNET_NDHWC = 0, MEDNET_NCDHW = 1 };
enum { MEDNET_ACT_NONE = 0, MEDNET_ACT_RELU = 1, MEDNET_ACT_LEAKY = 2, MEDNET_ACT_ELU = 3 };
enum { MEDNET_POOL_MAX = 0, MEDNET_POOL_AVG = 1 };
/* ALGO_EXACT = ALGO_AUTO, except that fp32-storage contractions use exact fp32 products (v_mfma_f32_32x32x2_f32) instead of
 * the split-bf16 contraction (three bf16 MFMAs per product, ~2^-16 relative): asked for by networks with ReLU / LeakyReLU,
 * whose gradients are discontinuous in the pre-activations (mednet_hip/config.py exact_products). */
enum { MEDNET_ALGO_AUTO = 0, MEDNET_ALGO_DIRECT = 1, MEDNET_ALGO_MFMA = 2, MEDNET_ALGO_EXACT = 3,
       /* the same request OR-ed onto a base choice: MEDNET_ALGO_MFMA | MEDNET_ALGO_EXACT_BIT = matrix-core path required AND
        * exact fp32 products (ALGO_MFMA alone takes the split-bf16 contraction in the fp32 storage mode) */
       MEDNET_ALGO_EXACT_BIT = 4,
       /* (ABI 3, round 6) SPLIT WEIGHTS, OR-ed onto any base choice; 16-bit storage modes only (ignored for fp32 storage).  The
        * matrix cores read 16-bit weight images: in fp16 storage that 11-bit rounding of the WEIGHTS -- not the 16-bit storage of
        * activations and gradients -- is what keeps the gradients of the reference's fp32 step outside 1e-3 (tools/attrib16.py,
        * profiles/r06_ab.md section 2).  With this bit every 3x3x3 forward / data-gradient convolution, ConvTranspose3d forward /
        * data gradient and the first layer also multiply the LOW image elt(w - elt(w)) of the layer's weights (two MFMAs per
        * product; the weight gradients have no weight operand and are unchanged).  The pack must hold the low images: pack with
        * elt_dtype | MEDNET_PACK_LOW (every bf16 pack holds them anyway). */
       MEDNET_ALGO_SPLITW_BIT = 8 };
/* flag of the elt_dtype argument of mednet_conv3d_pack_elt / mednet_conv3d_pack_many: an fp16 pack with the low images as well */
enum { MEDNET_PACK_LOW = 0x100,
       /* (round 6) mednet_conv3d_pack_many only: rewrite ONLY what the 16-bit matrix-core kernels of the training step read -- the
        * high 16-bit images (and the low ones iff MEDNET_PACK_LOW is set) -- and leave the fp32 images of the direct / fp32-matrix
        * kernels and the unrequested low images as they are, i.e. STALE after an optimizer step (2.8 GB per step for a 141 M
        * parameter network against 1.1 GB).  The caller then owns the bookkeeping: before a call that takes another path with such a
        * pack (fp32 tensors, MEDNET_ALGO_DIRECT / exact products, a sample of 4 GB and more, a ConvTranspose3d whose channels are not
        * multiples of 32) it must re-pack that layer in full (mednet_conv3d_pack_elt).  mednet_hip.nn does (`_packed(x)`). */
       MEDNET_PACK_HIGH_ONLY = 0x200 };
enum { MEDNET_REG_L2 = 0, MEDNET_REG_L1 = 1 };
enum {
  MEDNET_OK = 0, MEDNET_E_SHAPE = -1, MEDNET_E_DTYPE = -2, MEDNET_E_WORKSPACE = -3, MEDNET_E_HIP = -4,
  MEDNET_E_UNSUPPORTED = -5
};
#define MEDNET_NO_IGNORE INT32_MIN

int mednet_abi_version(void);
const char* mednet_last_error(void);
/* 1 if a gfx950 device is visible to this process, 0 otherwise (never throws). */
int mednet_device_ok(void);
/* Kernel-variant knobs for in-process A/B measurements (e.g. "conv32" 0|1); integers only, results never depend on them,
 * and the table is read without a lock: set them from one thread while no launch is being planned (tools/, tests/). */
int mednet_set_option(const char* name, int value);
/* the value a launcher would read now (default_value when the option was never set); tests and tools */
int mednet_get_option(const char* name, int default_value);

/* ---- nn.Conv3d(k=3,p=1 | k=1,p=0, stride 1)  components.py:8-9,44 ; model.py:77,179 ------------------------ */
/* Weight packing: PyTorch (Cout,Cin,k,k,k) [or ConvTranspose3d's (Cin,Cout,k,k,k) when transposed_src=1] ->
 * opaque buffer holding the tap-major layouts the forward, data-gradient and MFMA kernels read. */
size_t mednet_conv3d_pack_bytes(int cin, int cout, int ksize);  /* the buffer must be 16-byte aligned (hipMalloc / torch allocations are) */
/* mednet_conv3d_pack writes the matrix-core fragment images in bf16; _elt takes the storage type of the activations the
 * layer will see.  MEDNET_BF16 and MEDNET_F32 write the same thing: bf16 images of the HIGH halves bf16(w) plus images of
 * the LOW halves bf16(w - bf16(w)); 16-bit calls read the high images, fp32-storage calls (the mode that meets the reference
 * within 1e-3) contract against both (split-bf16 product, conv_x3_mfma.hip) -- so a buffer packed either way serves both
 * storage modes.  MEDNET_F16 (BASELINE config 5 stores fp16) writes fp16 high images and serves fp16-storage calls ONLY: the
 * exact-product fp32 kernels (MEDNET_ALGO_EXACT) still work on it, they read the fp32 tap-major images every pack holds, but
 * an automatic-algorithm fp32-storage call on an fp16 pack is a caller error.  The buffer size is the same in every case. */
int mednet_conv3d_pack_elt(const float* w, void* packed, int cin, int cout, int ksize, int transposed_src, int elt_dtype,
                           mednet_stream stream);
int mednet_conv3d_pack(const float* w, void* packed, int cin, int cout, int ksize, int transposed_src,
                       mednet_stream stream);
/* Every 3x3x3 layer with matrix-core images in ONE launch (after an optimizer step all weights have moved; the reference
 * has no counterpart: torch.nn keeps no packed copies).  mednet_conv3d_pack_table turns the caller's job list into the
 * table the kernel reads (table_host: mednet_conv3d_pack_table_bytes(njobs) bytes of host memory, to be copied to the
 * device once; *blocks = the block count of the launch, to be handed back to mednet_conv3d_pack_many);
 * mednet_conv3d_pack_many repeats what mednet_conv3d_pack_elt does for each job. */
typedef struct mednet_pack_job {
  const float* w;      /* parameter, PyTorch layout, device */
  void* packed;        /* its pack buffer (mednet_conv3d_pack_bytes), device */
  int cin, cout, ksize, transposed_src;
} mednet_pack_job;
size_t mednet_conv3d_pack_table_bytes(int njobs);
int mednet_conv3d_pack_table(const mednet_pack_job* jobs, int njobs, void* table_host, unsigned* blocks);
int mednet_conv3d_pack_many(const void* table_device, int njobs, unsigned blocks, int elt_dtype, mednet_stream stream);
/* y[n,z,y,x,co] = bias[co] + sum_{tap,ci} x[n,z+dz-1,y+dy-1,x+dx-1,ci] * W[co,ci,tap].
 * dgrad=1 runs the data gradient with the same kernel: pass x := dy, cin := Cout, cout := Cin of the layer. */
/* gn_partial (nullable): when the call takes the MFMA path the epilogue also writes the GroupNorm partial sums of the
 * output, [n][chunks][cout][2] = {sum y, sum y^2} per brick, chunks = mednet_conv3d_fused_stats_chunks(...) (0 = this
 * call cannot fuse them); mednet_gn_finalize turns them into statistics without another pass over y.  The sums are
 * kept per channel PAIR (entry 2j = channels 2j and 2j+1 together, entry 2j+1 = 0): exact for GroupNorm whenever the
 * channels per group are even -- ask for them only then. */
/* Audit aid (no device needed): the launch plan of the 16-bit matrix-core kernel that produces a 3x3x3 layer's partial rows,
 * from the launcher's own planning code.  gnb: the data-gradient form with GroupNorm-backward sums (pass the kernel's Cin /
 * Cout, i.e. the layer's Cout / Cin); stride 2: the ConvTranspose3d data gradient with GroupNorm-3 sums (d, h, w = the
 * low-resolution grid).  out13 = {kind (2: one row per wave and brick, row = 4 * brick-in-sample + wave; 3: accumulate mode,
 * a wave keeps its sums over its workgroup's items and writes row ((wg >> 3) / ncb * 8 + (wg & 7)) * 4 + wave for EVERY
 * sample; 4: the 32 -> 32 specialisation, row = 4 * wg + wave for every sample; 5 / 6: the two-block kernel with the rows of
 * 2 / 3 and "channel blocks" counting PAIRS of 32-channel blocks; 7: the ConvTranspose3d 64 -> 32 data gradient, row = 2 * wg +
 * z-plane of its bricks for every sample, both channel blocks), grid, work items, channel blocks, bricks,
 * bricks per sample, accumulate flag, rows per sample, and for kind 4 bricks per XCD, z-slab height, brick counts in x, y, z}.
 * Work item i of the general kernel = (brick (i >> 3) / ncb * 8 + (i & 7), channel block (i >> 3) % ncb); workgroup b takes
 * items b, b + grid, ... and stops at the first brick >= bricks (padding items). */
int mednet_conv3d_stats_plan(int n, int d, int h, int w, int cin, int cout, int dtype, int gnb, int stride, int* out13);
int mednet_conv3d_fused_stats_chunks(int n, int d, int h, int w, int cin, int cout, int ksize, int x_dtype,
                                     int y_dtype, int algo);
int mednet_conv3d_fwd(const void* x, const void* packed, const float* bias, void* y, int n, int d, int h, int w,
                      int cin, int cout, int ksize, int x_dtype, int x_layout, int y_dtype, int y_layout,
                      int dgrad, int algo, float* gn_partial, mednet_stream stream);
/* conv 3x3x3 without bias + activation (MEDNET_ACT_*) in the epilogue, bf16 NDHWC in and out, matrix-core path only:
 * the conv -> ReLU / LeakyReLU / ELU step of the 'gcr'-style orders (components.py:12-67; UNet3D's default).  gn_partial as
 * in mednet_conv3d_fwd; the statistics are those of the ACTIVATED output (the next GroupNorm's input).  Ask
 * mednet_conv3d_act_supported first; unsupported shapes return MEDNET_E_UNSUPPORTED (use conv3d_fwd + act_fwd). */
int mednet_conv3d_act_supported(int n, int d, int h, int w, int cin, int cout, int algo);
int mednet_conv3d_act_fwd(const void* x, const void* packed, void* y, int n, int d, int h, int w, int cin, int cout,
                          int act, int algo, float* gn_partial, int dtype, mednet_stream stream);
/* dx = dgrad(dy) + add: the data gradient of a 3x3x3 conv (layer Cin -> Cout; dy has Cout channels, dx and add have Cin)
 * with a second gradient of the same tensor summed in the epilogue (fp32 add, one bf16 rounding): in ExtResNetBlock the
 * first conv's output feeds conv2 AND the residual add (components.py:170-178), so its two gradients meet here instead
 * of in two more tensor reads of the GroupNorm backward.  bf16 NDHWC, matrix-core path only
 * (mednet_conv3d_act_supported(n,d,h,w,Cout,Cin,algo) tells). */
int mednet_conv3d_dgrad_add(const void* dy, const void* packed, const void* add, void* dx, int n, int d, int h, int w,
                            int cin, int cout, int algo, int dtype, mednet_stream stream);
/* mednet_conv3d_dgrad_add (add nullable) that ALSO takes the first pass of the backward of the GroupNorm (+ activation)
 * in front of the layer (components.py:57,36-40: `SingleConv` k-1 of the block produced this conv's input): with gn_y the
 * conv output that GroupNorm normalised (shape of dx), gn_coef[n][Cin] = {ca, cb} its forward affine (mednet_gn_stats /
 * _finalize) and gn_act its activation, the epilogue forms du = dx * act'(ca * gn_y + cb) from the stored dx rows and writes
 * gn_partial[n][rows][Cin][2] = per-channel {sum du, sum du * gn_y} (rows = mednet_conv3d_dgrad_gn_rows(...), 0 = not
 * supported for this shape).  mednet_gn_act_bwd_fused consumes it: no stand-alone pass re-reads dx and gn_y. */
int mednet_conv3d_dgrad_gn_rows(int n, int d, int h, int w, int cin, int cout, int algo);
/* ... by storage type: MEDNET_F32 = the fp32 storage mode, where the split-bf16 data-gradient kernel takes the sums (and the
 * summed second gradient) the same way; 16-bit types answer as mednet_conv3d_dgrad_gn_rows.  mednet_conv3d_dgrad_add_supported:
 * 1 if mednet_conv3d_dgrad_add takes this layer in this storage type. */
int mednet_conv3d_dgrad_gn_rows_dt(int n, int d, int h, int w, int cin, int cout, int algo, int dtype);
int mednet_conv3d_dgrad_add_supported(int n, int d, int h, int w, int cin, int cout, int algo, int dtype);
int mednet_conv3d_dgrad_gn(const void* dy, const void* packed, const void* add, void* dx, const void* gn_y,
                           const float* gn_coef, int gn_act, float* gn_partial, int n, int d, int h, int w, int cin,
                           int cout, int algo, int dtype, mednet_stream stream);
/* GroupNorm-3 of an ExtResNetBlock (out = act(GroupNorm(conv3(z2)) + z1), components.py:170-178): its backward's first
 * pass -- du = dout * act'(out), per-channel {sum du, sum du * y3} -- taken by the kernel that PRODUCES dout, from the
 * stored (rounded) dout rows: the data gradient of the 1x1x1 head (model.py:204-207; dy = planar fp32 logit gradients,
 * dx / gn_y = y3 / gn_z = out channels-last 16-bit, cin = block channels, cout = classes) for the last decoder block, and
 * the pooling backward + skip-gradient join (mednet_pool2_bwd; model.py:194-205; x = the block output) for the encoder
 * blocks that feed a pooling.  gn_partial[n][rows][C][2] with rows from the *_rows call (0 = not supported for this shape / dtype; even
 * d, h, w for the pooling form); mednet_gn_act_bwd_fused_res consumes it and also writes the residual-branch gradient. */
int mednet_head_dgrad_gn_rows(int n, int d, int h, int w, int cin, int dtype);
int mednet_head_dgrad_gn(const void* dy, const void* packed, void* dx, const void* gn_y, const void* gn_z, int gn_act,
                         float* gn_partial, int n, int d, int h, int w, int cin, int cout, int dtype, mednet_stream stream);
/* ... and the ConvTranspose3d data gradient (mednet_convt3d_dgrad: model.py:202-207, the decoder's upsampling consumes the
 * output of the block below; (n,d,h,w) = dx dims, gn_y / gn_z = that block's y3 / output, Cin channels), matrix-core path. */
int mednet_convt3d_dgrad_gn_rows(int n, int d, int h, int w, int cin, int cout, int dtype, int algo);
int mednet_convt3d_dgrad_gn(const void* dy, const void* packed, void* dx, const void* gn_y, const void* gn_z, int gn_act,
                            float* gn_partial, int n, int d, int h, int w, int cin, int cout, int dtype, int algo,
                            mednet_stream stream);
int mednet_pool2_bwd_gn_rows(int n, int d, int h, int w, int c, int dtype);
int mednet_pool2_bwd_gn(const void* dy, const void* x, const void* add, void* dx, const void* gn_y, int gn_act,
                        float* gn_partial, int n, int d, int h, int w, int c, int mode, int dtype, mednet_stream stream);
/* dw[co,ci,tap] = sum_{n,v} dy[n,v,co] * x[n,v+tap,ci]; dbias[co] = sum dy (nullable).
 * `workgroups` (>= 0) is part of the launch plan and therefore an ARGUMENT of the call and of its workspace query (the same
 * value to both): 0 = the library's plan, one persistent workgroup per CU; N > 0 = about N workgroups -- the trainer passes
 * half the CUs for launches that run on a second stream beside the main one (a weight-gradient workgroup takes a CU's whole
 * register file; mednet_hip/ops.py _OnSide).  Results are deterministic for a fixed value (fixed-order reduce of the
 * per-workgroup partial slabs); another value changes the rounding of the sums, once.  Kernels that do not split their work
 * this way (first layer, 1x1x1 head, direct kernels) ignore it. */
size_t mednet_conv3d_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout, int ksize, int workgroups);
/* 1 if this weight gradient runs on the co-resident kernel (wgrad_mfma4_kernel: 4 waves of < 256 registers, 104 KB of LDS --
 * bandwidth-bound kernels of another stream run on the same CUs beside it); 0 if it takes its CUs whole.  A caller that runs
 * weight gradients on a second stream passes workgroups = 0 (all CUs) in the first case and about half the CUs in the second. */
int mednet_conv3d_wgrad_coresident(int n, int d, int h, int w, int cin, int cout, int ksize, int x_dtype, int dy_dtype, int algo);
/* Launch plan of the 3x3x3 weight gradient (16-bit matrix-core path) made by the launcher's own planning code, for audits without
 * a device (tests/test_plan_audit.py): out10 = {kind (4: wgrad_mfma4_kernel, z-columns; 2: wgrad_mfma2_kernel, bricks), workgroups,
 * channel-block pairs, workgroups per pair, work items per pair, tiles_x, tiles_y, z-slabs (kind 4) or tiles_z (kind 2), planes
 * per slab (kind 4) or brick width (kind 2), XCD remap of the item order (kind 4)}. */
int mednet_conv3d_wgrad_plan(int n, int d, int h, int w, int cin, int cout, int dtype, int workgroups, int* out10);
/* The Cin = 1, 3x3x3 weight gradient with GroupNorm's backward apply folded into its staging: dz / y = gradient of, and input
 * to, act(GroupNorm(y)) (16-bit NDHWC, Cout channels), coef = the forward affine {ca, cb}, bcoef from
 * mednet_gn_bwd_coefficients.  dw gets what mednet_conv3d_wgrad(x, dy) returns for the dy mednet_gn_act_bwd_fused stores --
 * bit for bit -- without dy ever reaching memory.  Workspace: mednet_conv3d_wgrad_ws_bytes(n, d, h, w, 1, cout, 3, 0).
 * _supported: Cout in {16, 32, 64}, 16-bit dtype, x fp32 or that dtype. */
int mednet_conv3d_wgrad_c1_gn_supported(int cout, int x_dtype, int dtype);
int mednet_conv3d_wgrad_c1_gn(const void* x, const void* dz, const void* y, const float* coef, const float* bcoef, float* dw,
                              int n, int d, int h, int w, int cout, int act, int x_dtype, int dtype, void* ws, size_t ws_bytes,
                              mednet_stream stream);
int mednet_conv3d_wgrad(const void* x, const void* dy, float* dw, float* dbias, int n, int d, int h, int w,
                        int cin, int cout, int ksize, int x_dtype, int x_layout, int dy_dtype, int dy_layout,
                        int algo, int workgroups, void* ws, size_t ws_bytes, mednet_stream stream);

/* ---- nn.ConvTranspose3d(k=3,s=2,p=1,output_padding=1,bias) + `x += encoder_features`  components.py:259-264,283-284 */
/* (n,d,h,w) are the INPUT dims; output is (2d,2h,2w).  `skip` (nullable, y's dtype/shape) is added in the epilogue. */
int mednet_convt3d_fwd(const void* x, const void* packed, const float* bias, const void* skip, void* y, int n,
                       int d, int h, int w, int cin, int cout, int x_dtype, int y_dtype, int algo,
                       mednet_stream stream);
int mednet_convt3d_dgrad(const void* dy, const void* packed, void* dx, int n, int d, int h, int w, int cin,
                         int cout, int dy_dtype, int dx_dtype, int algo, mednet_stream stream);
/* (`workgroups`: as for mednet_conv3d_wgrad) */
size_t mednet_convt3d_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout, int workgroups);
int mednet_convt3d_wgrad(const void* x, const void* dy, float* dw, float* dbias, int n, int d, int h, int w,
                         int cin, int cout, int x_dtype, int dy_dtype, int algo, int workgroups, void* ws,
                         size_t ws_bytes, mednet_stream stream);

/* ---- nn.GroupNorm(G,C,eps) fused with the following activation and the residual add
 *      components.py:57 (GroupNorm), :36-40 (ReLU/LeakyReLU(0.1)/ELU), :177-178 (out += residual; non_linearity) */
size_t mednet_gn_ws_bytes(int n, int c, size_t spatial);
/* stats[n][g] = {mean, rstd}; coef[n][c] = {gamma*rstd, beta - mean*gamma*rstd} (both outputs, fp32). */
int mednet_gn_stats(const void* x, const float* gamma, const float* beta, float* stats, float* coef, int n,
                    size_t spatial, int c, int groups, float eps, int dtype, void* ws, size_t ws_bytes,
                    mednet_stream stream);
/* the second half of mednet_gn_stats for partial sums produced elsewhere (fused conv epilogue):
 * partial is [n][chunks][c][2]; ws needs n*c*2 floats. */
int mednet_gn_finalize(const float* partial, int chunks, const float* gamma, const float* beta, float* stats,
                       float* coef, int n, size_t spatial, int c, int groups, float eps, void* ws, size_t ws_bytes,
                       mednet_stream stream);
/* z = act(coef0*x + coef1 [+ residual]) */
int mednet_gn_act_fwd(const void* x, const float* coef, const void* residual, void* z, int n, size_t spatial,
                      int c, int act, int x_dtype, int z_dtype, mednet_stream stream);
/* du = (dz [+ dz2]) * act';  dgamma/dbeta;  dx = GroupNorm backward of du;  dres (nullable) := du.
 * act' comes from the activated output z when given (ATen's in-place semantics), else -- no residual branch -- it is
 * recomputed from x and the forward coefficients `coef`, which saves reading z back (one tensor less per pass). */
/* in_act (both forms below): MEDNET_ACT_* of the activation whose OUTPUT is x (conv -> ReLU -> this GroupNorm in the 'gcr'
 * orders, components.py:36-40): dx is multiplied by its derivative here, sparing the conv layer's backward a pass. */
int mednet_gn_act_bwd(const void* dz, const void* dz2, const void* x, const void* z, const float* coef,
                      const float* stats, const float* gamma, void* dx, void* dres, float* dgamma, float* dbeta, int n,
                      size_t spatial, int c, int groups, int act, int in_act, int dtype, void* ws, size_t ws_bytes,
                      mednet_stream stream);
/* mednet_gn_act_bwd without its first pass: `partial` [n][rows][c][2] = per-channel {sum du, sum du * x} comes from the
 * kernel that produced dz (mednet_conv3d_dgrad_gn); act' is recomputed from x and `coef` (z is not read). */
int mednet_gn_act_bwd_fused(const void* dz, const void* x, const float* coef, const float* stats, const float* gamma,
                            const float* partial, int rows, void* dx, float* dgamma, float* dbeta, int n,
                            size_t spatial, int c, int groups, int act, int in_act, int dtype, void* ws, size_t ws_bytes,
                            mednet_stream stream);
/* Passes 1b + 2 of mednet_gn_act_bwd_fused alone: bcoef[n][c][3] = {k1, k2, k3} with dx = k1 * du + k2 * x + k3
 * (du = dz * act'(coef_a * x + coef_b)), and dgamma / dbeta.  For a consumer that applies the coefficients while it reads
 * (dz, x) for its own purpose -- mednet_conv3d_wgrad_c1_gn below -- so the apply pass and its dx tensor disappear.  The first
 * SingleConv of the network (components.py:57 after model.py:63's first encoder) needs no gradient of the network input:
 * its GroupNorm backward only feeds the weight gradient. */
int mednet_gn_bwd_coefficients(const float* stats, const float* gamma, const float* partial, int rows, float* bcoef,
                               float* dgamma, float* dbeta, int n, size_t spatial, int c, int groups, void* ws,
                               size_t ws_bytes, mednet_stream stream);
int mednet_gn_act_bwd_fused_res(const void* dz, const void* x, const void* z, const float* coef, const float* stats,
                                const float* gamma, const float* fused_partial, int rows, void* dx, void* dres,
                                float* dgamma, float* dbeta, int n, size_t spatial, int c, int groups, int act, int dtype,
                                void* ws, size_t ws_bytes, mednet_stream stream);
/* ... when dz is the pooling backward + skip join of an encoder level and was NOT materialised: mednet_pool2_bwd_gn called with
 * dx = NULL takes the sums only, and this apply pass rebuilds dz per 2x2x2 window from the pooled gradient dy_pool, the arg-max of
 * the block output z (read for act' anyway) and the skip gradient (nullable), rounded as the pooling backward would have stored
 * it.  dx (= dy3) and dres are bit-identical to mednet_pool2_bwd_gn + mednet_gn_act_bwd_fused_res.  Even d, h, w; C % 8 == 0. */
int mednet_gn_act_bwd_fused_res_pool(const void* dy_pool, const void* skip_grad, const void* x, const void* z, const float* stats,
                                     const float* gamma, const float* fused_partial, int rows, void* dx, void* dres, float* dgamma,
                                     float* dbeta, int n, int d, int h, int w, int c, int groups, int act, int pool_mode, int dtype,
                                     void* ws, size_t ws_bytes, mednet_stream stream);
/* stand-alone activation (orders such as 'cr', 'crg'); in-place allowed (x == z). */
int mednet_act_fwd(const void* x, void* z, size_t count, int act, int dtype, mednet_stream stream);
int mednet_act_bwd(const void* dz, const void* z, void* dx, size_t count, int act, int dtype,
                   mednet_stream stream);
/* out = a + b (gradient joins the graph needs: residual + skip fan-in). */
int mednet_add(const void* a, const void* b, void* out, size_t count, int dtype, mednet_stream stream);

/* ---- nn.MaxPool3d(2) / nn.AvgPool3d(2)  components.py:208-212 ----------------------------------------------- */
int mednet_pool2_fwd(const void* x, void* y, int n, int d, int h, int w, int c, int mode, int dtype,
                     mednet_stream stream);
/* dx has the INPUT shape; ties route to the first maximum in (z,y,x) scan order like ATen.  `add` (nullable, input
 * shape): a second gradient of the same tensor -- the decoder's skip join of model.py:199-205 -- summed in the same pass. */
/* GroupNorm apply (+ residual + activation, as mednet_gn_act_fwd) fused with the 2x2x2 pooling that consumes its output (an
 * encoder block's last layer followed by the next level's pooling, components.py:177-178 -> :222-224): writes z AND the pooled
 * tensor in one pass; both bit-identical to mednet_gn_act_fwd + mednet_pool2_fwd.  Even d, h, w; C a multiple of 8. */
int mednet_gn_act_pool_supported(int d, int h, int w, int c, int dtype);
int mednet_gn_act_pool_fwd(const void* x, const float* coef, const void* residual, void* z, void* pooled, int n, int d, int h,
                           int w, int c, int act, int mode, int dtype, mednet_stream stream);
int mednet_pool2_bwd(const void* dy, const void* x, const void* add, void* dx, int n, int d, int h, int w, int c,
                     int mode, int dtype, mednet_stream stream);
/* ... with the derivative of the activation whose OUTPUT x is (a fused conv -> activation layer, components.py:57-63) folded into
 * dx = (pooling backward + add) * act'(x), rounded like the stand-alone join followed by mednet_act_bwd.  Even d, h, w.
 * add_channels (0 or c: dense): `add` points at the first of c channels inside voxel rows of add_channels channels -- the leading
 * slice of the gradient of UNet3D's concatenation (components.py:277-280), read where it lies. */
int mednet_pool2_bwd_act(const void* dy, const void* x, const void* add, int add_channels, void* dx, int n, int d, int h, int w, int c,
                         int mode, int in_act, int dtype, mednet_stream stream);

/* ---- F.interpolate(nearest, size=enc) + torch.cat((enc, x), 1)  components.py:277-280 (UNet3D decoder) ------- */
int mednet_upcat_fwd(const void* enc, const void* x, void* out, int n, int d, int h, int w, int c_enc, int xd,
                     int xh, int xw, int c_x, int dtype, mednet_stream stream);
/* mednet_upcat_fwd that also takes the GroupNorm partial sums of the tensor it writes (UNet3D's decoder opens with a GroupNorm over
 * the concatenation, components.py:46-57 after :277-280): partial[n][mednet_upcat_stats_chunks][c_enc + c_x][2] = {sum, sum of
 * squares}, to be finalised by mednet_gn_finalize.  chunks == 0: not for this shape (channel counts must be multiples of 8). */
int mednet_upcat_stats_chunks(int n, int d, int h, int w, int c_enc, int c_x, int dtype);
int mednet_upcat_fwd_stats(const void* enc, const void* x, void* out, float* partial, int n, int d, int h, int w, int c_enc, int xd,
                           int xh, int xw, int c_x, int dtype, mednet_stream stream);
/* (denc == NULL: only dx; the caller reads the encoder part of dout where it lies) */
int mednet_upcat_bwd(const void* dout, void* denc, void* dx, int n, int d, int h, int w, int c_enc, int xd, int xh,
                     int xw, int c_x, int dtype, mednet_stream stream);

/* ---- losses.  logits are fp32 with element strides (stride_n, stride_c) and unit voxel stride (NCDHW or a
 *      channel slice of it); labels are int64 N x spatial ------------------------------------------------------ */
size_t mednet_loss_ws_bytes(int n, int c, size_t spatial);
/* DiceLoss  loss.py:91-130 (+ :24-48, :58-88): softmax|sigmoid -> one-hot -> per-channel 2*w*I/clamp(D,eps) ->
 * mean(1-dice).  saved[c] = {I_c, D_c} for the backward; dice_out (nullable) gets the per-channel dice. */
int mednet_dice_fwd(const float* logits, const int64_t* labels, const float* weight, float* loss, float* saved,
                    float* dice_out, int n, int c, size_t spatial, int64_t stride_n, int64_t stride_c, float eps,
                    int sigmoid, int ignore_index, void* ws, size_t ws_bytes, mednet_stream stream);
/* dlogits has the layout of the logits view it belongs to (the same stride_n / stride_c; a contiguous N x C x spatial tensor
 * for contiguous logits): a channel slice of the network output gets its gradient written into the matching slice of a
 * full-size buffer (landmarks.py:71-72).  Scaled by *dloss (device scalar). */
int mednet_dice_bwd(const float* logits, const int64_t* labels, const float* weight, const float* saved,
                    const float* dloss, float* dlogits, int n, int c, size_t spatial, int64_t stride_n,
                    int64_t stride_c, float eps, int sigmoid, int ignore_index, mednet_stream stream);
/* The same with the labels as they lie: MEDNET_I64 or MEDNET_U8, element stride label_stride_n between samples (the last channel
 * of a uint8 N x C x D x H x W label volume: `batch['label'][:, -1]`, segmentation.py:60 / landmarks.py:70, without the cast). */
int mednet_dice_fwd_lt(const float* logits, const void* labels, int label_dtype, int64_t label_stride_n, const float* weight,
                       float* loss, float* saved, float* dice_out, int n, int c, size_t spatial, int64_t stride_n,
                       int64_t stride_c, float eps, int sigmoid, int ignore_index, void* ws, size_t ws_bytes, mednet_stream stream);
int mednet_dice_bwd_lt(const float* logits, const void* labels, int label_dtype, int64_t label_stride_n, const float* weight,
                       const float* saved, const float* dloss, float* dlogits, int n, int c, size_t spatial, int64_t stride_n,
                       int64_t stride_c, float eps, int sigmoid, int ignore_index, mednet_stream stream);
/* The 1x1x1 head (model.py:207 final_conv: nn.Conv3d(f_maps[0], out_channels, 1)) fused with DiceLoss (loss.py:114-130) --
 * what `outputs = self(inputs); loss = self.loss(outputs, labels)` (segmentation.py:61-62) runs between the last decoder block and
 * the scalar loss.  Forward: logits (N x C x spatial fp32, written once, for the caller) + loss + saved[c] = {I_c, D_c} in one pass
 * over the channels-last features z (N x spatial x Cin; f32 / bf16 / f16).  Backward: dz = W^T dlogits (Cin in {16, 32, 64},
 * C <= 4 classes), dW, dbias and -- when gn_y is given -- the first pass of the producing ExtResNetBlock's GroupNorm-3 backward
 * (gn_partial[n][mednet_head_dice_gn_rows][Cin][2], as mednet_head_dgrad_gn) in ONE pass; the logit gradient is never stored.
 * labels: MEDNET_I64 or MEDNET_U8, N x spatial with element stride label_stride_n between samples (the last channel of a
 * uint8 label volume is consumed where it lies).  packed: the head's pack buffer (mednet_conv3d_pack, ksize 1).
 * Logits, loss, dz and the GroupNorm sums are bit-identical to mednet_conv3d_fwd + mednet_dice_fwd / mednet_dice_bwd +
 * mednet_head_dgrad_gn; dW / dbias are summed in another fixed order than mednet_conv3d_wgrad's.
 * mednet_head_dice_bwd with gn_y == NULL and gn_act != MEDNET_ACT_NONE: z is the OUTPUT of a fused conv -> activation layer
 * (UNet3D's last block); dz is stored as (head data gradient) * act'(z), rounded like mednet_act_bwd applied to the stored form. */
int mednet_head_dice_supported(int cin, int cout, int dtype, int label_dtype);
size_t mednet_head_dice_ws_bytes(int n, size_t spatial, int cin, int cout);
int mednet_head_dice_gn_rows(int n, size_t spatial, int cin);
int mednet_head_dice_fwd(const void* z, const void* packed, const float* bias, const void* labels, int label_dtype,
                         int64_t label_stride_n, const float* weight, float* logits, float* loss, float* saved, int n,
                         size_t spatial, int cin, int cout, float eps, int sigmoid, int ignore_index, int z_dtype, void* ws,
                         size_t ws_bytes, mednet_stream stream);
int mednet_head_dice_bwd(const float* logits, const void* labels, int label_dtype, int64_t label_stride_n, const void* packed,
                         const float* weight, const float* saved, const float* dloss, void* dz, const void* gn_y, const void* z,
                         int gn_act, float* gn_partial, float* dw, float* dbias, int n, size_t spatial, int cin, int cout,
                         float eps, int sigmoid, int ignore_index, int z_dtype, void* ws, size_t ws_bytes, mednet_stream stream);

/* The landmark head (LandmarkNet, landmarks.py:66-83: final_conv 32 -> nh heat maps + ncls classes) fused with BOTH terms of its
 * loss (landmarks.py:125-134: sum_c w_c mean f(out_c - heatmap_c), f = square | abs, plus DiceLoss of the class channels,
 * loss.py:114-130) on the matrix cores, 16-bit storage only.  z: N x spatial x cin (NDHWC); heatmaps: uint8, N x nh x spatial,
 * sample stride heatmap_stride_n; labels: uint8, N x spatial, sample stride label_stride_n (both 4-byte aligned per sample);
 * `packed` = the head's mednet_conv3d_pack image (its fp32 [co][ci] block is used, split hi + lo for the MFMAs).
 * _fwd writes class_loss, reg_loss and saved[ncls][2] (the Dice sums the backward needs); the logits only if `logits` != NULL
 * (N x (nh + ncls) x spatial fp32).  _bwd rebuilds the logits with the same instructions, forms their gradient in registers and
 * writes dz (N x spatial x cin), dw [nh + ncls][cin], dbias (nullable) and -- gn_y != NULL -- gn_partial[n][rows][cin][2] =
 * {sum du, sum du * gn_y}, du = dz * act'(z), rows = mednet_head_landmark_gn_rows(spatial), for mednet_gn_act_bwd_fused_res.
 * _supported: cin == 32, 1 <= nh <= 16, 1 <= ncls <= 4, spatial % 4 == 0.  Results equal the unfused launches
 * (mednet_conv3d_fwd k=1, mednet_heatmap_loss_*, mednet_dice_*, mednet_head_dgrad_gn, mednet_conv3d_wgrad k=1) up to fp32
 * summation order. */
int mednet_head_landmark_supported(int cin, int nh, int ncls, int dtype, size_t spatial);
size_t mednet_head_landmark_ws_bytes(int n, size_t spatial, int nh, int ncls);
int mednet_head_landmark_gn_rows(size_t spatial);
int mednet_head_landmark_fwd(const void* z, const void* packed, const float* bias, const void* heatmaps, int64_t heatmap_stride_n,
                             const void* labels, int64_t label_stride_n, const float* class_weight, const float* reg_weight,
                             float* logits, float* class_loss, float* reg_loss, float* saved, int n, size_t spatial, int cin, int nh,
                             int ncls, int kind, float eps, int sigmoid, int ignore_index, int z_dtype, void* ws, size_t ws_bytes,
                             mednet_stream stream);
int mednet_head_landmark_bwd(const void* z, const void* packed, const float* bias, const void* heatmaps, int64_t heatmap_stride_n,
                             const void* labels, int64_t label_stride_n, const float* class_weight, const float* reg_weight,
                             const float* saved, const float* dclass_loss, const float* dreg_loss, void* dz, const void* gn_y,
                             int gn_act, float* gn_partial, float* dw, float* dbias, int n, size_t spatial, int cin, int nh, int ncls,
                             int kind, float eps, int sigmoid, int ignore_index, int z_dtype, void* ws, size_t ws_bytes,
                             mednet_stream stream);
/* nn.CrossEntropyLoss(weight)  segmentation.py:49: sum w_y * -log softmax_y / sum w_y.  saved[0] = sum w_y. */
int mednet_ce_fwd(const float* logits, const int64_t* labels, const float* weight, float* loss, float* saved,
                  int n, int c, size_t spatial, int64_t stride_n, int64_t stride_c, int ignore_index, void* ws,
                  size_t ws_bytes, mednet_stream stream);
int mednet_ce_bwd(const float* logits, const int64_t* labels, const float* weight, const float* saved,
                  const float* dloss, float* dlogits, int n, int c, size_t spatial, int64_t stride_n,
                  int64_t stride_c, int ignore_index, mednet_stream stream);
/* LandmarkNet.loss regression term  landmarks.py:129-132: sum_c w_c * mean_{n,v} f(out[:,c]-hm[:,c]),
 * f = square (L2) | abs (L1).  target is fp32 or uint8 (tgt_u8=1), contiguous N x C x spatial. */
int mednet_heatmap_loss_fwd(const float* out, const void* target, const float* cweight, float* loss, int n, int c,
                            size_t spatial, int64_t stride_n, int64_t stride_c, int kind, int tgt_u8, void* ws,
                            size_t ws_bytes, mednet_stream stream);
/* (dout: the layout of `out`, as dlogits above) */
int mednet_heatmap_loss_bwd(const float* out, const void* target, const float* cweight, const float* dloss,
                            float* dout, int n, int c, size_t spatial, int64_t stride_n, int64_t stride_c, int kind,
                            int tgt_u8, mednet_stream stream);
/* ... with an element stride between the samples of `target` (channel ch of sample n at + n * target_stride_n + ch * spatial): the
 * heat-map channels of a label volume (`batch['label'][:, :-1]`, landmarks.py:68) are consumed where they lie, no copy. */
int mednet_heatmap_loss_fwd_strided(const float* out, const void* target, int64_t target_stride_n, const float* cweight, float* loss,
                                    int n, int c, size_t spatial, int64_t stride_n, int64_t stride_c, int kind, int tgt_u8, void* ws,
                                    size_t ws_bytes, mednet_stream stream);
int mednet_heatmap_loss_bwd_strided(const float* out, const void* target, int64_t target_stride_n, const float* cweight,
                                    const float* dloss, float* dout, int n, int c, size_t spatial, int64_t stride_n,
                                    int64_t stride_c, int kind, int tgt_u8, mednet_stream stream);

/* ---- torch.optim.Adam(lr) step  segmentation.py:119-120 (betas .9/.999, eps 1e-8, wd 0), flat fp32 buffers --- */
int mednet_adam_step(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, float grad_scale,
                     mednet_stream stream);
/* fp16 storage (BASELINE config 5): the same update behind dynamic loss scaling, with no host synchronisation.
 * scaler_state (device, 4 floats) = {scale, good steps since the last change, optimizer steps taken, found_inf}; the caller
 * multiplies the loss by scaler_state[0] on the device.  Three launches: raise found_inf if any gradient is NaN/inf; Adam
 * on g / (scale * world) with the device-side step count, skipped entirely when found_inf; then the scale is multiplied
 * by backoff_factor (overflow) or a good step is counted and every growth_interval of them multiply it by growth_factor
 * (torch.cuda.amp.GradScaler's rule; the reference itself trains fp32, train_seg.py:127 has precision=16 commented out). */
int mednet_adam_step_scaled(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1, float beta2,
                            float eps, float weight_decay, float inv_world, float* scaler_state, float growth_factor,
                            float backoff_factor, int growth_interval, mednet_stream stream);

/* ---- inference path ------------------------------------------------------------------------ */
enum { MEDNET_PAD_CONSTANT = 0, MEDNET_PAD_SYMMETRIC = 1 };
/* midasmednet/dataset.py:349-390 (grid_patch_generator): patch b = padded(volume)[:, pos[b] : pos[b] + patch] where the
 * volume (C x D x H x W, fp32) is padded by the overlap in front (np.pad constant-0 or symmetric); pos = B x 3 int32 grid
 * positions on the device.  out: B x C x pD x pH x pW fp32 (what predict.py:85 feeds the network). */
int mednet_grid_gather(const float* volume, const int* pos, float* out, int batch, int c, int d, int h, int w, int pd,
                       int ph, int pw, int ov0, int ov1, int ov2, int pad_mode, mednet_stream stream);
/* examples/predict.py:88-95 + GridPatchSampler.add_processed_batch (dataset.py:446-474) in one pass: logits B x (H +
 * classes) x pD x pH x pW (planar fp32) -> result (H + 1) x D x H x W uint8 (heat maps clipped to 0..255 and truncated,
 * then the arg-max class); only the crop window [crop_start, crop_start + crop_shape) of each patch is written, at
 * pos[b] + offset, and what hangs over the volume is dropped.  The caller derives the window from the reference's slicing
 * (`data[:, o0:-o1, o1:-o1, o2:-o2]`, dataset.py:452-455). */
int mednet_predict_assemble(const float* logits, const int* pos, uint8_t* result, int batch, int num_heatmaps,
                            int num_classes, int d, int h, int w, int pd, int ph, int pw, int crop_start0, int crop_start1,
                            int crop_start2, int crop_d, int crop_h, int crop_w, mednet_stream stream);

/* ---- training-patch sampler ------------------------------------------------------------------ */
/* MedDataset.__getitem__'s crop + cast (dataset.py:313-331) from a device-resident volume src (C x D x H x W; f16 / f32
 * images, u8 labels or heat maps): for i < count,
 *   out[slot[i]][c_off + c][z][y][x] = cast(src[c][pos[i][0] + z][pos[i][1] + y][pos[i][2] + x])
 * out: B x c_total x pD x pH x pW (f32 for images, u8 for labels); pos (count x 3) and slot (count) are int32 on the device.
 * Positions come from the host-side restatement of the reference's sampling (same numpy generator calls). */
int mednet_crop_patches(const void* src, int src_dtype, const int* pos, const int* slot, int count, void* out, int dst_dtype,
                        int c, int d, int h, int w, int c_total, int c_off, int pd, int ph, int pw, mednet_stream stream);
/* ---- intensity augmentation of a batch of cropped patches, in place: the Compose of examples/train_seg.py:82-86
 * (batchgenerators BrightnessTransform -> GammaTransform -> ContrastAugmentationTransform, applied per sample at
 * dataset.py:340-341).  data: batch x channels x spatial fp32; params[batch][channels][3] (device) = {additive brightness,
 * gamma, contrast factor} drawn by the caller in batchgenerators' order (mednet_hip.sampler.draw_augmentation); the
 * data-dependent parts (sample range for the gamma map, channel mean / range for the contrast step) are computed on the
 * device: five launches, no synchronisation. */
size_t mednet_augment_ws_bytes(int batch, int channels, size_t spatial);
int mednet_augment_patches(float* data, const float* params, int batch, int channels, size_t spatial, void* ws,
                           size_t ws_bytes, mednet_stream stream);

#ifdef __cplusplus
}
#endif
#endif
