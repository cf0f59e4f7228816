// C-ABI entry points for the convolution family + error plumbing.  See include/mednet_hip.h.
#include <stdarg.h>
#include <string.h>
#include "common.h"
#include "conv.h"

static thread_local char g_err[512] = "";

int mednet_internal_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int mednet_internal_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mednet_internal_fail(MEDNET_E_HIP, "%s: %s", what, hipGetErrorString(e));
  return MEDNET_OK;
}

struct Option {
  char name[32];
  int value;
};
static Option g_options[48];
static int g_noptions = 0;
int mednet_internal_tuning_option(const char* name, int default_value) {
  for (int i = 0; i < g_noptions; ++i)
    if (strcmp(g_options[i].name, name) == 0) return g_options[i].value;
  return default_value;
}

// CUs of the current device, asked once (the one-workgroup-per-CU kernels are laid out for the MI355X's 256)
int mednet_internal_cu_count(void) {
  const int assumed = mednet_internal_tuning_option("assume_cus", 0);  // plan audits without a device (tests/test_plan_audit.py)
  if (assumed > 0) return assumed;
  static int cus = -1;
  if (cus < 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
      cus = n;
    else
      (void)hipGetLastError();
  }
  return cus;
}

namespace mednet_f16 {  // conv_mfma.hip compiled with -DMEDNET_ELT_F16 -Dmednet=mednet_f16 (fp16 storage)
#include "conv_mfma_decl.inc"
}
using namespace mednet;
// the 16-bit matrix-core family by element type: `dt` is the dtype of the 16-bit operand(s) of the call
#define ELT_CALL(dt, fn, ...) ((dt) == MEDNET_F16 ? mednet_f16::fn(__VA_ARGS__) : mednet::fn(__VA_ARGS__))
static inline bool is16(int dt) { return dt == MEDNET_BF16 || dt == MEDNET_F16; }
// `algo` arguments: a base choice (AUTO / DIRECT / MFMA) and, separately, the request for exact fp32 products in the fp32
// storage mode (MEDNET_ALGO_EXACT_BIT; MEDNET_ALGO_EXACT = AUTO with that request).  MFMA + exact = "the matrix-core path
// is required AND its products must be exact": the fp32 matrix instruction, never the split-bf16 contraction.
static inline bool algo_exact(int a) { return a == MEDNET_ALGO_EXACT || (a > MEDNET_ALGO_EXACT && (a & MEDNET_ALGO_EXACT_BIT)); }
static inline int algo_base(int a) { return (a & 3) == MEDNET_ALGO_EXACT ? MEDNET_ALGO_AUTO : (a & 3); }
// split weights (MEDNET_ALGO_SPLITW_BIT): the 16-bit matrix-core forward / data-gradient convolutions multiply the pack's LOW
// weight images too; the pack must hold them (MEDNET_PACK_LOW for fp16 packs, every bf16 pack).  -> distance of the low image
static inline size_t algo_lo_delta(int a, int dtype, const PackLayout& L) { return (a & MEDNET_ALGO_SPLITW_BIT) && is16(dtype) ? L.lo_delta : 0; }
static inline bool algo_split(int a, int dtype) { return (a & MEDNET_ALGO_SPLITW_BIT) && is16(dtype); }

extern "C" int mednet_set_option(const char* name, int value) {
  for (int i = 0; i < g_noptions; ++i)
    if (strcmp(g_options[i].name, name) == 0) {
      g_options[i].value = value;
      return MEDNET_OK;
    }
  MEDNET_REQUIRE(g_noptions < 48 && strlen(name) < 32, MEDNET_E_UNSUPPORTED, "set_option: table full or name too long");
  strcpy(g_options[g_noptions].name, name);
  g_options[g_noptions++].value = value;
  return MEDNET_OK;
}

extern "C" int mednet_get_option(const char* name, int default_value) { return mednet_internal_tuning_option(name, default_value); }

extern "C" int mednet_abi_version(void) { return 3; }  // 2: `workgroups` argument of the weight-gradient calls (round 5); 3: MEDNET_ALGO_SPLITW_BIT, MEDNET_PACK_LOW (round 6)
extern "C" const char* mednet_last_error(void) { return g_err; }
extern "C" int mednet_device_ok(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return 0;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 0;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// ---- packed weight buffer ---------------------------------------------------------------------------------------
extern "C" size_t mednet_conv3d_pack_bytes(int cin, int cout, int ksize) { return pack_layout(cin, cout, ksize).total; }

extern "C" int mednet_conv3d_pack_elt(const float* w, void* packed, int cin, int cout, int ksize, int transposed_src,
                                      int elt_dtype, mednet_stream stream);
extern "C" int mednet_conv3d_pack(const float* w, void* packed, int cin, int cout, int ksize, int transposed_src,
                                  mednet_stream stream) {
  return mednet_conv3d_pack_elt(w, packed, cin, cout, ksize, transposed_src, MEDNET_BF16, stream);
}
extern "C" int mednet_conv3d_pack_elt(const float* w, void* packed, int cin, int cout, int ksize, int transposed_src,
                                      int elt_dtype, mednet_stream stream) {
  // Every bf16 pack also holds the LOW images of the split-bf16 contraction (conv_x3_mfma.hip), so a buffer packed for
  // bf16 storage serves fp32-storage calls too: they contract against bf16(w) + bf16(w - bf16(w)) and must never find that
  // region unwritten (round 3 wrote it for MEDNET_F32 only; a caller who packed with mednet_conv3d_pack, as the header of
  // round 2 said, then ran an fp32 conv got uninitialised memory).  MEDNET_F32 = MEDNET_BF16 here.  An fp16 pack has fp16
  // images in the high slots and no room for bf16 ones: it serves fp16-storage calls only (include/mednet_hip.h).
  // (round 6) elt_dtype | MEDNET_PACK_LOW: an fp16 pack with the low images too -- elt(w - elt(w)), the same arithmetic in fp16 --
  // for the split-weight mode of the fp16 kernels (MEDNET_ALGO_SPLITW_BIT)
  const bool want_low = (elt_dtype & MEDNET_PACK_LOW) != 0;
  elt_dtype &= ~MEDNET_PACK_LOW;
  if (elt_dtype == MEDNET_F32) elt_dtype = MEDNET_BF16;
  const bool with_low = elt_dtype == MEDNET_BF16 || want_low;
  MEDNET_REQUIRE(is16(elt_dtype), MEDNET_E_DTYPE, "conv3d_pack: the matrix-core images are bf16 or fp16 (got dtype %d)", elt_dtype);
  MEDNET_REQUIRE(ksize == 3 || ksize == 1, MEDNET_E_UNSUPPORTED, "conv3d_pack: kernel size %d (supported: 1, 3)", ksize);
  MEDNET_REQUIRE(cin > 0 && cout > 0, MEDNET_E_SHAPE, "conv3d_pack: bad channels %d -> %d", cin, cout);
  MEDNET_REQUIRE(((size_t)packed & 15) == 0, MEDNET_E_SHAPE, "conv3d_pack: the pack buffer must be 16-byte aligned (the images are written 16 bytes at a time)");
  const PackLayout L = pack_layout(cin, cout, ksize);
  char* base = (char*)packed;
  hipStream_t s = (hipStream_t)stream;
  if (L.mfma_bytes)  // one launch writes the two bf16 fragment images and the two fp32 images
    return ELT_CALL(elt_dtype, launch_pack_mfma, w, base + L.mfma_fwd, base + L.mfma_bwd, (float*)(base + L.f32_fwd),
                    (float*)(base + L.f32_bwd), cin, cout, L.taps, transposed_src, s, with_low ? L.lo_delta : 0);
  return launch_pack_f32(w, (float*)(base + L.f32_fwd), (float*)(base + L.f32_bwd), cin, cout, L.taps, transposed_src, s);
}

// ---- all layers at once: the table is built once per model (pointers of parameters and pack buffers do not move), the launch
//      is repeated after every optimizer step
extern "C" size_t mednet_conv3d_pack_table_bytes(int njobs) { return (size_t)njobs * sizeof(PackJobDev); }
extern "C" int mednet_conv3d_pack_table(const mednet_pack_job* jobs, int njobs, void* table_host, unsigned* max_blocks) {
  MEDNET_REQUIRE(jobs && table_host && max_blocks && njobs > 0, MEDNET_E_SHAPE, "conv3d_pack_table: bad arguments");
  PackJobDev* t = (PackJobDev*)table_host;
  unsigned mb = 0;
  for (int i = 0; i < njobs; ++i) {
    const mednet_pack_job& j = jobs[i];
    const PackLayout L = pack_layout(j.cin, j.cout, j.ksize);
    MEDNET_REQUIRE(j.ksize == 3 && L.mfma_bytes, MEDNET_E_UNSUPPORTED,
                   "conv3d_pack_table: layer %d (%d -> %d, k=%d) has no matrix-core images; pack it with mednet_conv3d_pack", i, j.cin,
                   j.cout, j.ksize);
    MEDNET_REQUIRE(((size_t)j.packed & 15) == 0, MEDNET_E_SHAPE, "conv3d_pack_table: pack buffer of layer %d is not 16-byte aligned", i);
    char* base = (char*)j.packed;
    t[i].w = j.w;
    t[i].sec_fwd = base + L.mfma_fwd;
    t[i].sec_bwd = base + L.mfma_bwd;
    t[i].Pf = (float*)(base + L.f32_fwd);
    t[i].Pb = (float*)(base + L.f32_bwd);
    t[i].cin = j.cin;
    t[i].cout = j.cout;
    t[i].transposed = j.transposed_src;
    t[i].nblocks = pack_mfma_blocks(j.cin, j.cout);
    MEDNET_REQUIRE(L.lo_delta < 0x7fffffffu, MEDNET_E_UNSUPPORTED, "conv3d_pack_table: layer %d too large", i);
    t[i].lo_delta = (int)L.lo_delta;
    t[i].first_block = mb;
    mb += t[i].nblocks;
  }
  *max_blocks = mb;  // (the launch's grid.x: every layer's blocks side by side, see pack_mfma_many_kernel)
  return MEDNET_OK;
}
extern "C" int mednet_conv3d_pack_many(const void* table_device, int njobs, unsigned max_blocks, int elt_dtype,
                                       mednet_stream stream) {
  const bool want_low = (elt_dtype & MEDNET_PACK_LOW) != 0;
  const bool high_only = (elt_dtype & MEDNET_PACK_HIGH_ONLY) != 0;  // (the caller keeps track of what its packs hold: mednet_hip.h)
  elt_dtype &= ~(MEDNET_PACK_LOW | MEDNET_PACK_HIGH_ONLY);
  MEDNET_REQUIRE(!(high_only && elt_dtype == MEDNET_F32), MEDNET_E_UNSUPPORTED, "conv3d_pack_many: MEDNET_PACK_HIGH_ONLY is for the 16-bit storage modes");
  if (elt_dtype == MEDNET_F32) elt_dtype = MEDNET_BF16;
  const bool with_low = (elt_dtype == MEDNET_BF16 && !high_only) || want_low;  // bf16 high + low images always (see mednet_conv3d_pack_elt)
  MEDNET_REQUIRE(is16(elt_dtype), MEDNET_E_DTYPE, "conv3d_pack_many: the matrix-core images are bf16 or fp16 (got dtype %d)", elt_dtype);
  MEDNET_REQUIRE(table_device && njobs > 0 && njobs <= 65535 && max_blocks > 0, MEDNET_E_SHAPE, "conv3d_pack_many: bad arguments");
  return ELT_CALL(elt_dtype, launch_pack_mfma_many, table_device, njobs, max_blocks, (hipStream_t)stream, (with_low ? 1 : 0) | (high_only ? 2 : 0));
}

static int conv_common_checks(const char* who, int n, int d, int h, int w, int cin, int cout, int ksize, int dt1, int dt2) {
  MEDNET_REQUIRE(dtype_ok(dt1) && dtype_ok(dt2), MEDNET_E_DTYPE, "%s: bad dtype", who);
  MEDNET_REQUIRE(ksize == 3 || ksize == 1, MEDNET_E_UNSUPPORTED, "%s: kernel size %d (supported: 1, 3)", who, ksize);
  MEDNET_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, MEDNET_E_SHAPE,
                 "%s: bad shape n=%d d=%d h=%d w=%d cin=%d cout=%d", who, n, d, h, w, cin, cout);
  return MEDNET_OK;
}

extern "C" int mednet_conv3d_fused_stats_chunks(int n, int d, int h, int w, int cin, int cout, int ksize, int x_dtype,
                                                int y_dtype, int algo) {
  if (algo_base(algo) == MEDNET_ALGO_DIRECT || !tuning_option("conv_fuse_stats", 1)) return 0;
  if (x_dtype == MEDNET_F32 && y_dtype == MEDNET_F32) {  // fp32 storage: the split-bf16 forward kernels keep the sums per wave
    if (!(conv_f32_mfma_enabled() && !algo_exact(algo) && conv_x3_enabled())) return 0;
    if (conv_c1_x3_supported(cin, cout, ksize)) return tuning_option("x3_stats", 1) ? conv_c1_x3_stats_rows(d, h, w) : 0;
    return conv_x3_supported(cin, cout, ksize) && conv_x3_fits(d, h, w, cin) ? conv_x3_stats_rows(n, d, h, w, cout) : 0;
  }
  if (ELT_CALL(y_dtype, conv_c1_mfma_supported, cin, cout, ksize, x_dtype, y_dtype, MEDNET_NDHWC, false)) return ELT_CALL(y_dtype, conv_c1_stats_chunks, n, d, h, w, cout);
  if (!ELT_CALL(y_dtype, conv_mfma_supported, cin, cout, ksize, x_dtype, y_dtype, MEDNET_NDHWC, MEDNET_NDHWC, false)) return 0;
  return conv_mfma_stats_chunks(n, d, h, w, cin, cout, false, algo_split(algo, y_dtype));
}

// Launch plan of the producer of a 3x3x3 layer's fused partial rows, computed by the launcher's own planning code (no device
// needed): which kernel, its grid, its work items and how rows are assigned.  tests/test_plan_audit.py derives from it the set
// of (sample, row, channel block) slots the kernel writes and asserts it equals the set mednet_gn_finalize /
// mednet_gn_act_bwd_fused read.
extern "C" int mednet_conv3d_stats_plan(int n, int d, int h, int w, int cin, int cout, int dtype, int gnb, int stride, int* out13) {
  MEDNET_REQUIRE(out13 && n > 0 && d > 0 && h > 0 && w > 0 && (stride == 1 || stride == 2), MEDNET_E_SHAPE, "conv3d_stats_plan: bad arguments");
  MEDNET_REQUIRE(is16(dtype) && cin % 16 == 0 && cout % 16 == 0, MEDNET_E_UNSUPPORTED,
                 "conv3d_stats_plan: the 16-bit matrix-core kernels only (dtype %d, %d -> %d)", dtype, cin, cout);
  return ELT_CALL(dtype, conv_mfma_plan, n, d, h, w, cin, cout, (gnb & 1) != 0, stride, out13, (gnb & 2) != 0);  // (gnb bit 1: split weights)
}

extern "C" int mednet_conv3d_fwd(const void* x, const void* packed, const float* bias, void* y, int n, int d, int h,
                                 int w, int cin, int cout, int ksize, int x_dtype, int x_layout, int y_dtype,
                                 int y_layout, int dgrad, int algo, float* gn_partial, mednet_stream stream) {
  int rc = conv_common_checks("conv3d_fwd", n, d, h, w, cin, cout, ksize, x_dtype, y_dtype);
  if (rc) return rc;
  // the pack was built for the layer's (Cin,Cout); a dgrad call swaps the roles
  const PackLayout L = dgrad ? pack_layout(cout, cin, ksize) : pack_layout(cin, cout, ksize);
  const char* base = (const char*)packed;
  hipStream_t s = (hipStream_t)stream;
  const bool mfma_ok = ELT_CALL(y_dtype, conv_mfma_supported, cin, cout, ksize, x_dtype, y_dtype, x_layout, y_layout, bias != nullptr) &&
                       conv_mfma_fits(n, d, h, w, cin);
  // fp32 storage (the 1e-3 parity mode): 3x3x3 forward / data gradient on the matrix cores -- the split-bf16 contraction (three
  // bf16 MFMAs per product; the pack must hold the low images: mednet_conv3d_pack_elt(MEDNET_F32)) or, for shapes it does not
  // take and under ALGO_EXACT, the fp32 matrix instruction.  Both satisfy MEDNET_ALGO_MFMA ("a matrix-core path is required").
  const bool f32_mode = algo_base(algo) != MEDNET_ALGO_DIRECT && conv_f32_mfma_enabled() && ksize == 3 && x_dtype == MEDNET_F32 &&
                        y_dtype == MEDNET_F32 && (x_layout == MEDNET_NDHWC || cin == 1) && y_layout == MEDNET_NDHWC;
  if (algo_base(algo) == MEDNET_ALGO_MFMA && !mfma_ok && !f32_mode)
    return fail(MEDNET_E_UNSUPPORTED, "conv3d_fwd: MFMA path does not take cin=%d cout=%d k=%d dtypes %d->%d", cin, cout,
                ksize, x_dtype, y_dtype);
  if (mfma_ok && algo_base(algo) != MEDNET_ALGO_DIRECT)
    return ELT_CALL(y_dtype, launch_conv_mfma, x, base + (dgrad ? L.mfma_bwd : L.mfma_fwd), y, n, d, h, w, cin, cout, x_dtype,
                    y_dtype, gn_partial, s, MEDNET_ACT_NONE, nullptr, algo_lo_delta(algo, y_dtype, L));
  // first layer (one input channel): contraction over the 27 taps on the matrix cores
  if (!dgrad && algo_base(algo) != MEDNET_ALGO_DIRECT && x_layout == MEDNET_NDHWC &&
      ELT_CALL(y_dtype, conv_c1_mfma_supported, cin, cout, ksize, x_dtype, y_dtype, y_layout, bias != nullptr))
    return ELT_CALL(y_dtype, launch_conv_c1_mfma, x, (const float*)(base + L.f32_fwd), y, n, d, h, w, cout, gn_partial, s, x_dtype,
                    algo_split(algo, y_dtype) ? 1 : 0);
  // fp32 storage: the split-bf16 contraction ...
  if (f32_mode && (!algo_exact(algo) && conv_x3_enabled()) && L.mfma_bytes && conv_x3_supported(cin, cout, ksize) && conv_x3_fits(d, h, w, cin))
    return launch_conv_x3(x, base + (dgrad ? L.mfma_bwd : L.mfma_fwd), L.lo_delta, bias, y, n, d, h, w, cin, cout, gn_partial, s);
  if (f32_mode && !dgrad && !algo_exact(algo) && conv_x3_enabled() && conv_c1_x3_supported(cin, cout, ksize))  // first layer
    return launch_conv_c1_x3(x, (const float*)(base + L.f32_fwd), bias, y, n, d, h, w, cout, gn_partial, s);
  MEDNET_REQUIRE(gn_partial == nullptr, MEDNET_E_UNSUPPORTED,
                 "conv3d_fwd: fused GroupNorm partials are only produced by the MFMA paths (ask mednet_conv3d_fused_stats_chunks)");
  // ... or, for the shapes those kernels do not take (odd channel counts; ALGO_EXACT), on the fp32 matrix-core instruction
  if (f32_mode)
    return launch_conv_f32_mfma(x, (const float*)(base + (dgrad ? L.f32_bwd : L.f32_fwd)), bias, y, n, d, h, w, cin, cout, s);
  // 1x1x1 head forward (channels-last features -> planar fp32 logits): the packed backward image Pb[t=0][co][ci] = W[m][k]
  if (!dgrad && algo_base(algo) != MEDNET_ALGO_DIRECT && ksize == 1 && x_layout == MEDNET_NDHWC && y_layout == MEDNET_NCDHW &&
      y_dtype == MEDNET_F32 && head_vox_supported(cin))
    return launch_head_fwd_vox(x, (const float*)(base + L.f32_bwd), bias, (float*)y, n, (size_t)d * h * w, cin, cout, x_dtype, s);
  // data gradient of the 1x1x1 head: the packed backward image Pb[t=0][co][ci] is exactly W[m][k]
  if (dgrad && !bias && algo_base(algo) != MEDNET_ALGO_DIRECT && head_dgrad_supported(cin, cout, ksize, x_dtype, x_layout, y_layout))
    return launch_head_dgrad(x, (const float*)(base + L.f32_bwd), y, n, (size_t)d * h * w, cin, cout, y_dtype, s);
  ConvGeom g;
  g.n = n; g.od = d; g.oh = h; g.ow = w; g.id = d; g.ih = h; g.iw = w;
  g.k = cin; g.m = cout; g.ks = ksize;
  g.in_planar = x_layout == MEDNET_NCDHW; g.out_planar = y_layout == MEDNET_NCDHW;
  return launch_direct<MAP_CONV>(x, (const float*)(base + (dgrad ? L.f32_bwd : L.f32_fwd)), bias, nullptr, y, g, x_dtype,
                                 y_dtype, s);
}

extern "C" size_t mednet_conv3d_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout, int ksize, int workgroups) {
  const size_t a = wgrad_direct_ws_bytes((size_t)n * d * h * w, cout, cin, ksize);
  // (the 16-bit element types share one plan; `workgroups`: see mednet_conv3d_wgrad)
  const size_t b = wgrad_mfma_ws_bytes(n, d, h, w, cin, cout, ksize, workgroups);
  const size_t c1 = cin == 1 ? wgrad_c1_ws_bytes(n, d, h, w, cout) : 0;
  const size_t c2 = ksize == 1 ? wgrad_1x1_ws_bytes(n, (size_t)d * h * w, cin, cout) : 0;
  const size_t f = ksize == 3 ? wgrad_f32_mfma_ws_bytes(n, d, h, w, cout, cin, 0) : 0;
  const size_t f3 = conv_x3_supported(cin, cout, ksize) ? wgrad_x3_ws_bytes(n, d, h, w, cin, cout, workgroups) : 0;
  size_t m = a > b ? a : b;
  if (f > m) m = f;
  if (f3 > m) m = f3;
  if (c1 > m) m = c1;
  if (c2 > m) m = c2;
  // the bias-gradient partials live behind the weight-gradient partials
  return align_up(m, 256) + channel_sum_ws_bytes(n, (size_t)d * h * w, cout) + 256;
}

// Launch plan of a 3x3x3 weight gradient on the 16-bit matrix-core path, from the launcher's own planning code (no device needed)
extern "C" int mednet_conv3d_wgrad_plan(int n, int d, int h, int w, int cin, int cout, int dtype, int workgroups, int* out10) {
  MEDNET_REQUIRE(out10 && n > 0 && d > 0 && h > 0 && w > 0 && workgroups >= 0, MEDNET_E_SHAPE, "conv3d_wgrad_plan: bad arguments");
  MEDNET_REQUIRE(is16(dtype) && cin % 16 == 0 && cout % 16 == 0, MEDNET_E_UNSUPPORTED,
                 "conv3d_wgrad_plan: the 16-bit matrix-core kernels only (dtype %d, %d -> %d)", dtype, cin, cout);
  return ELT_CALL(dtype, wgrad_mfma_plan, n, d, h, w, cin, cout, workgroups, out10);
}

// Does this weight gradient run on the kernel that leaves half of every CU (registers, wave slots, 60 KB of LDS) to other streams?
// A caller that launches weight gradients beside its main stream asks for all CUs then (workgroups = 0) instead of half of them.
extern "C" int mednet_conv3d_wgrad_coresident(int n, int d, int h, int w, int cin, int cout, int ksize, int x_dtype, int dy_dtype,
                                              int algo) {
  if (algo_base(algo) == MEDNET_ALGO_DIRECT || !is16(dy_dtype) || x_dtype != dy_dtype || ksize != 3) return 0;
  if (!ELT_CALL(dy_dtype, wgrad_mfma_supported, cin, cout, ksize, x_dtype, dy_dtype, MEDNET_NDHWC, MEDNET_NDHWC) ||
      !wgrad_mfma_fits(n, d, h, w, cin > cout ? cin : cout, 1))
    return 0;
  return ELT_CALL(dy_dtype, wgrad_mfma_coresident, d, h, w) ? 1 : 0;
}

// First-layer (Cin = 1) 3x3x3 weight gradient taken straight from the gradient dz of the layer's activated GroupNorm output:
// GroupNorm's backward apply pass (mednet_gn_act_bwd_fused's third kernel) happens while dz and y are staged, so dy is neither
// written nor read back.  `bcoef` comes from mednet_gn_bwd_coefficients; workspace as mednet_conv3d_wgrad_ws_bytes(.., cin = 1, ..).
extern "C" int mednet_conv3d_wgrad_c1_gn_supported(int cout, int x_dtype, int dtype) {
  return is16(dtype) && wgrad_c1_gn_supported(cout, x_dtype, dtype) ? 1 : 0;
}
extern "C" int mednet_conv3d_wgrad_c1_gn(const void* x, const void* dz, const void* y, const float* coef, const float* bcoef,
                                         float* dw, int n, int d, int h, int w, int cout, int act, int x_dtype, int dtype,
                                         void* ws, size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(x && dz && y && coef && bcoef && dw && ws, MEDNET_E_SHAPE, "conv3d_wgrad_c1_gn: null argument");
  MEDNET_REQUIRE(is16(dtype), MEDNET_E_DTYPE, "conv3d_wgrad_c1_gn: dz / y must be 16-bit (dtype %d)", dtype);
  const int rc = conv_common_checks("conv3d_wgrad_c1_gn", n, d, h, w, 1, cout, 3, x_dtype, dtype);
  if (rc) return rc;
  return launch_wgrad_c1_gn(x, dz, y, coef, bcoef, act, dw, n, d, h, w, cout, x_dtype, dtype, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mednet_conv3d_wgrad(const void* x, const void* dy, float* dw, float* dbias, int n, int d, int h, int w,
                                   int cin, int cout, int ksize, int x_dtype, int x_layout, int dy_dtype, int dy_layout,
                                   int algo, int workgroups, void* ws, size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(workgroups >= 0, MEDNET_E_SHAPE, "conv3d_wgrad: workgroups %d (0 = one per CU)", workgroups);
  int rc = conv_common_checks("conv3d_wgrad", n, d, h, w, cin, cout, ksize, x_dtype, dy_dtype);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (dbias) {
    const size_t need = channel_sum_ws_bytes(n, (size_t)d * h * w, cout);
    MEDNET_REQUIRE(ws_bytes >= need, MEDNET_E_WORKSPACE, "conv3d_wgrad: workspace too small for the bias gradient");
    rc = launch_channel_sum(dy, dbias, n, (size_t)d * h * w, cout, dy_layout == MEDNET_NCDHW, dy_dtype,
                            (char*)ws + (ws_bytes - need) / 256 * 256, need, s);
    if (rc) return rc;
    ws_bytes = (ws_bytes - need) / 256 * 256;
  }
  if (algo_base(algo) != MEDNET_ALGO_DIRECT && wgrad_c1_supported(cin, cout, ksize, x_layout, dy_layout))
    return launch_wgrad_c1(x, dy, dw, n, d, h, w, cout, x_dtype, dy_dtype, ws, ws_bytes, s, !algo_exact(algo) && conv_x3_enabled());
  if (algo_base(algo) != MEDNET_ALGO_DIRECT && wgrad_1x1_supported(cin, cout, ksize, x_layout, dy_layout, dy_dtype))
    return launch_wgrad_1x1(x, dy, dw, n, (size_t)d * h * w, cin, cout, x_dtype, ws, ws_bytes, s);
  if (algo_base(algo) != MEDNET_ALGO_DIRECT && conv_f32_mfma_enabled() && ksize == 3 && x_dtype == MEDNET_F32 && dy_dtype == MEDNET_F32 &&
      (x_layout == MEDNET_NDHWC || cin == 1) && dy_layout == MEDNET_NDHWC) {
    const int cmax = cin > cout ? cin : cout;
    if ((!algo_exact(algo) && conv_x3_enabled()) && conv_x3_supported(cin, cout, ksize) && conv_x3_fits(d, h, w, cmax))
      return launch_wgrad_x3(x, dy, dw, n, d, h, w, cin, cout, ws, ws_bytes, s, workgroups);  // split-bf16 contraction over the voxels
    return launch_wgrad_f32_mfma(x, dy, dw, n, d, h, w, cin, cout, ws, ws_bytes, s);
  }
  const bool mfma_ok = ELT_CALL(dy_dtype, wgrad_mfma_supported, cin, cout, ksize, x_dtype, dy_dtype, x_layout, dy_layout) &&
                       wgrad_mfma_fits(n, d, h, w, cin > cout ? cin : cout, 1);
  if (algo_base(algo) == MEDNET_ALGO_MFMA && !mfma_ok)
    return fail(MEDNET_E_UNSUPPORTED, "conv3d_wgrad: MFMA path does not take cin=%d cout=%d k=%d", cin, cout, ksize);
  if (mfma_ok && algo_base(algo) != MEDNET_ALGO_DIRECT)
    return ELT_CALL(dy_dtype, launch_wgrad_mfma, x, dy, dw, n, d, h, w, cin, cout, x_dtype, ws, ws_bytes, s, workgroups);
  WgradGeom g;
  g.n = n; g.ad = d; g.ah = h; g.aw = w; g.bd = d; g.bh = h; g.bw = w;
  g.ka = cout; g.kb = cin; g.ks = ksize; g.stride2 = 0;
  g.a_planar = dy_layout == MEDNET_NCDHW; g.b_planar = x_layout == MEDNET_NCDHW;
  g.chunk = 0;
  return launch_wgrad_direct(dy, x, dw, g, dy_dtype, x_dtype, ws, ws_bytes, s);
}

// conv 3x3x3 (no bias) + activation in the epilogue: the 'gcr' / 'gcl' / 'gce' orders of components.py:12-67 (UNet3D)
extern "C" int mednet_conv3d_act_supported(int n, int d, int h, int w, int cin, int cout, int algo) {
  return algo_base(algo) != MEDNET_ALGO_DIRECT &&
         conv_mfma_supported(cin, cout, 3, MEDNET_BF16, MEDNET_BF16, MEDNET_NDHWC, MEDNET_NDHWC, false) &&
         conv_mfma_fits(n, d, h, w, cin) && conv_mfma_fits(n, d, h, w, cout);
}
extern "C" int mednet_conv3d_act_fwd(const void* x, const void* packed, void* y, int n, int d, int h, int w, int cin,
                                     int cout, int act, int algo, float* gn_partial, int dtype, mednet_stream stream) {
  MEDNET_REQUIRE(is16(dtype), MEDNET_E_DTYPE, "conv3d_act_fwd: 16-bit storage only (dtype %d)", dtype);
  int rc = conv_common_checks("conv3d_act_fwd", n, d, h, w, cin, cout, 3, dtype, dtype);
  if (rc) return rc;
  MEDNET_REQUIRE(act >= MEDNET_ACT_NONE && act <= MEDNET_ACT_ELU, MEDNET_E_UNSUPPORTED, "conv3d_act_fwd: activation %d", act);
  if (!mednet_conv3d_act_supported(n, d, h, w, cin, cout, algo))
    return fail(MEDNET_E_UNSUPPORTED, "conv3d_act_fwd: only the bf16 matrix-core path fuses the activation (cin=%d cout=%d)", cin, cout);
  const PackLayout L = pack_layout(cin, cout, 3);
  return ELT_CALL(dtype, launch_conv_mfma, x, (const char*)packed + L.mfma_fwd, y, n, d, h, w, cin, cout, dtype, dtype,
                  gn_partial, (hipStream_t)stream, act, nullptr, algo_lo_delta(algo, dtype, L));
}

// data gradient of a 3x3x3 conv with a second gradient of the same tensor summed in the epilogue (matrix-core path only)
// fp32 storage: is the split-bf16 data-gradient kernel (which can sum a second gradient and take GroupNorm-backward sums) taking
// the layer Cin -> Cout?
static bool x3_dgrad_ok(int n, int d, int h, int w, int cin, int cout, int algo) {
  (void)n;
  return algo_base(algo) != MEDNET_ALGO_DIRECT && !algo_exact(algo) && conv_f32_mfma_enabled() && conv_x3_enabled() &&
         conv_x3_supported(cin, cout, 3) && conv_x3_fits(d, h, w, cout);
}
extern "C" int mednet_conv3d_dgrad_add_supported(int n, int d, int h, int w, int cin, int cout, int algo, int dtype) {
  if (dtype == MEDNET_F32) return x3_dgrad_ok(n, d, h, w, cin, cout, algo) ? 1 : 0;
  return is16(dtype) ? mednet_conv3d_act_supported(n, d, h, w, cout, cin, algo) : 0;
}
extern "C" int mednet_conv3d_dgrad_add(const void* dy, const void* packed, const void* add, void* dx, int n, int d, int h,
                                       int w, int cin, int cout, int algo, int dtype, mednet_stream stream) {
  if (dtype == MEDNET_F32) {
    int rc32 = conv_common_checks("conv3d_dgrad_add", n, d, h, w, cin, cout, 3, dtype, dtype);
    if (rc32) return rc32;
    MEDNET_REQUIRE(x3_dgrad_ok(n, d, h, w, cin, cout, algo), MEDNET_E_UNSUPPORTED,
                   "conv3d_dgrad_add: fp32 storage fuses the add only on the split-bf16 path (cin=%d cout=%d)", cin, cout);
    const PackLayout L32 = pack_layout(cin, cout, 3);
    return launch_conv_x3_dgrad(dy, (const char*)packed + L32.mfma_bwd, L32.lo_delta, dx, n, d, h, w, cout, cin, add, nullptr, nullptr,
                                MEDNET_ACT_NONE, nullptr, (hipStream_t)stream);
  }
  MEDNET_REQUIRE(is16(dtype), MEDNET_E_DTYPE, "conv3d_dgrad_add: 16-bit storage only (dtype %d)", dtype);
  int rc = conv_common_checks("conv3d_dgrad_add", n, d, h, w, cin, cout, 3, dtype, dtype);
  if (rc) return rc;
  // (roles swapped as in mednet_conv3d_fwd(dgrad=1): the kernel reads dy with Cout channels and writes Cin channels)
  if (!mednet_conv3d_act_supported(n, d, h, w, cout, cin, algo))
    return fail(MEDNET_E_UNSUPPORTED, "conv3d_dgrad_add: only the bf16 matrix-core path fuses the add (cin=%d cout=%d)", cin, cout);
  const PackLayout L = pack_layout(cin, cout, 3);
  return ELT_CALL(dtype, launch_conv_mfma, dy, (const char*)packed + L.mfma_bwd, dx, n, d, h, w, cout, cin, dtype, dtype, nullptr,
                  (hipStream_t)stream, MEDNET_ACT_NONE, add, algo_lo_delta(algo, dtype, L));
}

// data gradient of the 1x1x1 head + the first pass of the GroupNorm-3 backward of the ExtResNetBlock whose output it is
extern "C" int mednet_head_dgrad_gn_rows(int n, int d, int h, int w, int cin, int dtype) {
  (void)n;
  if (!tuning_option("gn3_fuse", 1)) return 0;
  return head_dgrad_gn_rows((size_t)d * h * w, cin, dtype);
}
extern "C" int mednet_head_dgrad_gn(const void* dy, const void* packed, void* dx, const void* gn_y, const void* gn_z, int gn_act,
                                    float* gn_partial, int n, int d, int h, int w, int cin, int cout, int dtype,
                                    mednet_stream stream) {
  MEDNET_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && dy && packed && dx && gn_y && gn_z && gn_partial,
                 MEDNET_E_SHAPE, "head_dgrad_gn: bad arguments");
  MEDNET_REQUIRE(mednet_head_dgrad_gn_rows(n, d, h, w, cin, dtype) > 0, MEDNET_E_UNSUPPORTED,
                 "head_dgrad_gn: cin=%d dtype=%d not supported", cin, dtype);
  const PackLayout L = pack_layout(cin, cout, 1);
  return launch_head_dgrad_gn(dy, (const float*)((const char*)packed + L.f32_bwd), dx, gn_y, gn_z, gn_act, gn_partial, n,
                              (size_t)d * h * w, cout, cin, dtype, (hipStream_t)stream);
}

// ---- the landmark head fused with its two losses (head_mfma.hip) ---------------------------------------------------------
extern "C" int mednet_head_landmark_supported(int cin, int nh, int ncls, int dtype, size_t spatial) {
  return is16(dtype) && ELT_CALL(dtype, head_lm_supported, cin, nh, ncls, dtype, spatial) ? 1 : 0;
}
extern "C" size_t mednet_head_landmark_ws_bytes(int n, size_t spatial, int nh, int ncls) {
  return head_lm_ws_bytes(n, spatial, nh, ncls);
}
extern "C" int mednet_head_landmark_gn_rows(size_t spatial) { return head_lm_chunks(spatial); }
extern "C" int mednet_head_landmark_fwd(const void* z, const void* packed, const float* bias, const void* heatmaps,
                                        int64_t heatmap_stride_n, const void* labels, int64_t label_stride_n,
                                        const float* class_weight, const float* reg_weight, float* logits, float* class_loss,
                                        float* reg_loss, float* saved, int n, size_t spatial, int cin, int nh, int ncls, int kind,
                                        float eps, int sigmoid, int ignore_index, int z_dtype, void* ws, size_t ws_bytes,
                                        mednet_stream stream) {
  MEDNET_REQUIRE(n > 0 && spatial > 0 && z && packed && heatmaps && labels && class_loss && reg_loss && saved && ws, MEDNET_E_SHAPE,
                 "head_landmark_fwd: bad arguments");
  MEDNET_REQUIRE(mednet_head_landmark_supported(cin, nh, ncls, z_dtype, spatial), MEDNET_E_UNSUPPORTED,
                 "head_landmark_fwd: %d -> %d heat maps + %d classes, dtype %d, %zu voxels", cin, nh, ncls, z_dtype, spatial);
  MEDNET_REQUIRE(kind == MEDNET_REG_L2 || kind == MEDNET_REG_L1, MEDNET_E_UNSUPPORTED, "head_landmark_fwd: regression kind %d", kind);
  MEDNET_REQUIRE(ws_bytes >= head_lm_ws_bytes(n, spatial, nh, ncls), MEDNET_E_WORKSPACE, "head_landmark_fwd: workspace too small");
  const PackLayout L = pack_layout(cin, nh + ncls, 1);
  const float* W = (const float*)((const char*)packed + L.f32_bwd);  // [co][ci]
  hipStream_t s = (hipStream_t)stream;
  const int chunks = head_lm_chunks(spatial);
  float* hm_partial = (float*)ws;
  float* dice_partial = hm_partial + (size_t)n * nh * chunks;
  int rc = ELT_CALL(z_dtype, launch_head_lm_fwd, z, W, bias, heatmaps, heatmap_stride_n, labels, label_stride_n, logits, hm_partial,
                    dice_partial, n, spatial, nh, ncls, kind, sigmoid, ignore_index, s);
  if (rc) return rc;
  rc = launch_hm_finalize(hm_partial, reg_weight, reg_loss, n, nh, chunks, spatial, s);
  if (rc) return rc;
  return launch_dice_finalize(dice_partial, class_weight, class_loss, saved, ncls, n * chunks, eps, s);
}
extern "C" int mednet_head_landmark_bwd(const void* z, const void* packed, const float* bias, const void* heatmaps,
                                        int64_t heatmap_stride_n, const void* labels, int64_t label_stride_n,
                                        const float* class_weight, const float* reg_weight, const float* saved,
                                        const float* dclass_loss, const float* dreg_loss, void* dz, const void* gn_y, int gn_act,
                                        float* gn_partial, float* dw, float* dbias, int n, size_t spatial, int cin, int nh, int ncls,
                                        int kind, float eps, int sigmoid, int ignore_index, int z_dtype, void* ws, size_t ws_bytes,
                                        mednet_stream stream) {
  MEDNET_REQUIRE(n > 0 && spatial > 0 && z && packed && heatmaps && labels && saved && dclass_loss && dreg_loss && dz && dw && ws,
                 MEDNET_E_SHAPE, "head_landmark_bwd: bad arguments");
  MEDNET_REQUIRE(mednet_head_landmark_supported(cin, nh, ncls, z_dtype, spatial), MEDNET_E_UNSUPPORTED,
                 "head_landmark_bwd: %d -> %d heat maps + %d classes, dtype %d, %zu voxels", cin, nh, ncls, z_dtype, spatial);
  const PackLayout L = pack_layout(cin, nh + ncls, 1);
  const float* W = (const float*)((const char*)packed + L.f32_bwd);
  return ELT_CALL(z_dtype, launch_head_lm_bwd, z, W, bias, heatmaps, heatmap_stride_n, labels, label_stride_n, saved, class_weight,
                  reg_weight, dclass_loss, dreg_loss, eps, dz, gn_y, gn_act, gn_partial, dw, dbias, n, spatial, nh, ncls, kind,
                  sigmoid, ignore_index, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mednet_conv3d_dgrad_gn_rows(int n, int d, int h, int w, int cin, int cout, int algo) {
  if (!tuning_option("conv_fuse_gnb", 1) || !mednet_conv3d_act_supported(n, d, h, w, cout, cin, algo)) return 0;
  // (the kernel reads the layer's Cout channels, writes its Cin; 16-bit storage only, so the split-weight request counts as given)
  return conv_mfma_stats_chunks(n, d, h, w, cout, cin, true, (algo & MEDNET_ALGO_SPLITW_BIT) != 0);
}
extern "C" int mednet_conv3d_dgrad_gn_rows_dt(int n, int d, int h, int w, int cin, int cout, int algo, int dtype) {
  if (dtype == MEDNET_F32)  // (the kernel writes the layer's Cin channels: they are its "output" channel blocks)
    return tuning_option("conv_fuse_gnb", 1) && x3_dgrad_ok(n, d, h, w, cin, cout, algo) ? conv_x3_stats_rows(n, d, h, w, cin) : 0;
  return is16(dtype) ? mednet_conv3d_dgrad_gn_rows(n, d, h, w, cin, cout, algo) : 0;
}
extern "C" int mednet_conv3d_dgrad_gn(const void* dy, const void* packed, const void* add, void* dx, const void* gn_y,
                                      const float* gn_coef, int gn_act, float* gn_partial, int n, int d, int h, int w,
                                      int cin, int cout, int algo, int dtype, mednet_stream stream) {
  if (dtype == MEDNET_F32) {
    int rc32 = conv_common_checks("conv3d_dgrad_gn", n, d, h, w, cin, cout, 3, dtype, dtype);
    if (rc32) return rc32;
    MEDNET_REQUIRE(gn_y && gn_coef && gn_partial, MEDNET_E_SHAPE, "conv3d_dgrad_gn: gn_y, gn_coef and gn_partial are required");
    MEDNET_REQUIRE(gn_act >= MEDNET_ACT_NONE && gn_act <= MEDNET_ACT_ELU, MEDNET_E_UNSUPPORTED, "conv3d_dgrad_gn: activation %d", gn_act);
    MEDNET_REQUIRE(mednet_conv3d_dgrad_gn_rows_dt(n, d, h, w, cin, cout, algo, dtype) > 0, MEDNET_E_UNSUPPORTED,
                   "conv3d_dgrad_gn: fp32 storage fuses the GroupNorm sums only on the split-bf16 path (cin=%d cout=%d)", cin, cout);
    const PackLayout L32 = pack_layout(cin, cout, 3);
    return launch_conv_x3_dgrad(dy, (const char*)packed + L32.mfma_bwd, L32.lo_delta, dx, n, d, h, w, cout, cin, add, gn_y, gn_coef,
                                gn_act, gn_partial, (hipStream_t)stream);
  }
  MEDNET_REQUIRE(is16(dtype), MEDNET_E_DTYPE, "conv3d_dgrad_gn: 16-bit storage only (dtype %d)", dtype);
  int rc = conv_common_checks("conv3d_dgrad_gn", n, d, h, w, cin, cout, 3, dtype, dtype);
  if (rc) return rc;
  MEDNET_REQUIRE(gn_y && gn_coef && gn_partial, MEDNET_E_SHAPE, "conv3d_dgrad_gn: gn_y, gn_coef and gn_partial are required");
  MEDNET_REQUIRE(gn_act >= MEDNET_ACT_NONE && gn_act <= MEDNET_ACT_ELU, MEDNET_E_UNSUPPORTED, "conv3d_dgrad_gn: activation %d", gn_act);
  if (mednet_conv3d_dgrad_gn_rows(n, d, h, w, cin, cout, algo) == 0)
    return fail(MEDNET_E_UNSUPPORTED, "conv3d_dgrad_gn: only the bf16 matrix-core path fuses the GroupNorm sums (cin=%d cout=%d)", cin, cout);
  const PackLayout L = pack_layout(cin, cout, 3);
  return ELT_CALL(dtype, launch_conv_mfma_gnb, dy, (const char*)packed + L.mfma_bwd, dx, n, d, h, w, cout, cin, add, gn_y,
                  gn_coef, gn_act, gn_partial, (hipStream_t)stream, algo_lo_delta(algo, dtype, L));
}

// ---- ConvTranspose3d(k3,s2,p1,op1) ------------------------------------------------------------------------------------
extern "C" int mednet_convt3d_fwd(const void* x, const void* packed, const float* bias, const void* skip, void* y, int n,
                                  int d, int h, int w, int cin, int cout, int x_dtype, int y_dtype, int algo,
                                  mednet_stream stream) {
  int rc = conv_common_checks("convt3d_fwd", n, d, h, w, cin, cout, 3, x_dtype, y_dtype);
  if (rc) return rc;
  const PackLayout L = pack_layout(cin, cout, 3);
  const bool mfma_ok = L.mfma_bytes && cin % 32 == 0 && cout % 32 == 0 && is16(x_dtype) && y_dtype == x_dtype;
  const bool f32_mfma = conv_f32_mfma_enabled() && x_dtype == MEDNET_F32 && y_dtype == MEDNET_F32;  // (fp32 storage: see conv3d_fwd)
  if (algo_base(algo) == MEDNET_ALGO_MFMA && !mfma_ok && !f32_mfma)
    return fail(MEDNET_E_UNSUPPORTED, "convt3d_fwd: MFMA path does not take cin=%d cout=%d", cin, cout);
  if (mfma_ok && algo_base(algo) != MEDNET_ALGO_DIRECT)
    return ELT_CALL(x_dtype, launch_convt_fwd_mfma, x, (const char*)packed + L.mfma_fwd, bias, skip, y, n, d, h, w, cin, cout,
                    (hipStream_t)stream, algo_lo_delta(algo, x_dtype, L));
  if (algo_base(algo) != MEDNET_ALGO_DIRECT && conv_f32_mfma_enabled() && x_dtype == MEDNET_F32 && y_dtype == MEDNET_F32) {
    if ((!algo_exact(algo) && conv_x3_enabled()) && L.mfma_bytes && conv_x3_supported(cin, cout, 3))
      return launch_convt_fwd_x3(x, (const char*)packed + L.mfma_fwd, L.lo_delta, bias, skip, y, n, d, h, w, cin, cout,
                                 (hipStream_t)stream);
    return launch_convt_fwd_f32_mfma(x, (const float*)((const char*)packed + L.f32_fwd), bias, skip, y, n, d, h, w, cin, cout,
                                     (hipStream_t)stream);
  }
  ConvGeom g;
  g.n = n; g.od = 2 * d; g.oh = 2 * h; g.ow = 2 * w; g.id = d; g.ih = h; g.iw = w;
  g.k = cin; g.m = cout; g.ks = 3; g.in_planar = 0; g.out_planar = 0;
  return launch_direct<MAP_CT_FWD>(x, (const float*)((const char*)packed + L.f32_fwd), bias, skip, y, g, x_dtype, y_dtype,
                                   (hipStream_t)stream);
}

extern "C" int mednet_convt3d_dgrad(const void* dy, const void* packed, void* dx, int n, int d, int h, int w, int cin,
                                    int cout, int dy_dtype, int dx_dtype, int algo, mednet_stream stream) {
  int rc = conv_common_checks("convt3d_dgrad", n, d, h, w, cin, cout, 3, dy_dtype, dx_dtype);
  if (rc) return rc;
  const PackLayout L = pack_layout(cin, cout, 3);
  const bool mfma_ok = L.mfma_bytes && cin % 32 == 0 && cout % 32 == 0 && is16(dy_dtype) && dx_dtype == dy_dtype &&
                       conv_mfma_fits(n, 2 * d, 2 * h, 2 * w, cout);
  const bool f32_mfma = conv_f32_mfma_enabled() && dy_dtype == MEDNET_F32 && dx_dtype == MEDNET_F32;
  if (algo_base(algo) == MEDNET_ALGO_MFMA && !mfma_ok && !f32_mfma)
    return fail(MEDNET_E_UNSUPPORTED, "convt3d_dgrad: MFMA path does not take cin=%d cout=%d", cin, cout);
  if (mfma_ok && algo_base(algo) != MEDNET_ALGO_DIRECT)
    return ELT_CALL(dy_dtype, launch_convt_dgrad_mfma, dy, (const char*)packed + L.mfma_bwd, dx, n, d, h, w, cin, cout,
                    (hipStream_t)stream, algo_lo_delta(algo, dy_dtype, L));
  if (algo_base(algo) != MEDNET_ALGO_DIRECT && conv_f32_mfma_enabled() && dy_dtype == MEDNET_F32 && dx_dtype == MEDNET_F32) {
    if ((!algo_exact(algo) && conv_x3_enabled()) && L.mfma_bytes && conv_x3_supported(cin, cout, 3) && conv_x3_fits(2 * d, 2 * h, 2 * w, cout))
      return launch_convt_dgrad_x3(dy, (const char*)packed + L.mfma_bwd, L.lo_delta, dx, n, d, h, w, cin, cout, (hipStream_t)stream);
    return launch_convt_dgrad_f32_mfma(dy, (const float*)((const char*)packed + L.f32_bwd), dx, n, d, h, w, cin, cout,
                                       (hipStream_t)stream);
  }
  ConvGeom g;
  g.n = n; g.od = d; g.oh = h; g.ow = w; g.id = 2 * d; g.ih = 2 * h; g.iw = 2 * w;
  g.k = cout; g.m = cin; g.ks = 3; g.in_planar = 0; g.out_planar = 0;
  return launch_direct<MAP_CT_DG>(dy, (const float*)((const char*)packed + L.f32_bwd), nullptr, nullptr, dx, g, dy_dtype,
                                  dx_dtype, (hipStream_t)stream);
}

static bool convt_dgrad_mfma_ok(int n, int d, int h, int w, int cin, int cout, int dtype) {
  const PackLayout L = pack_layout(cin, cout, 3);
  return L.mfma_bytes && cin % 32 == 0 && cout % 32 == 0 && is16(dtype) && conv_mfma_fits(n, 2 * d, 2 * h, 2 * w, cout);
}
extern "C" int mednet_convt3d_dgrad_gn_rows(int n, int d, int h, int w, int cin, int cout, int dtype, int algo) {
  if (algo_base(algo) == MEDNET_ALGO_DIRECT || !tuning_option("gn3_fuse", 1) || !convt_dgrad_mfma_ok(n, d, h, w, cin, cout, dtype)) return 0;
  return ELT_CALL(dtype, convt_dgrad_gn_rows, n, d, h, w, cin, cout, algo_split(algo, dtype));
}
extern "C" int mednet_convt3d_dgrad_gn(const void* dy, const void* packed, void* dx, const void* gn_y, const void* gn_z, int gn_act,
                                       float* gn_partial, int n, int d, int h, int w, int cin, int cout, int dtype, int algo,
                                       mednet_stream stream) {
  int rc = conv_common_checks("convt3d_dgrad_gn", n, d, h, w, cin, cout, 3, dtype, dtype);
  if (rc) return rc;
  MEDNET_REQUIRE(gn_y && gn_z && gn_partial, MEDNET_E_SHAPE, "convt3d_dgrad_gn: gn_y, gn_z and gn_partial are required");
  MEDNET_REQUIRE(mednet_convt3d_dgrad_gn_rows(n, d, h, w, cin, cout, dtype, algo) > 0, MEDNET_E_UNSUPPORTED,
                 "convt3d_dgrad_gn: cin=%d cout=%d dtype=%d not on the matrix-core path", cin, cout, dtype);
  const PackLayout L = pack_layout(cin, cout, 3);
  return ELT_CALL(dtype, launch_convt_dgrad_gn_mfma, dy, (const char*)packed + L.mfma_bwd, dx, gn_y, gn_z, gn_act, gn_partial, n, d,
                  h, w, cin, cout, (hipStream_t)stream, algo_lo_delta(algo, dtype, L));
}

extern "C" size_t mednet_convt3d_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout, int workgroups) {
  const size_t a = wgrad_direct_ws_bytes((size_t)n * d * h * w, cin, cout, 3);
  size_t b = convt_wgrad_mfma_ws_bytes(n, d, h, w, cin, cout, workgroups);
  const size_t f = wgrad_f32_mfma_ws_bytes(n, d, h, w, cin, cout, 1);
  if (f > b) b = f;
  const size_t f3 = conv_x3_supported(cin, cout, 3) ? convt_wgrad_x3_ws_bytes(n, d, h, w, cin, cout, workgroups) : 0;
  if (f3 > b) b = f3;
  return align_up(a > b ? a : b, 256) + channel_sum_ws_bytes(n, (size_t)8 * d * h * w, cout) + 256;
}

extern "C" int mednet_convt3d_wgrad(const void* x, const void* dy, float* dw, float* dbias, int n, int d, int h, int w,
                                    int cin, int cout, int x_dtype, int dy_dtype, int algo, int workgroups, void* ws,
                                    size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(workgroups >= 0, MEDNET_E_SHAPE, "convt3d_wgrad: workgroups %d (0 = one per CU)", workgroups);
  int rc = conv_common_checks("convt3d_wgrad", n, d, h, w, cin, cout, 3, x_dtype, dy_dtype);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (dbias) {
    const size_t need = channel_sum_ws_bytes(n, (size_t)8 * d * h * w, cout);
    MEDNET_REQUIRE(ws_bytes >= need, MEDNET_E_WORKSPACE, "convt3d_wgrad: workspace too small for the bias gradient");
    rc = launch_channel_sum(dy, dbias, n, (size_t)8 * d * h * w, cout, 0, dy_dtype,
                            (char*)ws + (ws_bytes - need) / 256 * 256, need, s);
    if (rc) return rc;
    ws_bytes = (ws_bytes - need) / 256 * 256;
  }
  if (algo_base(algo) != MEDNET_ALGO_DIRECT && conv_f32_mfma_enabled() && x_dtype == MEDNET_F32 && dy_dtype == MEDNET_F32) {
    if (!algo_exact(algo) && conv_x3_enabled() && tuning_option("x3_convt_wgrad", 1) && conv_x3_supported(cin, cout, 3) &&
        conv_x3_fits(2 * d, 2 * h, 2 * w, cout) && conv_x3_fits(d, h, w, cin))
      return launch_convt_wgrad_x3(x, dy, dw, n, d, h, w, cin, cout, ws, ws_bytes, s, workgroups);  // split-bf16, output-parity planes
    return launch_convt_wgrad_f32_mfma(x, dy, dw, n, d, h, w, cin, cout, ws, ws_bytes, s);
  }
  const bool mfma_ok = cin % 32 == 0 && cout % 32 == 0 && is16(x_dtype) && dy_dtype == x_dtype &&
                       wgrad_mfma_fits(n, d, h, w, cin > 8 * cout ? cin : 8 * cout, 1);
  if (algo_base(algo) == MEDNET_ALGO_MFMA && !mfma_ok)
    return fail(MEDNET_E_UNSUPPORTED, "convt3d_wgrad: MFMA path does not take cin=%d cout=%d", cin, cout);
  if (mfma_ok && algo_base(algo) != MEDNET_ALGO_DIRECT)
    return ELT_CALL(x_dtype, launch_convt_wgrad_mfma, x, dy, dw, n, d, h, w, cin, cout, ws, ws_bytes, s, workgroups);
  WgradGeom g;
  g.n = n; g.ad = d; g.ah = h; g.aw = w; g.bd = 2 * d; g.bh = 2 * h; g.bw = 2 * w;
  g.ka = cin; g.kb = cout; g.ks = 3; g.stride2 = 1; g.a_planar = 0; g.b_planar = 0; g.chunk = 0;
  return launch_wgrad_direct(x, dy, dw, g, x_dtype, dy_dtype, ws, ws_bytes, s);
}
