// Internal declarations shared by the convolution translation units.
#pragma once
#include "common.h"

namespace mednet {

enum { MAP_CONV = 0, MAP_CT_FWD = 1, MAP_CT_DG = 2 };

struct ConvGeom {
  int n, od, oh, ow;  // output dims
  int id, ih, iw;     // input dims
  int k, m;           // input channels, output channels
  int ks;             // 3 or 1
  int in_planar, out_planar;
};

struct WgradGeom {
  int n, ad, ah, aw;  // dims of A's grid (the loop grid)
  int bd, bh, bw;     // dims of B's grid
  int ka, kb, ks, stride2;
  int a_planar, b_planar;
  size_t chunk;  // voxels per chunk (filled by the launcher)
};


template <int MAP>
int launch_direct(const void* x, const float* P, const float* bias, const void* skip, void* y, const ConvGeom& g,
                  int x_dtype, int y_dtype, hipStream_t s);
size_t wgrad_direct_ws_bytes(size_t nvox, int ka, int kb, int ks);
int launch_wgrad_direct(const void* A, const void* B, float* dw, WgradGeom g, int a_dtype, int b_dtype, void* ws,
                        size_t ws_bytes, hipStream_t s);
size_t channel_sum_ws_bytes(int n, size_t spatial, int c);
int launch_channel_sum(const void* x, float* out, int n, size_t spatial, int c, int planar, int dtype, void* ws,
                       size_t ws_bytes, hipStream_t s);
bool head_dgrad_supported(int cin, int cout, int ksize, int x_dtype, int x_layout, int y_layout);
int launch_head_dgrad(const void* dy, const float* Pb, void* dz, int n, size_t spatial, int m, int k, int out_dtype,
                      hipStream_t s);
int head_dgrad_gn_rows(size_t spatial, int k, int dtype);
int launch_head_dgrad_gn(const void* dy, const float* Pb, void* dz, const void* gy, const void* gz, int act, float* partial,
                         int n, size_t spatial, int m, int k, int dtype, hipStream_t s);
bool head_vox_supported(int k);
int launch_head_fwd_vox(const void* z, const float* Pb, const float* bias, float* y, int n, size_t spatial, int k, int m,
                        int z_dtype, hipStream_t s);
bool wgrad_1x1_supported(int cin, int cout, int ksize, int x_layout, int dy_layout, int dy_dtype);
size_t wgrad_1x1_ws_bytes(int n, size_t spatial, int cin, int cout);
int launch_wgrad_1x1(const void* z, const void* dy, float* dw, int n, size_t spatial, int cin, int cout, int z_dtype,
                     void* ws, size_t ws_bytes, hipStream_t s);
bool wgrad_c1_supported(int cin, int cout, int ksize, int x_layout, int dy_layout);
size_t wgrad_c1_ws_bytes(int n, int d, int h, int w, int cout);
bool wgrad_c1_gn_supported(int cout, int x_dtype, int dtype);
int launch_wgrad_c1_gn(const void* x, const void* dz, const void* y, const float* coef, const float* bcoef, int act, float* dw,
                       int n, int d, int h, int w, int cout, int x_dtype, int dtype, void* ws, size_t ws_bytes, hipStream_t s);
int launch_wgrad_c1(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cout, int x_dtype,
                    int dy_dtype, void* ws, size_t ws_bytes, hipStream_t s, bool split_bf16 = false);
int launch_pack_f32(const float* w, float* Pf, float* Pb, int cin, int cout, int T, int transposed_src, hipStream_t s);

// loss.hip: the single-workgroup fp64 combines of the per-workgroup loss partials
int launch_hm_finalize(const float* partial, const float* cweight, float* loss, int n, int c, int nblocks, size_t spatial,
                       hipStream_t s);
int launch_dice_finalize(const float* partial, const float* weight, float* loss, float* saved, int c, int nblocks, float eps,
                         hipStream_t s);

// 16-bit matrix-core kernels, conv_mfma.hip (namespace mednet: bf16; api.hip declares the same set in namespace mednet_f16
// for the fp16 build of that file)
#include "conv_mfma_decl.inc"

// fp32 matrix-core kernels (v_mfma_f32_32x32x2_f32), conv_f32_mfma.hip: the parity mode's 3x3x3 family
bool conv_f32_mfma_enabled();
int launch_conv_f32_mfma(const void* x, const float* P, const float* bias, void* y, int n, int d, int h, int w, int k, int m,
                         hipStream_t s);
int launch_convt_dgrad_f32_mfma(const void* dy, const float* Pb, void* dx, int n, int d, int h, int w, int cin, int cout,
                                hipStream_t s);
int launch_convt_fwd_f32_mfma(const void* x, const float* Pf, const float* bias, const void* skip, void* y, int n, int d, int h,
                              int w, int cin, int cout, hipStream_t s);
size_t wgrad_f32_mfma_ws_bytes(int n, int d, int h, int w, int ka, int kb, int stride2);
int launch_wgrad_f32_mfma(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cin, int cout, void* ws,
                          size_t ws_bytes, hipStream_t s);
int launch_convt_wgrad_f32_mfma(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cin, int cout, void* ws,
                                size_t ws_bytes, hipStream_t s);

// split-bf16 matrix-core kernels (three v_mfma_f32_32x32x16_bf16 per product on hi/lo halves), conv_x3_mfma.hip: the parity
// mode's 3x3x3 family at bf16 matrix-core speed.  `sec_hi` = the layer's bf16 fragment image (PackLayout::mfma_fwd / mfma_bwd),
// the low image sits PackLayout::lo_delta bytes behind it (mednet_conv3d_pack_elt(MEDNET_F32)).
bool conv_x3_enabled();
bool conv_x3_supported(int cin, int cout, int ksize);
bool conv_x3_fits(int d, int h, int w, int c);
int conv_x3_stats_rows(int n, int d, int h, int w, int cout);
int launch_conv_x3(const void* x, const void* sec_hi, size_t lo_delta, const float* bias, void* y, int n, int d, int h, int w, int k,
                   int m, float* stats, hipStream_t s);
bool conv_c1_x3_supported(int cin, int cout, int ksize);
int conv_c1_x3_stats_rows(int d, int h, int w);
int launch_conv_c1_x3(const void* x, const float* w_pf, const float* bias, void* y, int n, int d, int h, int w, int cout,
                      float* stats, hipStream_t s);
int launch_conv_x3_dgrad(const void* dy, const void* sec_hi, size_t lo_delta, void* dx, int n, int d, int h, int w, int k, int m,
                         const void* add, const void* gn_y, const float* gn_coef, int gn_act, float* gn_partial, hipStream_t s);
int launch_convt_dgrad_x3(const void* dy, const void* sec_hi, size_t lo_delta, void* dx, int n, int d, int h, int w, int cin,
                          int cout, hipStream_t s);
int launch_convt_fwd_x3(const void* x, const void* sec_hi, size_t lo_delta, const float* bias, const void* skip, void* y, int n,
                        int d, int h, int w, int cin, int cout, hipStream_t s);
size_t wgrad_x3_ws_bytes(int n, int d, int h, int w, int cin, int cout, int workgroups);
bool wgrad_c1_x3_supported(int cout, int x_dtype, int dy_dtype);
int wgrad_c1_x3_blocks(int n, int d, int h, int w);
int launch_wgrad_c1_x3(const void* x, const void* dy, float* part, int n, int d, int h, int w, hipStream_t s);
int launch_wgrad_x3(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cin, int cout, void* ws,
                    size_t ws_bytes, hipStream_t s, int workgroups);

size_t convt_wgrad_x3_ws_bytes(int n, int d, int h, int w, int cin, int cout, int workgroups);
int launch_convt_wgrad_x3(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cin, int cout, void* ws,
                          size_t ws_bytes, hipStream_t s, int workgroups);

}  // namespace mednet
