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

// Byte offsets of the sections of a packed weight buffer (mednet_conv3d_pack).
struct PackLayout {
  int taps;
  size_t f32_fwd, f32_bwd;    // float [T][Cin][Cout], float [T][Cout][Cin] (taps mirrored for Conv3d sources)
  size_t mfma_fwd, mfma_bwd;  // bf16 fragment-ordered images for the MFMA kernels (0 bytes when not applicable)
  size_t mfma_bytes;          // size of ONE mfma section
  size_t total;
};
PackLayout pack_layout(int cin, int cout, int ksize);

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
bool head_vox_supported(int k);
int launch_head_fwd_vox(const void* z, const float* Pb, const float* bias, float* y, int n, size_t spatial, int k, int m,
                        int z_dtype, hipStream_t s);
bool wgrad_1x1_supported(int cin, int cout, int ksize, int x_layout, int dy_layout, int dy_dtype);
size_t wgrad_1x1_ws_bytes(int n, size_t spatial, int cin, int cout);
int launch_wgrad_1x1(const void* z, const void* dy, float* dw, int n, size_t spatial, int cin, int cout, int z_dtype,
                     void* ws, size_t ws_bytes, hipStream_t s);
bool wgrad_c1_supported(int cin, int cout, int ksize, int x_layout, int dy_layout);
size_t wgrad_c1_ws_bytes(int n, int d, int h, int w, int cout);
int launch_wgrad_c1(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cout, int x_dtype,
                    int dy_dtype, void* ws, size_t ws_bytes, hipStream_t s);
bool wgrad_c1_mfma_supported(int cout, int x_dtype, int dy_dtype);
int wgrad_c1_mfma_blocks(int n, int d, int h, int w);
int launch_wgrad_c1_mfma(const void* x, const void* dy, float* part, int n, int d, int h, int w, int cout, hipStream_t s);
int launch_pack_f32(const float* w, float* Pf, float* Pb, int cin, int cout, int T, int transposed_src, hipStream_t s);

// MFMA (bf16 matrix-core) kernels, conv_mfma.hip
bool conv_mfma_fits(int n, int d, int h, int w, int c);
bool conv_mfma_supported(int cin, int cout, int ksize, int x_dtype, int y_dtype, int x_layout, int y_layout, bool bias);
int launch_conv_mfma(const void* x, const void* packed_section, void* y, int n, int d, int h, int w, int cin, int cout,
                     int x_dtype, int y_dtype, float* gn_partial, hipStream_t s, int act = MEDNET_ACT_NONE,
                     const void* add = nullptr);
int launch_conv_mfma_gnb(const void* dy, const void* packed_section, void* dx, int n, int d, int h, int w, int cin, int cout,
                         const void* add, const void* gn_y, const float* gn_coef, int gn_act, float* gn_partial, hipStream_t s);
int conv_mfma_stats_chunks(int n, int d, int h, int w, int cout);
int launch_pack_mfma(const float* w, void* sec_fwd, void* sec_bwd, float* Pf, float* Pb, int cin, int cout, int T,
                     int transposed_src, hipStream_t s);
bool wgrad_mfma_supported(int cin, int cout, int ksize, int x_dtype, int dy_dtype, int x_layout, int dy_layout);
bool wgrad_mfma_fits(int n, int d, int h, int w, int cmax, int scale);
size_t wgrad_mfma_ws_bytes(int n, int d, int h, int w, int cin, int cout, int ksize);
int launch_wgrad_mfma(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cin, int cout, int dtype,
                      void* ws, size_t ws_bytes, hipStream_t s);
bool conv_c1_mfma_supported(int cin, int cout, int ksize, int x_dtype, int y_dtype, int y_layout, bool bias);
int conv_c1_stats_chunks(int d, int h, int w);
int launch_conv_c1_mfma(const void* x, const float* w_pt, void* y, int n, int d, int h, int w, int cout, float* gn_partial,
                        hipStream_t s);
int launch_convt_fwd_mfma(const void* x, const void* sec, const float* bias, const void* skip, void* y, int n, int d,
                          int h, int w, int cin, int cout, hipStream_t s);
int launch_convt_dgrad_mfma(const void* dy, const void* packed_section, void* dx, int n, int d, int h, int w, int cin,
                            int cout, hipStream_t s);
size_t convt_wgrad_mfma_ws_bytes(int n, int d, int h, int w, int cin, int cout);
int launch_convt_wgrad_mfma(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cin, int cout,
                            void* ws, size_t ws_bytes, hipStream_t s);

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

}  // namespace mednet
