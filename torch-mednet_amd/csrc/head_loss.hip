// The 1x1x1 head (model.py:207 `final_conv`) fused with the front end of the Dice loss (loss.py:114-130) -- SURVEY K12.
//
// Unfused, the logits (N x C x D x H x W fp32: 134 MB per class-4 batch of config 2) are written by the head, read by the
// Dice forward, read again by the Dice backward, which writes their gradient, which the head's data gradient and the head's
// weight gradient each read: five passes over planes that carry 4 floats per voxel, next to the 32-channel feature rows
// every one of these kernels streams anyway.  Here
//   forward   head_dice_fwd_kernel   = head_fwd_vox_kernel + dice_fwd_kernel: the logits are written once (the caller's
//             `outputs`) and the softmax terms of the Dice sums are taken from the registers that hold them;
//   backward  head_dice_bwd_kernel   = dice_bwd_kernel + head_dgrad_gn_kernel + wgrad_1x1_kernel + the bias sums: the logit
//             gradient of a voxel lives in registers between its closed form (from the stored logits, the label and the
//             per-channel Dice sums) and its three uses -- dz = W^T dl (stored, with the first pass of the producing block's
//             GroupNorm-3 backward taken from the stored row as before), dW += dl (x) z, db += dl.  No dlogits tensor exists.
// Every per-voxel expression is the one of the kernel it replaces, in the same order, so logits, loss, dz and the
// GroupNorm sums are BIT-IDENTICAL to the unfused launches (tests/test_gpu_ops.py); dW / db are summed in another (fixed)
// order.
#include "common.h"
#include "conv.h"

namespace mednet {

constexpr int HL_BLOCK_VOX = 256 * 8;  // = LOSS_BLOCK_VOX of loss.hip: the partial rows are dice_finalize_kernel's
constexpr int HL_MAXC = 4;              // classes kept in registers (the segmentation heads of the callers: 2 and 4)
constexpr int HL_VPT = 32;              // = HEAD_GN_VPT of conv_direct.hip: the GroupNorm rows are head_dgrad_gn_kernel's

// a voxel's 8-channel piece as it lies in memory (4 registers for the 16-bit types): converted where it is used, so rows in
// flight do not hold 8 registers each
template <typename T>
struct Raw8;
template <>
struct Raw8<bf16> {
  bf16x8 r;
  __device__ __forceinline__ void load(const bf16* p, size_t i) { r = *reinterpret_cast<const bf16x8*>(p + i); }
  __device__ __forceinline__ float at(int j) const { return (float)r[j]; }
};
template <>
struct Raw8<f16> {
  f16x8 r;
  __device__ __forceinline__ void load(const f16* p, size_t i) { r = *reinterpret_cast<const f16x8*>(p + i); }
  __device__ __forceinline__ float at(int j) const { return (float)r[j]; }
};
template <>
struct Raw8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p, size_t i) {
    a = *reinterpret_cast<const f32x4*>(p + i);
    b = *reinterpret_cast<const f32x4*>(p + i + 4);
  }
  __device__ __forceinline__ float at(int j) const { return j < 4 ? a[j] : b[j - 4]; }
};

template <typename TL>
__device__ __forceinline__ int label_at(const TL* __restrict__ lab, size_t i) { return (int)lab[i]; }

// softmax / sigmoid of the class logits of one voxel (loss.hip probs_of, on registers)
__device__ __forceinline__ void probs_reg(const float* lg, int c, int sigmoid, float* p) {
  if (sigmoid) {
#pragma unroll
    for (int k = 0; k < HL_MAXC; ++k)
      if (k < c) p[k] = 1.f / (1.f + expf(-lg[k]));
  } else {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < HL_MAXC; ++k)
      if (k < c) {
        p[k] = lg[k];
        mx = fmaxf(mx, p[k]);
      }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < HL_MAXC; ++k)
      if (k < c) {
        p[k] = expf(p[k] - mx);
        den += p[k];
      }
    const float inv = 1.f / den;
#pragma unroll
    for (int k = 0; k < HL_MAXC; ++k)
      if (k < c) p[k] *= inv;
  }
}

// ---- forward: one voxel per lane, weights wave-uniform (SGPR operands), 8 voxels per thread ------------------------------
// logits[n][i][v] = bias[i] + sum_k z[n][v][k] W[i][k];  partial[n][block][c][2] = {sum p t mask, sum (p + t) mask}
template <typename TI, int K, typename TL>
__global__ __launch_bounds__(256) void head_dice_fwd_kernel(const TI* __restrict__ z, const float* __restrict__ Pb /*[m][K]*/,
                                                            const float* __restrict__ bias, const TL* __restrict__ lab,
                                                            int64_t lab_sn, float* __restrict__ y, float* __restrict__ partial,
                                                            size_t spatial, int m, int sigmoid, int ignore) {
  __shared__ float scratch[4];
  const int n = blockIdx.y;
  float I[HL_MAXC], D[HL_MAXC];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < HL_MAXC; ++k) I[k] = D[k] = 0.f;
  const size_t v0 = (size_t)blockIdx.x * HL_BLOCK_VOX;
  for (int it = 0; it < 8; ++it) {
    const size_t v = v0 + (size_t)it * 256 + threadIdx.x;
    if (v < spatial) {
      float zv[K];
#pragma unroll
      for (int q = 0; q < K / 8; ++q) {
        const F8 t = ld8(z, ((size_t)n * spatial + v) * K + q * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) zv[q * 8 + j] = t.v[j];
      }
      const int yl = label_at(lab, (size_t)n * lab_sn + v);
      float lg[HL_MAXC], p[HL_MAXC];
#pragma unroll
      for (int i = 0; i < HL_MAXC; ++i) {
        lg[i] = 0.f;
        if (i < m) {  // (head_fwd_vox_kernel's two chains: the logits are bit for bit the unfused head's)
          float s0 = bias ? bias[i] : 0.f, s1 = 0.f;
#pragma unroll
          for (int j = 0; j < K; j += 2) {
            s0 = fmaf(zv[j], Pb[(size_t)i * K + j], s0);
            s1 = fmaf(zv[j + 1], Pb[(size_t)i * K + j + 1], s1);
          }
          lg[i] = s0 + s1;
          y[((size_t)n * m + i) * spatial + v] = lg[i];
        }
      }
      probs_reg(lg, m, sigmoid, p);
      bad |= (unsigned)yl >= (unsigned)m;  // (loss.hip dice_fwd_kernel: an out-of-range label poisons the loss)
#pragma unroll
      for (int k = 0; k < HL_MAXC; ++k)
        if (k < m) {
          const float t = (k == yl) ? 1.f : 0.f;
          const float mk = (ignore != MEDNET_NO_IGNORE && t == (float)ignore) ? 0.f : 1.f;
          I[k] = fmaf(p[k] * mk, t * mk, I[k]);
          D[k] += (p[k] + t) * mk;
        }
    }
  }
  if (bad) I[0] = D[0] = __builtin_nanf("");
  float* out = partial + ((size_t)n * gridDim.x + blockIdx.x) * m * 2;
#pragma unroll
  for (int k = 0; k < HL_MAXC; ++k)
    if (k < m) {
      const float a = block_sum<4>(I[k], scratch);
      const float b = block_sum<4>(D[k], scratch);
      if (threadIdx.x == 0) {
        out[2 * k] = a;
        out[2 * k + 1] = b;
      }
    }
}

// ---- backward: K/8 lanes per voxel, each owns 8 channels of the voxel's row; two voxels per trip --------------------------
// wpart[n][block][wave][m * K + m]: this wave's partial of dW (row-major [class][channel]) and, behind it, of db
// FOLD: z is the output of a fused conv -> activation layer and there are no GroupNorm sums to take (UNet3D's last block): the
// stored gradient carries act'(z).  Its own instantiation: as a runtime branch it cost the main form its third wave per SIMD.
template <typename TO, int K, typename TL, bool FOLD = false>
__global__ __launch_bounds__(256, (sizeof(TO) == 2 && !FOLD) ? 3 : 2) void head_dice_bwd_kernel(const float* __restrict__ lgs, const TL* __restrict__ lab, int64_t lab_sn,
                                                            const float* __restrict__ Pb /*[m][K]*/, const float* __restrict__ weight,
                                                            const float* __restrict__ saved, const float* __restrict__ dloss,
                                                            float eps, int sigmoid, int ignore, TO* __restrict__ dz,
                                                            const TO* __restrict__ gy /*nullable*/, const TO* __restrict__ gz, int act,
                                                            float* __restrict__ gn_partial /*nullable*/, float* __restrict__ wpart,
                                                            size_t spatial, int m) {
  constexpr int CG = K / 8, VPW = 256 / CG;
  const int n = blockIdx.y;
  const int cgi = threadIdx.x % CG, vi = threadIdx.x / CG;
  float wreg[HL_MAXC][8];
#pragma unroll
  for (int i = 0; i < HL_MAXC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) wreg[i][j] = i < m ? Pb[(size_t)i * K + cgi * 8 + j] : 0.f;
  // per-class coefficients of the closed form (loss.hip dice_bwd_kernel)
  const float go = *dloss;
  float gI[HL_MAXC], gD[HL_MAXC];
#pragma unroll
  for (int k = 0; k < HL_MAXC; ++k) {
    gI[k] = gD[k] = 0.f;
    if (k < m) {
      const float w = weight ? weight[k] : 1.f;
      const float I = saved[2 * k], D = saved[2 * k + 1];
      const float Dc = fmaxf(D, eps);
      gI[k] = -2.f * w / ((float)m * Dc) * go + ((I != I || D != D) ? __builtin_nanf("") : 0.f);
      gD[k] = (D >= eps ? 2.f * w * I / ((float)m * Dc * Dc) : 0.f) * go;
    }
  }
  auto dlogits_of = [&](const float* lg, int yl, float* dl) {  // the logit gradient of one voxel
    float p[HL_MAXC], g[HL_MAXC];
    probs_reg(lg, m, sigmoid, p);
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < HL_MAXC; ++k)
      if (k < m) {
        const float t = (k == yl) ? 1.f : 0.f;
        const float mk = (ignore != MEDNET_NO_IGNORE && t == (float)ignore) ? 0.f : 1.f;
        g[k] = mk * (gI[k] * t * mk + gD[k]);
        dot = fmaf(p[k], g[k], dot);
      }
#pragma unroll
    for (int k = 0; k < HL_MAXC; ++k) dl[k] = k < m ? (sigmoid ? g[k] * p[k] * (1.f - p[k]) : p[k] * (g[k] - dot)) : 0.f;
  };
  float ss[8], sq[8], wacc[HL_MAXC][8], bacc[HL_MAXC];
#pragma unroll
  for (int j = 0; j < 8; ++j) ss[j] = sq[j] = 0.f;
#pragma unroll
  for (int i = 0; i < HL_MAXC; ++i) {
    bacc[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) wacc[i][j] = 0.f;
  }
  const size_t v0 = (size_t)blockIdx.x * (VPW * HL_VPT) + vi;
  for (int it = 0; it < HL_VPT; it += 2) {
    const size_t v = v0 + (size_t)it * VPW;
    if (v >= spatial) break;
    const size_t vb = v + VPW;
    const bool hb = (it + 1 < HL_VPT) && vb < spatial;
    const size_t vbs = hb ? vb : v;
    // every load of both voxels before the first use
    float la[HL_MAXC], lb[HL_MAXC];
    if constexpr (CG == 4) {
      // the lanes of a voxel are one DPP quad (32 features; with 64 the extra registers spilled): lane q of the quad loads class q for both voxels and the quad hands the
      // values round -- 2 planar loads per lane and trip instead of 8 (each a whole wave instruction for 64 useful bytes)
      const int cq = threadIdx.x & 3;
      const float oa = cq < m ? lgs[((size_t)n * m + cq) * spatial + v] : 0.f;
      const float ob = cq < m ? lgs[((size_t)n * m + cq) * spatial + vbs] : 0.f;
      la[0] = dpp_f32<(0) | (0 << 2) | (0 << 4) | (0 << 6)>(oa);
      la[1] = dpp_f32<(1) | (1 << 2) | (1 << 4) | (1 << 6)>(oa);
      la[2] = dpp_f32<(2) | (2 << 2) | (2 << 4) | (2 << 6)>(oa);
      la[3] = dpp_f32<(3) | (3 << 2) | (3 << 4) | (3 << 6)>(oa);
      lb[0] = dpp_f32<(0) | (0 << 2) | (0 << 4) | (0 << 6)>(ob);
      lb[1] = dpp_f32<(1) | (1 << 2) | (1 << 4) | (1 << 6)>(ob);
      lb[2] = dpp_f32<(2) | (2 << 2) | (2 << 4) | (2 << 6)>(ob);
      lb[3] = dpp_f32<(3) | (3 << 2) | (3 << 4) | (3 << 6)>(ob);
      static_assert(HL_MAXC == 4, "one class per lane of a quad");
    } else {
#pragma unroll
      for (int i = 0; i < HL_MAXC; ++i) {
        la[i] = i < m ? lgs[((size_t)n * m + i) * spatial + v] : 0.f;
        lb[i] = i < m ? lgs[((size_t)n * m + i) * spatial + vbs] : 0.f;
      }
    }
    const int ya = label_at(lab, (size_t)n * lab_sn + v), yb = label_at(lab, (size_t)n * lab_sn + vbs);
    const size_t rowa = ((size_t)n * spatial + v) * K + cgi * 8, rowb = ((size_t)n * spatial + vbs) * K + cgi * 8;
    Raw8<TO> zva, zvb, yva, yvb;
    zva.load(gz, rowa);
    zvb.load(gz, rowb);
    if (gy) {
      yva.load(gy, rowa);
      yvb.load(gy, rowb);
    } else {
      yva = zva;
      yvb = zvb;
    }
    float da[HL_MAXC], db[HL_MAXC];
    dlogits_of(la, ya, da);
    dlogits_of(lb, yb, db);
    if (!hb) {
#pragma unroll
      for (int i = 0; i < HL_MAXC; ++i) db[i] = 0.f;
    }
    F8 t, u;
#pragma unroll
    for (int j = 0; j < 8; ++j) t.v[j] = u.v[j] = 0.f;
#pragma unroll
    for (int i = 0; i < HL_MAXC; ++i) {
      if (i < m) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          t.v[j] = fmaf(da[i], wreg[i][j], t.v[j]);
          u.v[j] = fmaf(db[i], wreg[i][j], u.v[j]);
          wacc[i][j] = fmaf(db[i], zvb.at(j), fmaf(da[i], zva.at(j), wacc[i][j]));  // dW[i][k] += dl_i * z_k
        }
        bacc[i] += da[i] + db[i];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // the stored value is what GroupNorm-3's second pass reads
      t.v[j] = (float)(TO)t.v[j];
      u.v[j] = (float)(TO)u.v[j];
    }
    if constexpr (FOLD) {
      // x is the OUTPUT of a fused conv -> activation layer (UNet3D's last block, components.py:57-63): its derivative is folded
      // into the stored gradient -- on the rounded value, as mednet_act_bwd would read it -- and that layer's backward skips its
      // activation pass (ops.ActMaskHook)
      float za[8], zb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        za[j] = zva.at(j);
        zb[j] = zvb.at(j);
      }
      act_grad_n<8>(t.v, za, act);
      act_grad_n<8>(u.v, zb, act);
    }
    st8(dz, rowa, t);
    if (hb) st8(dz, rowb, u);
    if (!FOLD && gy) {
      float za[8], zb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        za[j] = zva.at(j);
        zb[j] = zvb.at(j);
      }
      act_grad_n<8>(t.v, za, act);
      act_grad_n<8>(u.v, zb, act);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ss[j] += t.v[j] + u.v[j];
        sq[j] = fmaf(u.v[j], yvb.at(j), fmaf(t.v[j], yva.at(j), sq[j]));
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wrow = ((size_t)n * gridDim.x + blockIdx.x) * 4 + wave;
  if (gn_partial) {
    float* out = gn_partial + wrow * K * 2 + cgi * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = lane_class_sum<CG>(ss[j]), b = lane_class_sum<CG>(sq[j]);
      if (lane < CG) {
        out[2 * j] = a;
        out[2 * j + 1] = b;
      }
    }
  }
  float* wout = wpart + wrow * ((size_t)m * K + m);
#pragma unroll
  for (int i = 0; i < HL_MAXC; ++i) {
    if (i < m) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = lane_class_sum<CG>(wacc[i][j]);
        if (lane < CG) wout[(size_t)i * K + cgi * 8 + j] = a;
      }
      const float b = lane_class_sum<CG>(bacc[i]);  // (every lane of a voxel added the same dl: class cgi == 0 holds the sum once)
      if (lane == 0) wout[(size_t)m * K + i] = b;
    }
  }
}

// dW[e] (e < m * K) and db[e - m * K] from the waves' partial rows, in two coalesced stages (one wave per output walking 16 384
// rows with a 528-byte stride took 60 us): stage 1, workgroup g sums its slice of the rows, thread e one output (consecutive
// threads read consecutive floats of a row), fp64, into part2[g][e]; stage 2 sums the HL_WF_GROUPS partials in order.
constexpr int HL_WF_GROUPS = 128;
__global__ __launch_bounds__(192) void head_dice_wfinal1_kernel(const float* __restrict__ wpart, double* __restrict__ part2, int rows,
                                                                int width) {
  const int per = (rows + HL_WF_GROUPS - 1) / HL_WF_GROUPS;
  const int r0 = blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  for (int e = threadIdx.x; e < width; e += 192) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
      s0 += (double)wpart[(size_t)r * width + e];
      s1 += (double)wpart[(size_t)(r + 1) * width + e];
      s2 += (double)wpart[(size_t)(r + 2) * width + e];
      s3 += (double)wpart[(size_t)(r + 3) * width + e];
    }
    for (; r < r1; ++r) s0 += (double)wpart[(size_t)r * width + e];
    part2[(size_t)blockIdx.x * width + e] = (s0 + s1) + (s2 + s3);
  }
}
__global__ __launch_bounds__(192) void head_dice_wfinal2_kernel(const double* __restrict__ part2, float* __restrict__ dw,
                                                                float* __restrict__ dbias, int width, int mk) {
  for (int e = threadIdx.x; e < width; e += 192) {
    double s = 0.0;
    for (int g = 0; g < HL_WF_GROUPS; ++g) s += part2[(size_t)g * width + e];
    if (e < mk) dw[e] = (float)s;
    else if (dbias) dbias[e - mk] = (float)s;
  }
}

// (defined in loss.hip)
__global__ void dice_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ weight, float* __restrict__ loss,
                                     float* __restrict__ saved, float* __restrict__ dice_out, int c, int nblocks, float eps);

static inline unsigned hl_fwd_blocks(size_t spatial) { return (unsigned)((spatial + HL_BLOCK_VOX - 1) / HL_BLOCK_VOX); }
static inline unsigned hl_bwd_blocks(size_t spatial, int k) {
  const size_t per_wg = (size_t)(256 / (k / 8)) * HL_VPT;
  return (unsigned)((spatial + per_wg - 1) / per_wg);
}

}  // namespace mednet

using namespace mednet;

extern "C" int mednet_head_dice_supported(int cin, int cout, int dtype, int label_dtype) {
  return (cin == 16 || cin == 32 || cin == 64) && cout >= 1 && cout <= HL_MAXC && dtype_ok(dtype) &&
         (label_dtype == MEDNET_U8 || label_dtype == MEDNET_I64) && tuning_option("head_loss_fuse", 1);
}
extern "C" size_t mednet_head_dice_ws_bytes(int n, size_t spatial, int cin, int cout) {
  const size_t fwd = ((size_t)n * hl_fwd_blocks(spatial) * cout * 2 + 64) * sizeof(float);
  const size_t width = (size_t)cout * cin + cout;
  const size_t bwd = ((size_t)n * hl_bwd_blocks(spatial, cin) * 4 * width + 64) * sizeof(float) + (size_t)HL_WF_GROUPS * width * sizeof(double) + 16;
  return fwd > bwd ? fwd : bwd;
}
extern "C" int mednet_head_dice_gn_rows(int n, size_t spatial, int cin) {
  (void)n;
  return 4 * (int)hl_bwd_blocks(spatial, cin);
}

extern "C" int mednet_head_dice_fwd(const void* z, const void* packed, const float* bias, const void* labels, int label_dtype,
                                    int64_t label_stride_n, const float* weight, float* logits, float* loss, float* saved, int n,
                                    size_t spatial, int cin, int cout, float eps, int sigmoid, int ignore_index, int z_dtype,
                                    void* ws, size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(mednet_head_dice_supported(cin, cout, z_dtype, label_dtype), MEDNET_E_UNSUPPORTED,
                 "head_dice_fwd: %d -> %d classes, dtype %d, labels %d", cin, cout, z_dtype, label_dtype);
  MEDNET_REQUIRE(n > 0 && spatial > 0 && z && packed && labels && logits && loss && saved, MEDNET_E_SHAPE, "head_dice_fwd: bad arguments");
  MEDNET_REQUIRE(ws_bytes >= mednet_head_dice_ws_bytes(n, spatial, cin, cout), MEDNET_E_WORKSPACE, "head_dice_fwd: workspace too small");
  const PackLayout L = pack_layout(cin, cout, 1);
  const float* Pb = (const float*)((const char*)packed + L.f32_bwd);  // Pb[t = 0][co][ci] = W[co][ci]
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = hl_fwd_blocks(spatial);
  float* partial = (float*)ws;
  const dim3 grid(nb, n);
#define HF(TI_, K_, TL_) hipLaunchKernelGGL((head_dice_fwd_kernel<TI_, K_, TL_>), grid, dim3(256), 0, s, (const TI_*)z, Pb, bias, (const TL_*)labels, label_stride_n, logits, partial, spatial, cout, sigmoid, ignore_index)
#define HF_L(TI_, K_) do { if (label_dtype == MEDNET_U8) HF(TI_, K_, uint8_t); else HF(TI_, K_, int64_t); } while (0)
#define HF_K(TI_) do { if (cin == 16) HF_L(TI_, 16); else if (cin == 32) HF_L(TI_, 32); else HF_L(TI_, 64); } while (0)
  if (z_dtype == MEDNET_F32) HF_K(float);
  else if (z_dtype == MEDNET_BF16) HF_K(bf16);
  else HF_K(f16);
#undef HF_K
#undef HF_L
#undef HF
  int rc = check_launch("head_dice_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, s, partial, weight, loss, saved, (float*)nullptr, cout, (int)(nb * n), eps);
  return check_launch("dice_finalize");
}

extern "C" int mednet_head_dice_bwd(const float* logits, const void* labels, int label_dtype, int64_t label_stride_n,
                                    const void* packed, const float* weight, const float* saved, const float* dloss, void* dz,
                                    const void* gn_y, const void* z, int gn_act, float* gn_partial, float* dw, float* dbias, int n,
                                    size_t spatial, int cin, int cout, float eps, int sigmoid, int ignore_index, int z_dtype,
                                    void* ws, size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(mednet_head_dice_supported(cin, cout, z_dtype, label_dtype), MEDNET_E_UNSUPPORTED,
                 "head_dice_bwd: %d -> %d classes, dtype %d, labels %d", cin, cout, z_dtype, label_dtype);
  MEDNET_REQUIRE(n > 0 && spatial > 0 && logits && labels && packed && saved && dloss && dz && z && dw, MEDNET_E_SHAPE, "head_dice_bwd: bad arguments");
  MEDNET_REQUIRE((gn_y == nullptr) == (gn_partial == nullptr), MEDNET_E_SHAPE, "head_dice_bwd: gn_y and gn_partial go together");
  MEDNET_REQUIRE(ws_bytes >= mednet_head_dice_ws_bytes(n, spatial, cin, cout), MEDNET_E_WORKSPACE, "head_dice_bwd: workspace too small");
  const PackLayout L = pack_layout(cin, cout, 1);
  const float* Pb = (const float*)((const char*)packed + L.f32_bwd);
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = hl_bwd_blocks(spatial, cin);
  float* wpart = (float*)ws;
  const dim3 grid(nb, n);
  const bool fold = gn_y == nullptr && gn_act != MEDNET_ACT_NONE;
#define HB_(TO_, K_, TL_, F_) hipLaunchKernelGGL((head_dice_bwd_kernel<TO_, K_, TL_, F_>), grid, dim3(256), 0, s, logits, (const TL_*)labels, label_stride_n, Pb, weight, saved, dloss, eps, sigmoid, ignore_index, (TO_*)dz, (const TO_*)gn_y, (const TO_*)z, gn_act, gn_partial, wpart, spatial, cout)
#define HB(TO_, K_, TL_) do { if (fold) HB_(TO_, K_, TL_, true); else HB_(TO_, K_, TL_, false); } while (0)
#define HB_L(TO_, K_) do { if (label_dtype == MEDNET_U8) HB(TO_, K_, uint8_t); else HB(TO_, K_, int64_t); } while (0)
#define HB_K(TO_) do { if (cin == 16) HB_L(TO_, 16); else if (cin == 32) HB_L(TO_, 32); else HB_L(TO_, 64); } while (0)
  if (z_dtype == MEDNET_F32) HB_K(float);
  else if (z_dtype == MEDNET_BF16) HB_K(bf16);
  else HB_K(f16);
#undef HB_K
#undef HB_L
#undef HB_
#undef HB
  int rc = check_launch("head_dice_bwd");
  if (rc) return rc;
  const int width = cout * cin + cout, rows = (int)(n * nb * 4);
  double* part2 = (double*)((char*)ws + (((size_t)rows * width + 64) * sizeof(float) + 7) / 8 * 8);
  hipLaunchKernelGGL(head_dice_wfinal1_kernel, dim3(HL_WF_GROUPS), dim3(192), 0, s, wpart, part2, rows, width);
  hipLaunchKernelGGL(head_dice_wfinal2_kernel, dim3(1), dim3(192), 0, s, part2, dw, dbias, width, cout * cin);
  return check_launch("head_dice_wfinal");
}
