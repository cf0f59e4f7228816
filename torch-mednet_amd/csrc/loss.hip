// Fused per-voxel losses: soft Dice, weighted cross-entropy, weighted heat-map regression.
//
// The reference builds Dice from ~8 full-size intermediates (softmax, one-hot, two permuted copies, product, sums:
// loss.py:116,81-86,10-21,43-47).  Here forward is ONE pass over logits+labels that keeps the per-channel sums in
// registers, and backward is ONE pass using the closed form
//     dL/dp[c,v] = mask * ( -2 w_c t / (C D'_c) + 2 w_c I_c [D_c >= eps] / (C D'_c^2) ),  D' = max(D, eps)
//     softmax: dL/dz_c = p_c (g_c - sum_k p_k g_k)      sigmoid: dL/dz_c = g_c p_c (1 - p_c)
// Logits are planar fp32 (NCDHW or a channel slice of it: element strides stride_n/stride_c, unit voxel stride), so a
// wave reads 64 consecutive voxels of each channel plane: C coalesced 256-B rows per wave-instruction group.
#include <limits.h>
#include "common.h"

namespace mednet {

constexpr int LOSS_BLOCK_VOX = 256 * 8;  // voxels per workgroup

static inline unsigned loss_blocks(size_t spatial) { return (unsigned)((spatial + LOSS_BLOCK_VOX - 1) / LOSS_BLOCK_VOX); }

template <int MAXC>
__device__ __forceinline__ void probs_of(const float* __restrict__ lg, size_t base, int64_t sc, int c, int sigmoid,
                                         float* p) {
  if (sigmoid) {
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
      if (k < c) p[k] = 1.f / (1.f + expf(-lg[base + (size_t)k * sc]));
  } else {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
      if (k < c) {
        p[k] = lg[base + (size_t)k * sc];
        mx = fmaxf(mx, p[k]);
      }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
      if (k < c) {
        p[k] = expf(p[k] - mx);
        den += p[k];
      }
    const float inv = 1.f / den;
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
      if (k < c) p[k] *= inv;
  }
}

// partial[n][block][c][2] = {sum p*t*mask, sum (p+t)*mask}
template <int MAXC, typename TL>
__global__ __launch_bounds__(256) void dice_fwd_kernel(const float* __restrict__ lg, const TL* __restrict__ lab, int64_t lab_sn,
                                                       float* __restrict__ partial, int c, size_t spatial, int64_t sn,
                                                       int64_t sc, int sigmoid, int ignore) {
  __shared__ float scratch[4];
  const int n = blockIdx.y;
  float I[MAXC], D[MAXC];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) I[k] = D[k] = 0.f;
  const size_t v0 = (size_t)blockIdx.x * LOSS_BLOCK_VOX;
  for (int it = 0; it < 8; ++it) {
    const size_t v = v0 + (size_t)it * 256 + threadIdx.x;
    if (v < spatial) {
      float p[MAXC];
      probs_of<MAXC>(lg, (size_t)n * sn + v, sc, c, sigmoid, p);
      const int y = (int)lab[(size_t)n * lab_sn + v];
      // A label outside [0, C) makes the reference raise (scatter_ index error, loss.py:81-86).  Raising from a kernel
      // would cost a host sync per step; instead the loss (and with it every gradient) becomes NaN: loud, not silent.
      bad |= (unsigned)y >= (unsigned)c;
#pragma unroll
      for (int k = 0; k < MAXC; ++k)
        if (k < c) {
          const float t = (k == y) ? 1.f : 0.f;
          // loss.py:31-36: the mask is computed on the ONE-HOT target (values 0/1), not on the label image
          const float m = (ignore != MEDNET_NO_IGNORE && t == (float)ignore) ? 0.f : 1.f;
          I[k] = fmaf(p[k] * m, t * m, I[k]);
          D[k] += (p[k] + t) * m;
        }
    }
  }
  if (bad) I[0] = D[0] = __builtin_nanf("");
  float* out = partial + ((size_t)n * gridDim.x + blockIdx.x) * c * 2;
#pragma unroll
  for (int k = 0; k < MAXC; ++k)
    if (k < c) {
      const float a = block_sum<4>(I[k], scratch);
      const float b = block_sum<4>(D[k], scratch);
      if (threadIdx.x == 0) {
        out[2 * k] = a;
        out[2 * k + 1] = b;
      }
    }
}

// single workgroup: fp64 fixed-order combine, then the scalar
__global__ __launch_bounds__(256) void dice_finalize_kernel(const float* __restrict__ partial,
                                                            const float* __restrict__ weight, float* __restrict__ loss,
                                                            float* __restrict__ saved, float* __restrict__ dice_out,
                                                            int c, int nblocks, float eps) {
  __shared__ double sh[2][256];
  __shared__ double dice_s[64];
  for (int k = 0; k < c; ++k) {
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
      a += (double)partial[((size_t)i * c + k) * 2];
      b += (double)partial[((size_t)i * c + k) * 2 + 1];
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) {
        sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
        sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float I = (float)sh[0][0], D = (float)sh[1][0];
      saved[2 * k] = I;
      saved[2 * k + 1] = D;
      const float wI = weight ? weight[k] * I : I;
      const float dice = 2.f * wI / fmaxf(D, eps);
      if (dice_out) dice_out[k] = dice;
      dice_s[k] = (double)(1.f - dice);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && loss) {
    float s = 0.f;
    for (int k = 0; k < c; ++k) s += (float)dice_s[k];
    *loss = s / (float)c;
  }
}

template <int MAXC, typename TL>
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ lg, const TL* __restrict__ lab, int64_t lab_sn,
                                                       const float* __restrict__ weight,
                                                       const float* __restrict__ saved, const float* __restrict__ dloss,
                                                       float* __restrict__ dlg, int c, size_t spatial, int64_t sn,
                                                       int64_t sc, float eps, int sigmoid, int ignore) {
  const int n = blockIdx.y;
  const float go = *dloss;
  float gI[MAXC], gD[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k)
    if (k < c) {
      const float w = weight ? weight[k] : 1.f;
      const float I = saved[2 * k], D = saved[2 * k + 1];
      const float Dc = fmaxf(D, eps);  // (fmaxf drops a NaN: the poison of an out-of-range label is re-applied below)
      gI[k] = -2.f * w / ((float)c * Dc) * go + ((I != I || D != D) ? __builtin_nanf("") : 0.f);
      gD[k] = (D >= eps ? 2.f * w * I / ((float)c * Dc * Dc) : 0.f) * go;
    }
  const size_t v0 = (size_t)blockIdx.x * LOSS_BLOCK_VOX;
  for (int it = 0; it < 8; ++it) {
    const size_t v = v0 + (size_t)it * 256 + threadIdx.x;
    if (v < spatial) {
      float p[MAXC], g[MAXC];
      probs_of<MAXC>(lg, (size_t)n * sn + v, sc, c, sigmoid, p);
      const int y = (int)lab[(size_t)n * lab_sn + v];
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < MAXC; ++k)
        if (k < c) {
          const float t = (k == y) ? 1.f : 0.f;
          const float m = (ignore != MEDNET_NO_IGNORE && t == (float)ignore) ? 0.f : 1.f;
          g[k] = m * (gI[k] * t * m + gD[k]);
          dot = fmaf(p[k], g[k], dot);
        }
#pragma unroll
      for (int k = 0; k < MAXC; ++k)
        if (k < c) {
          const float dz = sigmoid ? g[k] * p[k] * (1.f - p[k]) : p[k] * (g[k] - dot);
          dlg[(size_t)n * sn + (size_t)k * sc + v] = dz;
        }
    }
  }
}

// ---- weighted cross-entropy: partial[n][block][2] = {sum w_y * nll, sum w_y} -------------------------------------
template <int MAXC>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ lg, const int64_t* __restrict__ lab,
                                                     const float* __restrict__ weight, float* __restrict__ partial, int c,
                                                     size_t spatial, int64_t sn, int64_t sc, int ignore) {
  __shared__ float scratch[4];
  const int n = blockIdx.y;
  float num = 0.f, den = 0.f;
  const size_t v0 = (size_t)blockIdx.x * LOSS_BLOCK_VOX;
  for (int it = 0; it < 8; ++it) {
    const size_t v = v0 + (size_t)it * 256 + threadIdx.x;
    if (v < spatial) {
      const int y = (int)lab[(size_t)n * spatial + v];
      // nll_loss raises for a class index outside [0, C) that is not ignore_index; here the loss becomes NaN (see dice_fwd)
      if (y != ignore && (unsigned)y >= (unsigned)c) num = __builtin_nanf("");
      if (y != ignore && y >= 0 && y < c) {
        const size_t base = (size_t)n * sn + v;
        float mx = -INFINITY, zy = 0.f;
#pragma unroll
        for (int k = 0; k < MAXC; ++k)
          if (k < c) {
            const float z = lg[base + (size_t)k * sc];
            mx = fmaxf(mx, z);
            if (k == y) zy = z;
          }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXC; ++k)
          if (k < c) s += expf(lg[base + (size_t)k * sc] - mx);
        const float w = weight ? weight[y] : 1.f;
        num = fmaf(w, (mx + logf(s)) - zy, num);
        den += w;
      }
    }
  }
  num = block_sum<4>(num, scratch);
  den = block_sum<4>(den, scratch);
  if (threadIdx.x == 0) {
    float* o = partial + ((size_t)n * gridDim.x + blockIdx.x) * 2;
    o[0] = num;
    o[1] = den;
  }
}
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ partial, float* __restrict__ loss,
                                                          float* __restrict__ saved, int nblocks) {
  __shared__ double sh[2][256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    a += (double)partial[2 * (size_t)i];
    b += (double)partial[2 * (size_t)i + 1];
  }
  sh[0][threadIdx.x] = a;
  sh[1][threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    saved[0] = (float)sh[1][0];
    *loss = (float)(sh[0][0] / sh[1][0]);
  }
}
template <int MAXC>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ lg, const int64_t* __restrict__ lab,
                                                     const float* __restrict__ weight, const float* __restrict__ saved,
                                                     const float* __restrict__ dloss, float* __restrict__ dlg, int c,
                                                     size_t spatial, int64_t sn, int64_t sc, int ignore) {
  const int n = blockIdx.y;
  const float scale = *dloss / saved[0];
  const size_t v0 = (size_t)blockIdx.x * LOSS_BLOCK_VOX;
  for (int it = 0; it < 8; ++it) {
    const size_t v = v0 + (size_t)it * 256 + threadIdx.x;
    if (v < spatial) {
      const int y = (int)lab[(size_t)n * spatial + v];
      const bool live = (y != ignore && y >= 0 && y < c);
      float p[MAXC];
      probs_of<MAXC>(lg, (size_t)n * sn + v, sc, c, 0, p);
      const float w = live ? (weight ? weight[y] : 1.f) * scale : 0.f;
#pragma unroll
      for (int k = 0; k < MAXC; ++k)
        if (k < c) dlg[(size_t)n * sn + (size_t)k * sc + v] = w * (p[k] - (k == y ? 1.f : 0.f));
    }
  }
}

// ---- heat-map regression: partial[(n*c + ch)][block] = sum f(out - tgt) -----------------------------------------
template <typename TT>
__global__ __launch_bounds__(256) void hm_fwd_kernel(const float* __restrict__ out, const TT* __restrict__ tgt, int64_t tgt_sn,
                                                     float* __restrict__ partial, int c, size_t spatial, int64_t sn,
                                                     int64_t sc, int kind) {
  __shared__ float scratch[4];
  const int ch = blockIdx.y, n = blockIdx.z;
  float s = 0.f;
  const size_t v0 = (size_t)blockIdx.x * LOSS_BLOCK_VOX;
  for (int it = 0; it < 8; ++it) {
    const size_t v = v0 + (size_t)it * 256 + threadIdx.x;
    if (v < spatial) {
      const float d = out[(size_t)n * sn + (size_t)ch * sc + v] - ld(tgt, (size_t)n * tgt_sn + (size_t)ch * spatial + v);
      s += kind == MEDNET_REG_L2 ? d * d : fabsf(d);
    }
  }
  s = block_sum<4>(s, scratch);
  if (threadIdx.x == 0) partial[((size_t)n * c + ch) * gridDim.x + blockIdx.x] = s;
}
// loss = sum_c w_c * (sum over n, blocks) / (N * spatial)   -- landmarks.py:129-132 evaluates channel by channel
// (16 waves, a wave per channel, four loads in flight per lane, one barrier: as one 256-thread loop over the channels with an LDS
//  tree per channel this was 111 us for the 16 heat maps)
__global__ __launch_bounds__(1024) void hm_finalize_kernel(const float* __restrict__ partial,
                                                           const float* __restrict__ cweight, float* __restrict__ loss,
                                                           int n, int c, int nblocks, double count) {
  __shared__ double chsum[256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int ch0 = 0; ch0 < c; ch0 += 256) {  // (more than 256 channels: in rounds; the order of the final sum stays 0 .. c-1)
    for (int ch = ch0 + wv; ch < c && ch < ch0 + 256; ch += 16) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      for (int nn = 0; nn < n; ++nn) {
        const float* p = partial + ((size_t)nn * c + ch) * nblocks;
        int b = lane;
        for (; b + 192 < nblocks; b += 256) {
          a0 += (double)p[b];
          a1 += (double)p[b + 64];
          a2 += (double)p[b + 128];
          a3 += (double)p[b + 192];
        }
        for (; b < nblocks; b += 64) a0 += (double)p[b];
      }
      const double a = wave_sum((a0 + a1) + (a2 + a3));
      if (lane == 0) chsum[ch - ch0] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float total = ch0 ? *loss : 0.f;
      for (int ch = ch0; ch < c && ch < ch0 + 256; ++ch) total += (cweight ? cweight[ch] : 1.f) * (float)(chsum[ch - ch0] / count);
      *loss = total;
    }
    __syncthreads();
  }
}
template <typename TT>
__global__ __launch_bounds__(256) void hm_bwd_kernel(const float* __restrict__ out, const TT* __restrict__ tgt, int64_t tgt_sn,
                                                     const float* __restrict__ cweight, const float* __restrict__ dloss,
                                                     float* __restrict__ dout, int c, size_t spatial, int64_t sn,
                                                     int64_t sc, int kind, float inv_count) {
  const int ch = blockIdx.y, n = blockIdx.z;
  const float scale = *dloss * (cweight ? cweight[ch] : 1.f) * inv_count;
  const size_t v0 = (size_t)blockIdx.x * LOSS_BLOCK_VOX;
  for (int it = 0; it < 8; ++it) {
    const size_t v = v0 + (size_t)it * 256 + threadIdx.x;
    if (v < spatial) {
      const size_t o = (size_t)n * tgt_sn + (size_t)ch * spatial + v, os = (size_t)n * sn + (size_t)ch * sc + v;
      const float d = out[os] - ld(tgt, o);
      dout[os] = kind == MEDNET_REG_L2 ? 2.f * d * scale : (d > 0.f ? scale : (d < 0.f ? -scale : 0.f));
    }
  }
}

// (for head_mfma.hip's fused landmark head, whose partial sums have hm_fwd_kernel's / dice_fwd_kernel's layouts)
int launch_hm_finalize(const float* partial, const float* cweight, float* loss, int n, int c, int nblocks, size_t spatial,
                       hipStream_t s) {
  hipLaunchKernelGGL(hm_finalize_kernel, dim3(1), dim3(1024), 0, s, partial, cweight, loss, n, c, nblocks, (double)n * (double)spatial);
  return check_launch("heatmap_loss_finalize");
}
int launch_dice_finalize(const float* partial, const float* weight, float* loss, float* saved, int c, int nblocks, float eps,
                         hipStream_t s) {
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, s, partial, weight, loss, saved, (float*)nullptr, c, nblocks, eps);
  return check_launch("dice_finalize");
}

}  // namespace mednet

using namespace mednet;

extern "C" size_t mednet_loss_ws_bytes(int n, int c, size_t spatial) {
  return ((size_t)n * loss_blocks(spatial) * c * 2 + 64) * sizeof(float);
}

#define LOSS_DISPATCH_C(c, CALL)       \
  do {                                 \
    if ((c) <= 2) { CALL(2); }         \
    else if ((c) <= 4) { CALL(4); }    \
    else if ((c) <= 8) { CALL(8); }    \
    else if ((c) <= 16) { CALL(16); }  \
    else { CALL(32); }                 \
  } while (0)

// labels: MEDNET_I64 (N x spatial, the reference's `.long()`) or MEDNET_U8, element stride label_stride_n between samples (the last
// channel of a uint8 label volume is consumed where it lies: no cast kernel, landmarks.py:70 / segmentation.py:60)
extern "C" int mednet_dice_fwd_lt(const float* logits, const void* labels, int label_dtype, int64_t label_stride_n, const float* weight,
                                  float* loss, float* saved, float* dice_out, int n, int c, size_t spatial, int64_t stride_n,
                                  int64_t stride_c, float eps, int sigmoid, int ignore_index, void* ws, size_t ws_bytes,
                                  mednet_stream stream) {
  MEDNET_REQUIRE(c >= 1 && c <= 32, MEDNET_E_UNSUPPORTED, "dice_fwd: C=%d (supported: 1..32)", c);
  MEDNET_REQUIRE(n > 0 && spatial > 0, MEDNET_E_SHAPE, "dice_fwd: empty input");
  MEDNET_REQUIRE(label_dtype == MEDNET_I64 || label_dtype == MEDNET_U8, MEDNET_E_DTYPE, "dice_fwd: labels are int64 or uint8 (got %d)", label_dtype);
  MEDNET_REQUIRE(ws_bytes >= mednet_loss_ws_bytes(n, c, spatial), MEDNET_E_WORKSPACE, "dice_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = loss_blocks(spatial);
  float* partial = (float*)ws;
#define CALL(M)                                                                                                                        \
  do {                                                                                                                                 \
    if (label_dtype == MEDNET_U8)                                                                                                      \
      hipLaunchKernelGGL((dice_fwd_kernel<M, uint8_t>), dim3(nb, n), dim3(256), 0, s, logits, (const uint8_t*)labels, label_stride_n, \
                         partial, c, spatial, stride_n, stride_c, sigmoid, ignore_index);                                             \
    else                                                                                                                               \
      hipLaunchKernelGGL((dice_fwd_kernel<M, int64_t>), dim3(nb, n), dim3(256), 0, s, logits, (const int64_t*)labels, label_stride_n, \
                         partial, c, spatial, stride_n, stride_c, sigmoid, ignore_index);                                             \
  } while (0)
  LOSS_DISPATCH_C(c, CALL);
#undef CALL
  int rc = check_launch("dice_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, s, partial, weight, loss, saved, dice_out, c, (int)(nb * n), eps);
  return check_launch("dice_finalize");
}
extern "C" int mednet_dice_fwd(const float* logits, const int64_t* labels, const float* weight, float* loss,
                               float* saved, float* dice_out, int n, int c, size_t spatial, int64_t stride_n,
                               int64_t stride_c, float eps, int sigmoid, int ignore_index, void* ws, size_t ws_bytes,
                               mednet_stream stream) {
  return mednet_dice_fwd_lt(logits, labels, MEDNET_I64, (int64_t)spatial, weight, loss, saved, dice_out, n, c, spatial, stride_n, stride_c,
                            eps, sigmoid, ignore_index, ws, ws_bytes, stream);
}

extern "C" int mednet_dice_bwd_lt(const float* logits, const void* labels, int label_dtype, int64_t label_stride_n, const float* weight,
                                  const float* saved, const float* dloss, float* dlogits, int n, int c, size_t spatial,
                                  int64_t stride_n, int64_t stride_c, float eps, int sigmoid, int ignore_index, mednet_stream stream) {
  MEDNET_REQUIRE(c >= 1 && c <= 32, MEDNET_E_UNSUPPORTED, "dice_bwd: C=%d (supported: 1..32)", c);
  MEDNET_REQUIRE(label_dtype == MEDNET_I64 || label_dtype == MEDNET_U8, MEDNET_E_DTYPE, "dice_bwd: labels are int64 or uint8 (got %d)", label_dtype);
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = loss_blocks(spatial);
#define CALL(M)                                                                                                                        \
  do {                                                                                                                                 \
    if (label_dtype == MEDNET_U8)                                                                                                      \
      hipLaunchKernelGGL((dice_bwd_kernel<M, uint8_t>), dim3(nb, n), dim3(256), 0, s, logits, (const uint8_t*)labels, label_stride_n, \
                         weight, saved, dloss, dlogits, c, spatial, stride_n, stride_c, eps, sigmoid, ignore_index);                  \
    else                                                                                                                               \
      hipLaunchKernelGGL((dice_bwd_kernel<M, int64_t>), dim3(nb, n), dim3(256), 0, s, logits, (const int64_t*)labels, label_stride_n, \
                         weight, saved, dloss, dlogits, c, spatial, stride_n, stride_c, eps, sigmoid, ignore_index);                  \
  } while (0)
  LOSS_DISPATCH_C(c, CALL);
#undef CALL
  return check_launch("dice_bwd");
}
extern "C" int mednet_dice_bwd(const float* logits, const int64_t* labels, const float* weight, const float* saved,
                               const float* dloss, float* dlogits, int n, int c, size_t spatial, int64_t stride_n,
                               int64_t stride_c, float eps, int sigmoid, int ignore_index, mednet_stream stream) {
  return mednet_dice_bwd_lt(logits, labels, MEDNET_I64, (int64_t)spatial, weight, saved, dloss, dlogits, n, c, spatial, stride_n, stride_c,
                            eps, sigmoid, ignore_index, stream);
}

extern "C" int mednet_ce_fwd(const float* logits, const int64_t* labels, const float* weight, float* loss, float* saved,
                             int n, int c, size_t spatial, int64_t stride_n, int64_t stride_c, int ignore_index,
                             void* ws, size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(c >= 1 && c <= 32, MEDNET_E_UNSUPPORTED, "ce_fwd: C=%d (supported: 1..32)", c);
  MEDNET_REQUIRE(ws_bytes >= mednet_loss_ws_bytes(n, c, spatial), MEDNET_E_WORKSPACE, "ce_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = loss_blocks(spatial);
  float* partial = (float*)ws;
#define CALL(M) hipLaunchKernelGGL(ce_fwd_kernel<M>, dim3(nb, n), dim3(256), 0, s, logits, labels, weight, partial, c, spatial, stride_n, stride_c, ignore_index)
  LOSS_DISPATCH_C(c, CALL);
#undef CALL
  int rc = check_launch("ce_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, partial, loss, saved, (int)(nb * n));
  return check_launch("ce_finalize");
}

extern "C" int mednet_ce_bwd(const float* logits, const int64_t* labels, const float* weight, const float* saved,
                             const float* dloss, float* dlogits, int n, int c, size_t spatial, int64_t stride_n,
                             int64_t stride_c, int ignore_index, mednet_stream stream) {
  MEDNET_REQUIRE(c >= 1 && c <= 32, MEDNET_E_UNSUPPORTED, "ce_bwd: C=%d (supported: 1..32)", c);
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = loss_blocks(spatial);
#define CALL(M) hipLaunchKernelGGL(ce_bwd_kernel<M>, dim3(nb, n), dim3(256), 0, s, logits, labels, weight, saved, dloss, dlogits, c, spatial, stride_n, stride_c, ignore_index)
  LOSS_DISPATCH_C(c, CALL);
#undef CALL
  return check_launch("ce_bwd");
}

// target_stride_n: element stride between the samples of `target` (channel ch of sample n at + n * target_stride_n + ch * spatial):
// the heat-map channels of a label volume are consumed where they lie, no .contiguous() copy (landmarks.py:68)
extern "C" int mednet_heatmap_loss_fwd_strided(const float* out, const void* target, int64_t target_stride_n, const float* cweight,
                                               float* loss, int n, int c, size_t spatial, int64_t stride_n, int64_t stride_c, int kind,
                                               int tgt_u8, void* ws, size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(n > 0 && c > 0 && spatial > 0 && c <= 65535 && n <= 65535, MEDNET_E_SHAPE, "heatmap_loss_fwd: bad shape");
  MEDNET_REQUIRE(ws_bytes >= mednet_loss_ws_bytes(n, c, spatial), MEDNET_E_WORKSPACE, "heatmap_loss_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = loss_blocks(spatial);
  float* partial = (float*)ws;
  if (tgt_u8) hipLaunchKernelGGL(hm_fwd_kernel<uint8_t>, dim3(nb, c, n), dim3(256), 0, s, out, (const uint8_t*)target, target_stride_n, partial, c, spatial, stride_n, stride_c, kind);
  else hipLaunchKernelGGL(hm_fwd_kernel<float>, dim3(nb, c, n), dim3(256), 0, s, out, (const float*)target, target_stride_n, partial, c, spatial, stride_n, stride_c, kind);
  int rc = check_launch("heatmap_loss_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(hm_finalize_kernel, dim3(1), dim3(1024), 0, s, partial, cweight, loss, n, c, (int)nb, (double)n * (double)spatial);
  return check_launch("heatmap_loss_finalize");
}
extern "C" int mednet_heatmap_loss_fwd(const float* out, const void* target, const float* cweight, float* loss, int n,
                                       int c, size_t spatial, int64_t stride_n, int64_t stride_c, int kind, int tgt_u8,
                                       void* ws, size_t ws_bytes, mednet_stream stream) {
  return mednet_heatmap_loss_fwd_strided(out, target, (int64_t)c * (int64_t)spatial, cweight, loss, n, c, spatial, stride_n, stride_c, kind,
                                         tgt_u8, ws, ws_bytes, stream);
}

extern "C" int mednet_heatmap_loss_bwd_strided(const float* out, const void* target, int64_t target_stride_n, const float* cweight,
                                               const float* dloss, float* dout, int n, int c, size_t spatial, int64_t stride_n,
                                               int64_t stride_c, int kind, int tgt_u8, mednet_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = loss_blocks(spatial);
  const float inv = (float)(1.0 / ((double)n * (double)spatial));
  if (tgt_u8) hipLaunchKernelGGL(hm_bwd_kernel<uint8_t>, dim3(nb, c, n), dim3(256), 0, s, out, (const uint8_t*)target, target_stride_n, cweight, dloss, dout, c, spatial, stride_n, stride_c, kind, inv);
  else hipLaunchKernelGGL(hm_bwd_kernel<float>, dim3(nb, c, n), dim3(256), 0, s, out, (const float*)target, target_stride_n, cweight, dloss, dout, c, spatial, stride_n, stride_c, kind, inv);
  return check_launch("heatmap_loss_bwd");
}
extern "C" int mednet_heatmap_loss_bwd(const float* out, const void* target, const float* cweight, const float* dloss,
                                       float* dout, int n, int c, size_t spatial, int64_t stride_n, int64_t stride_c,
                                       int kind, int tgt_u8, mednet_stream stream) {
  return mednet_heatmap_loss_bwd_strided(out, target, (int64_t)c * (int64_t)spatial, cweight, dloss, dout, n, c, spatial, stride_n, stride_c,
                                         kind, tgt_u8, stream);
}
