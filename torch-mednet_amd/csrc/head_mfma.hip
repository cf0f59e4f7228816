// The landmark head on the matrix cores (16-bit storage): the 1x1x1 `final_conv` (model.py:207) of LandmarkNet -- 32 features ->
// nh heat maps + ncls classes (landmarks.py:71-75: 16 + 2) -- fused with BOTH of its losses, the per-channel weighted heat-map
// regression (landmarks.py:125-134) and the Dice loss of the class channels (loss.py:114-130).
//
// Unfused, the 18 planar fp32 logit planes (604 MB at config 4) are written by the head, read by the two loss forwards, read
// again by the two loss backwards, which write their gradient (604 MB), which the head's data gradient, weight gradient and
// bias sums each read again: 2.4 ms of a 21.1 ms step, and with 18 x 32 multiply-adds per voxel and direction the VALU forms
// of the head are compute-bound (head_dgrad_gn_kernel 653 us, wgrad_1x1_kernel 914 us for 1.1 - 1.7 GB each).  Here
//   forward   head_lm_kernel<false>: logits^T = W z^T on v_mfma_f32_32x32x16 (W split hi + lo: fp32-level products), loss terms taken
//             from the accumulator registers; nothing but per-workgroup partial sums is written (the logits only on request).
//   backward  head_lm_kernel<true>: the same MFMAs rebuild the logits bit for bit, the logit gradient is formed in registers
//             (closed forms of loss.hip's hm_bwd_kernel / dice_bwd_kernel) and feeds, split hi + lo,
//               dz^T = W^T dl^T      (stored; the first pass of the producing block's GroupNorm-3 backward is taken from the stored
//                                     rows as head_dgrad_gn_kernel does),
//               dW  += dl^T z        (both operands through the transposing LDS read, contraction over 32 voxels per step),
//               db  += dl            (lane sums).
//             It reads z, the GroupNorm input, the uint8 targets and labels once and writes dz once: 1.75 GB at config 4.
// A wave owns runs of 128 consecutive voxels; lane (g, h) = (lane % 32, lane / 32) holds voxels 4g .. 4g+3 of the run (sub-tile
// j = voxels {4g + j}) and, of each voxel row, the two 16-byte pieces at bytes 16h and 32 + 16h -- channels 8h .. 8h+7 and
// 16+8h .. 16+8h+7 -- for z, the GroupNorm input and dz alike: the row order of the weight operands is permuted (free: they
// are built once per wave from the fp32 weights) so that every MFMA result lands in the lane that owns those bytes.
#include "conv.h"
#include "elt16.inc"

namespace mednet {

constexpr int HLM_RUN = 128;       // voxels per wave and trip
constexpr int HLM_MAXH = 16;       // heat maps (one MFMA k-block)
constexpr int HLM_MAXC = 4;        // classes (second k-block, lanes 0..31)
constexpr int HLM_WIDTH = 32 * 33; // per-workgroup partial of (dW [32 k'][32 ci], db [32 k']); k' = heat map c, or 16 + class
constexpr int HLM_SCR = 33;        // floats per thread of the end-of-kernel LDS scratch

struct HlmArgs {
  const elt* z;          // [n][spatial][32]: head input (the last block's output)
  const float* W;        // [nh + ncls][32] fp32
  const float* bias;     // [nh + ncls], nullable
  const uint8_t* tgt;    // heat-map targets, planar: sample stride tgt_sn, plane stride spatial
  const uint8_t* lab;    // class labels: sample stride lab_sn
  int64_t tgt_sn, lab_sn;
  size_t spatial;        // % 4 == 0
  int nh, ncls;
  int runs, chunk_runs, chunks;  // runs per sample, runs per workgroup, workgroups per sample
  int kind, sigmoid, ignore;
  // forward
  float* hm_partial;     // [(n * nh + c)][chunks]           (hm_finalize_kernel's layout)
  float* dice_partial;   // [n][chunk][ncls][2]              (dice_finalize_kernel's layout)
  float* logits;         // nullable: [n][nh + ncls][spatial] fp32
  // backward
  const float* saved;    // [ncls][2] = {I, D} of the Dice forward
  const float* cls_weight;  // [ncls] nullable
  const float* reg_weight;  // [nh] nullable
  const float* dcls;     // upstream gradient of the class loss (scalar)
  const float* dreg;     // ... of the regression loss
  float eps, inv_count;
  elt* dz;               // [n][spatial][32]
  const elt* gn_y;       // nullable: input of the GroupNorm whose activated (+ residual) output z is
  int gn_act;
  float* gn_partial;     // [n][chunks][32][2] = {sum du, sum du * gn_y}, du = dz * act'(z)
  float* wpart;          // [n * chunks][HLM_WIDTH]
};

template <bool BWD>
__global__ __launch_bounds__(256, 2) void head_lm_kernel(HlmArgs a) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  __shared__ __attribute__((aligned(16))) char smem[256 * HLM_SCR * 4];
  __shared__ float cst[2][32];
  __shared__ u32x4 wops[8][64];  // the 8 weight operands (below), one 16-byte fragment per lane: the same in every wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane & 31, h = lane >> 5;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int m = a.nh + a.ncls;

  // ---- weight operands, built once: hi + lo parts of the fp32 weights ---------------------------------------------
  // logits^T [k' row i][voxel]: A = W rows in the order that gives lane (g, h) its heat maps 8h .. 8h+7 in acc[0..7] and (h = 0)
  // the classes in acc[8..11]: row i = 8q + 4hh + t -> q 0/1: heat map 8hh + 4q + t; q 2, hh 0: class t
  const int ri = lane & 31, rq = ri >> 3, rhh = (ri & 7) >> 2, rt = ri & 3;
  const int lrow = rq < 2 ? (8 * rhh + 4 * rq + rt < a.nh ? 8 * rhh + 4 * rq + rt : -1)
                          : ((rq == 2 && rhh == 0 && rt < a.ncls) ? a.nh + rt : -1);
  if (wv == 0) {
    eltx8 wl_hi[2], wl_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float w = lrow >= 0 ? a.W[lrow * 32 + 16 * s + 8 * h + e] : 0.f;
        wl_hi[s][e] = (elt)w;
        wl_lo[s][e] = (elt)(w - (float)wl_hi[s][e]);
      }
      wops[s][lane] = __builtin_bit_cast(u32x4, wl_hi[s]);
      wops[2 + s][lane] = __builtin_bit_cast(u32x4, wl_lo[s]);
    }
  }
  // dz^T [channel row i][voxel]: row i = 8q + 4hh + t stands for channel (q / 2) * 16 + 8hh + (q % 2) * 4 + t, so that lane
  // (g, h) receives channels 8h .. 8h+7 in acc[0..7] and 16+8h .. in acc[8..15]; k-slots 8h + e: block 0 = heat map 8h + e,
  // block 1 = class e (lanes of h = 0)
  if (BWD && wv == 1) {
    eltx8 wd_hi[2], wd_lo[2];
    const int ch = (rq >> 1) * 16 + 8 * rhh + (rq & 1) * 4 + rt;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float w0 = 8 * h + e < a.nh ? a.W[(8 * h + e) * 32 + ch] : 0.f;
      const float w1 = (h == 0 && e < a.ncls) ? a.W[(a.nh + e) * 32 + ch] : 0.f;
      wd_hi[0][e] = (elt)w0;
      wd_lo[0][e] = (elt)(w0 - (float)wd_hi[0][e]);
      wd_hi[1][e] = (elt)w1;
      wd_lo[1][e] = (elt)(w1 - (float)wd_hi[1][e]);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      wops[4 + b][lane] = __builtin_bit_cast(u32x4, wd_hi[b]);
      wops[6 + b][lane] = __builtin_bit_cast(u32x4, wd_lo[b]);
    }
  }
  // ---- constants of a lane half, in LDS (they would hold 36 registers for the whole kernel): per h the biases of its 8 heat maps
  // [0..7] and of the classes [8..11], the heat maps' gradient scales [12..19] and the Dice gradient terms gI [20..23], gD [24..27]
  if (wv == 2) {
    const int hh = lane >> 5, i = lane & 31;
    float v = 0.f;
    if (i < 8) v = (a.bias && 8 * hh + i < a.nh) ? a.bias[8 * hh + i] : 0.f;
    else if (i < 12) v = (a.bias && i - 8 < a.ncls) ? a.bias[a.nh + i - 8] : 0.f;
    else if (BWD && i < 20) {
      const int c = 8 * hh + i - 12;
      v = c < a.nh ? *a.dreg * (a.reg_weight ? a.reg_weight[c] : 1.f) * a.inv_count : 0.f;
    } else if (BWD && i < 28) {  // loss.hip dice_bwd_kernel
      const int k = (i - 20) & 3;
      if (k < a.ncls) {
        const float gc = *a.dcls, w = a.cls_weight ? a.cls_weight[k] : 1.f;
        const float I = a.saved[2 * k], D = a.saved[2 * k + 1];
        const float Dc = fmaxf(D, a.eps);
        v = i < 24 ? -2.f * w / ((float)a.ncls * Dc) * gc + ((I != I || D != D) ? __builtin_nanf("") : 0.f)
                   : (D >= a.eps ? 2.f * w * I / ((float)a.ncls * Dc * Dc) : 0.f) * gc;
      }
    }
    cst[hh][i] = v;
  }
  __syncthreads();
  // 0,1: logits hi; 2,3: lo; 4,5: dz hi; 6,7: lo.  `wbase` is made opaque once per sub-tile: left alone, the compiler hoists the
  // eight loop-invariant fragments back into 32 registers for the whole kernel
  const u32x4* wbase = &wops[0][lane];
  const float* cbase = &cst[h][0];
  auto wop = [&](int i) { return __builtin_bit_cast(eltx8, wbase[i * 64]); };
  // ---- accumulators ---------------------------------------------------------------------------------------------------
  float hm_acc[8], dI[HLM_MAXC], dD[HLM_MAXC];  // forward
  float ss[16], sq[16], dbh[8], dbc[HLM_MAXC];  // backward
  f32x16 accw;
  bool bad = false;
#pragma unroll
  for (int e = 0; e < 8; ++e) hm_acc[e] = dbh[e] = 0.f;
#pragma unroll
  for (int k = 0; k < HLM_MAXC; ++k) dI[k] = dD[k] = dbc[k] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) ss[i] = sq[i] = accw[i] = 0.f;

  // wave-private LDS tiles of the backward: 32 voxel rows x 64 B each of z, dl (hi) and dl (lo)
  char* zt = smem + wv * 6144;
  char* dlh = zt + 2048;
  char* dll = zt + 4096;
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = lane >> 4;
  const int troff = (8 * h + tq) * 64 + (16 * (tg & 1) + 4 * tp) * 2;  // tr_operand: voxel rows 8h + tq (+ 4), 4 columns
  if constexpr (BWD) {  // columns 20 .. 31 of the dl tiles are never written: zero once (rows of dW that nobody reads, but no NaNs)
    const u32x4 zero = {0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(dlh + lane * 32) = zero;
    *reinterpret_cast<u32x4*>(dlh + lane * 32 + 16) = zero;
    *reinterpret_cast<u32x4*>(dll + lane * 32) = zero;
    *reinterpret_cast<u32x4*>(dll + lane * 32 + 16) = zero;
    wave_lds_fence();
  }

  const elt* zs = a.z + (size_t)n * a.spatial * 32;
  const elt* ys = (BWD && a.gn_y) ? a.gn_y + (size_t)n * a.spatial * 32 : nullptr;
  elt* dzs = BWD ? a.dz + (size_t)n * a.spatial * 32 : nullptr;
  const uint8_t* tg8 = a.tgt + (size_t)n * a.tgt_sn;
  const uint8_t* lb8 = a.lab + (size_t)n * a.lab_sn;
  const int run_end = min(a.runs, (chunk + 1) * a.chunk_runs);

  // (Measured, profiles/r05_ab.md: requesting zp[j] of the NEXT run as soon as sub-tile j has used it -- a register-neutral software
  //  pipeline -- made both kernels slower, 0.47 -> 0.52 ms and 0.17 -> 0.22 ms at config 4: the loads stay at the top of a run.)
  for (int run = chunk * a.chunk_runs + wv; run < run_end; run += 4) {
    const size_t vb = (size_t)run * HLM_RUN + 4 * g;  // this lane's first voxel
    const bool live = vb < a.spatial;                 // (spatial % 4 == 0: all four voxels or none)
    // ---- loads: the lane's 4 voxel rows (2 pieces each) of z and of the GroupNorm input, 4 target bytes per heat map, 4 labels
    u32x4 zp[4][2], yp[2][2];  // (yp: sub-tile j in yp[j & 1], sub-tile j + 1 fetched at the top of trip j)
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) zp[j][s] = live ? *reinterpret_cast<const u32x4*>(zs + (vb + j) * 32 + 16 * s + 8 * h) : zero4;
    auto fetch_y = [&](int j) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
        yp[j & 1][s] = (live && ys) ? *reinterpret_cast<const u32x4*>(ys + (vb + j) * 32 + 16 * s + 8 * h) : zero4;
    };
    if constexpr (BWD) fetch_y(0);
    unsigned tb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      tb[e] = (live && 8 * h + e < a.nh) ? *reinterpret_cast<const unsigned*>(tg8 + (size_t)(8 * h + e) * a.spatial + vb) : 0u;
    const unsigned lb = (live && h == 0) ? *reinterpret_cast<const unsigned*>(lb8 + vb) : 0u;

#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("" : "+v"(wbase), "+v"(cbase));
      if constexpr (BWD) {
        if (j < 3) fetch_y(j + 1);
      }
      // ---- logits of the 32 voxels {4g + j}: 4 MFMAs (2 k-steps x hi / lo)
      f32x16 lg;
#pragma unroll
      for (int i = 0; i < 16; ++i) lg[i] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const eltx8 zb = __builtin_bit_cast(eltx8, zp[j][s]);
        lg = MEDNET_MFMA_32x32x16(wop(s), zb, lg, 0, 0, 0);
        lg = MEDNET_MFMA_32x32x16(wop(2 + s), zb, lg, 0, 0, 0);
      }
      float lh[8], lc[HLM_MAXC], p[HLM_MAXC];
#pragma unroll
      for (int e = 0; e < 8; ++e) lh[e] = lg[e] + cbase[e];
#pragma unroll
      for (int k = 0; k < HLM_MAXC; ++k) lc[k] = lg[8 + k] + cbase[8 + k];
      if (!BWD && a.logits && live) {
        float* lo = a.logits + (size_t)n * m * a.spatial + vb + j;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (8 * h + e < a.nh) lo[(size_t)(8 * h + e) * a.spatial] = lh[e];
        if (h == 0) {
#pragma unroll
          for (int k = 0; k < HLM_MAXC; ++k)
            if (k < a.ncls) lo[(size_t)(a.nh + k) * a.spatial] = lc[k];
        }
      }
      const bool cls_lane = live && h == 0;
      const int yl = (int)((lb >> (8 * j)) & 0xFFu);
      {
        // softmax / sigmoid of the class logits (loss.hip probs_of on registers)
        if (a.sigmoid) {
#pragma unroll
          for (int k = 0; k < HLM_MAXC; ++k) p[k] = k < a.ncls ? 1.f / (1.f + expf(-lc[k])) : 0.f;
        } else {
          float mx = -INFINITY;
#pragma unroll
          for (int k = 0; k < HLM_MAXC; ++k)
            if (k < a.ncls) mx = fmaxf(mx, lc[k]);
          float den = 0.f;
#pragma unroll
          for (int k = 0; k < HLM_MAXC; ++k) {
            p[k] = k < a.ncls ? expf(lc[k] - mx) : 0.f;
            den += p[k];
          }
          const float inv = 1.f / den;
#pragma unroll
          for (int k = 0; k < HLM_MAXC; ++k) p[k] *= inv;
        }
      }
      if constexpr (!BWD) {
        // ---- loss terms: hm_fwd_kernel / dice_fwd_kernel on registers
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = lh[e] - (float)((tb[e] >> (8 * j)) & 0xFFu);
          const float t = a.kind == MEDNET_REG_L2 ? d * d : fabsf(d);
          hm_acc[e] += (live && 8 * h + e < a.nh) ? t : 0.f;
        }
        if (cls_lane) {
          bad |= (unsigned)yl >= (unsigned)a.ncls;
#pragma unroll
          for (int k = 0; k < HLM_MAXC; ++k)
            if (k < a.ncls) {
              const float t = (k == yl) ? 1.f : 0.f;
              const float mk = (a.ignore != MEDNET_NO_IGNORE && t == (float)a.ignore) ? 0.f : 1.f;
              dI[k] = fmaf(p[k] * mk, t * mk, dI[k]);
              dD[k] += (p[k] + t) * mk;
            }
        }
      } else {
        // ---- logit gradient in registers: hm_bwd_kernel / dice_bwd_kernel
        float dh[8], dc[HLM_MAXC];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = lh[e] - (float)((tb[e] >> (8 * j)) & 0xFFu);
          const float sc = live ? cbase[12 + e] : 0.f;  // (0 for heat maps >= nh)
          dh[e] = a.kind == MEDNET_REG_L2 ? 2.f * d * sc : (d > 0.f ? sc : (d < 0.f ? -sc : 0.f));
          dbh[e] += dh[e];
        }
        {
          float gg[HLM_MAXC], dot = 0.f;
#pragma unroll
          for (int k = 0; k < HLM_MAXC; ++k) {
            const float t = (k == yl) ? 1.f : 0.f;
            const float mk = (a.ignore != MEDNET_NO_IGNORE && t == (float)a.ignore) ? 0.f : 1.f;
            gg[k] = k < a.ncls ? mk * (cbase[20 + k] * t * mk + cbase[24 + k]) : 0.f;
            dot = fmaf(p[k], gg[k], dot);
          }
#pragma unroll
          for (int k = 0; k < HLM_MAXC; ++k) {
            const float v = a.sigmoid ? gg[k] * p[k] * (1.f - p[k]) : p[k] * (gg[k] - dot);
            dc[k] = (cls_lane && k < a.ncls) ? v : 0.f;
            dbc[k] += dc[k];
          }
        }
        eltx8 dh_hi, dh_lo, dc_hi, dc_lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dh_hi[e] = (elt)dh[e];
          dh_lo[e] = (elt)(dh[e] - (float)dh_hi[e]);
          const float c = e < HLM_MAXC ? dc[e & (HLM_MAXC - 1)] : 0.f;
          dc_hi[e] = (elt)c;
          dc_lo[e] = (elt)(c - (float)dc_hi[e]);
        }
        // ---- dz^T = W^T dl^T: 6 MFMAs (two k-blocks x {hi hi, lo hi, hi lo})
        f32x16 dzv;
#pragma unroll
        for (int i = 0; i < 16; ++i) dzv[i] = 0.f;
        dzv = MEDNET_MFMA_32x32x16(wop(4), dh_hi, dzv, 0, 0, 0);
        dzv = MEDNET_MFMA_32x32x16(wop(6), dh_hi, dzv, 0, 0, 0);
        dzv = MEDNET_MFMA_32x32x16(wop(4), dh_lo, dzv, 0, 0, 0);
        dzv = MEDNET_MFMA_32x32x16(wop(5), dc_hi, dzv, 0, 0, 0);
        dzv = MEDNET_MFMA_32x32x16(wop(7), dc_hi, dzv, 0, 0, 0);
        dzv = MEDNET_MFMA_32x32x16(wop(5), dc_lo, dzv, 0, 0, 0);
        eltx8 o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o0[e] = (elt)dzv[e];
          o1[e] = (elt)dzv[8 + e];
        }
        if (live) {
          *reinterpret_cast<u32x4*>(dzs + (vb + j) * 32 + 8 * h) = __builtin_bit_cast(u32x4, o0);
          *reinterpret_cast<u32x4*>(dzs + (vb + j) * 32 + 16 + 8 * h) = __builtin_bit_cast(u32x4, o1);
        }
        // ---- first pass of the GroupNorm backward in front, from the STORED rows (head_dgrad_gn_kernel): du = dz * act'(z)
        if (ys) {  // (workgroup-uniform)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const eltx8 zv = __builtin_bit_cast(eltx8, zp[j][s]), yv = __builtin_bit_cast(eltx8, yp[j & 1][s]);
            const eltx8 ov = s ? o1 : o0;
            float du[8], zz[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              du[e] = (float)ov[e];
              zz[e] = (float)zv[e];
            }
            act_grad_n<8>(du, zz, a.gn_act);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              ss[8 * s + e] += du[e];
              sq[8 * s + e] = fmaf(du[e], (float)yv[e], sq[8 * s + e]);
            }
          }
        }
        // ---- dW += dl^T z over these 32 voxels: tiles [voxel row g][64 B] in this wave's LDS, read back transposed
        wave_lds_fence();  // (the reads of the previous sub-tile are above these writes)
        *reinterpret_cast<u32x4*>(zt + g * 64 + 16 * h) = zp[j][0];
        *reinterpret_cast<u32x4*>(zt + g * 64 + 32 + 16 * h) = zp[j][1];
        *reinterpret_cast<u32x4*>(dlh + g * 64 + 16 * h) = __builtin_bit_cast(u32x4, dh_hi);
        *reinterpret_cast<u32x4*>(dll + g * 64 + 16 * h) = __builtin_bit_cast(u32x4, dh_lo);
        if (h == 0) {
          typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
          const u32x4 ch4 = __builtin_bit_cast(u32x4, dc_hi), cl4 = __builtin_bit_cast(u32x4, dc_lo);
          const u32x2 ch2 = {ch4[0], ch4[1]}, cl2 = {cl4[0], cl4[1]};
          *reinterpret_cast<u32x2*>(dlh + g * 64 + 32) = ch2;
          *reinterpret_cast<u32x2*>(dll + g * 64 + 32) = cl2;
        }
        wave_lds_fence();  // the rows below were written by OTHER lanes of this wave
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const eltx8 fz = tr_operand(zt + ks * 1024 + troff, 256);
          const eltx8 fh = tr_operand(dlh + ks * 1024 + troff, 256);
          const eltx8 fl = tr_operand(dll + ks * 1024 + troff, 256);
          accw = MEDNET_MFMA_32x32x16(fh, fz, accw, 0, 0, 0);
          accw = MEDNET_MFMA_32x32x16(fl, fz, accw, 0, 0, 0);
        }
      }
    }
  }

  // ---- end of kernel: lane and wave sums through LDS in a fixed order -----------------------------------------------
  float* scr = reinterpret_cast<float*>(smem);
  float* mine = scr + tid * HLM_SCR;
  __syncthreads();
  if constexpr (!BWD) {
    if (bad) dI[0] = dD[0] = __builtin_nanf("");
#pragma unroll
    for (int e = 0; e < 8; ++e) mine[e] = hm_acc[e];
#pragma unroll
    for (int k = 0; k < HLM_MAXC; ++k) {
      mine[8 + k] = dI[k];
      mine[12 + k] = dD[k];
    }
    __syncthreads();
    if (tid < 16 + 2 * HLM_MAXC) {
      // heat map c = 8hh + e: lanes of half hh, value e; class sums: lanes of half 0
      const int hh = tid < 16 ? tid >> 3 : 0, idx = tid < 16 ? tid & 7 : tid - 8;
      float s = 0.f;
      for (int w = 0; w < 4; ++w)
        for (int gg = 0; gg < 32; ++gg) s += scr[(w * 64 + hh * 32 + gg) * HLM_SCR + idx];
      if (tid < 16) {
        if (tid < a.nh) a.hm_partial[((size_t)n * a.nh + tid) * a.chunks + chunk] = s;
      } else {
        const int k = (tid - 16) & (HLM_MAXC - 1), which = (tid - 16) / HLM_MAXC;
        if (k < a.ncls) a.dice_partial[(((size_t)n * a.chunks + chunk) * a.ncls + k) * 2 + which] = s;
      }
    }
  } else {
    // (a) GroupNorm sums: 16 channels x {ss, sq} per lane
    if (a.gn_partial) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        mine[i] = ss[i];
        mine[16 + i] = sq[i];
      }
      __syncthreads();
      if (tid < 64) {
        const int which = tid & 1, idx = (tid >> 1) & 15, hh = tid >> 5;
        float s = 0.f;
        for (int w = 0; w < 4; ++w)
          for (int gg = 0; gg < 32; ++gg) s += scr[(w * 64 + hh * 32 + gg) * HLM_SCR + which * 16 + idx];
        const int ch = (idx >> 3) * 16 + 8 * hh + (idx & 7);
        a.gn_partial[(((size_t)n * a.chunks + chunk) * 32 + ch) * 2 + which] = s;
      }
      __syncthreads();
    }
    float* wp = a.wpart + ((size_t)n * a.chunks + chunk) * HLM_WIDTH;
    // (b) dW: 16 accumulator values per lane, summed over the 4 waves
#pragma unroll
    for (int i = 0; i < 16; ++i) mine[i] = accw[i];
    // (c) db behind them
#pragma unroll
    for (int e = 0; e < 8; ++e) mine[16 + e] = dbh[e];
#pragma unroll
    for (int k = 0; k < HLM_MAXC; ++k) mine[24 + k] = dbc[k];
    __syncthreads();
    for (int o = tid; o < 1024; o += 256) {
      const int L = o & 63, i = o >> 6;
      const float s = (scr[(0 * 64 + L) * HLM_SCR + i] + scr[(1 * 64 + L) * HLM_SCR + i]) +
                      (scr[(2 * 64 + L) * HLM_SCR + i] + scr[(3 * 64 + L) * HLM_SCR + i]);
      const int kp = 8 * (i >> 2) + 4 * (L >> 5) + (i & 3);  // row k' of dW, column (input channel) L % 32
      wp[kp * 32 + (L & 31)] = s;
    }
    if (tid < 16 + HLM_MAXC) {
      const int hh = tid < 16 ? tid >> 3 : 0, idx = tid < 16 ? 16 + (tid & 7) : 24 + (tid - 16);
      float s = 0.f;
      for (int w = 0; w < 4; ++w)
        for (int gg = 0; gg < 32; ++gg) s += scr[(w * 64 + hh * 32 + gg) * HLM_SCR + idx];
      wp[1024 + tid] = s;  // k' = heat map tid, or 16 + class
    }
  }
}

// dw[co][ci] = sum over workgroups of wpart[.][k'(co)][ci], db[co] likewise; fixed order (deterministic).  A workgroup = 16 outputs x
// 16 row slices (a thread: every 16th row, 8 loads in flight), then the slices in order.
__global__ __launch_bounds__(256) void head_lm_wfinal_kernel(const float* __restrict__ wpart, int rows, int nh, int ncls,
                                                            float* __restrict__ dw, float* __restrict__ db) {
  __shared__ double sh[16][17];
  const int e = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int m = nh + ncls, total = m * 33;
  const int o = blockIdx.x * 16 + e;
  double acc = 0.0;
  if (o < total) {
    const int co = o < m * 32 ? o / 32 : o - m * 32, kp = co < nh ? co : 16 + (co - nh);
    const int src = o < m * 32 ? kp * 32 + o % 32 : 1024 + kp;
    const float* p = wpart + src;
    int r = q;
    for (; r + 7 * 16 < rows; r += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(r + 16 * u) * HLM_WIDTH];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; r < rows; r += 16) acc += (double)p[(size_t)r * HLM_WIDTH];
  }
  sh[q][e] = acc;
  __syncthreads();
  if (q == 0 && o < total) {
    double t = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) t += sh[u][e];
    if (o < m * 32) dw[o] = (float)t;
    else if (db) db[o - m * 32] = (float)t;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
bool head_lm_supported(int cin, int nh, int ncls, int dtype, size_t spatial) {
  return cin == 32 && nh >= 1 && nh <= HLM_MAXH && ncls >= 1 && ncls <= HLM_MAXC && dtype == ELT_DTYPE && spatial % 4 == 0 &&
         tuning_option("head_lm_mfma", 1);
}
static void head_lm_plan(size_t spatial, int& runs, int& chunk_runs, int& chunks) {
  runs = (int)((spatial + HLM_RUN - 1) / HLM_RUN);
  chunk_runs = 64;  // 16 trips per wave
  chunks = (runs + chunk_runs - 1) / chunk_runs;
}
int head_lm_chunks(size_t spatial) {
  int runs, cr, chunks;
  head_lm_plan(spatial, runs, cr, chunks);
  return chunks;
}
size_t head_lm_ws_bytes(int n, size_t spatial, int nh, int ncls) {
  const size_t chunks = head_lm_chunks(spatial);
  const size_t fwd = (size_t)n * chunks * (nh + 2 * ncls);
  const size_t bwd = (size_t)n * chunks * HLM_WIDTH;
  return ((fwd > bwd ? fwd : bwd) + 64) * sizeof(float);
}

int launch_head_lm_fwd(const void* z, const float* W, const float* bias, const void* tgt, int64_t tgt_sn, const void* lab,
                       int64_t lab_sn, float* logits, float* hm_partial, float* dice_partial, int n, size_t spatial, int nh, int ncls,
                       int kind, int sigmoid, int ignore, hipStream_t s) {
  HlmArgs a = {};
  a.z = (const elt*)z; a.W = W; a.bias = bias; a.tgt = (const uint8_t*)tgt; a.lab = (const uint8_t*)lab;
  a.tgt_sn = tgt_sn; a.lab_sn = lab_sn; a.spatial = spatial; a.nh = nh; a.ncls = ncls;
  head_lm_plan(spatial, a.runs, a.chunk_runs, a.chunks);
  a.kind = kind; a.sigmoid = sigmoid; a.ignore = ignore;
  a.hm_partial = hm_partial; a.dice_partial = dice_partial; a.logits = logits;
  MEDNET_REQUIRE(tgt_sn % 4 == 0 && lab_sn % 4 == 0 && ((uintptr_t)tgt & 3) == 0 && ((uintptr_t)lab & 3) == 0, MEDNET_E_SHAPE,
                 "head_lm: targets and labels must be 4-byte aligned per sample");
  hipLaunchKernelGGL(head_lm_kernel<false>, dim3(a.chunks, n), dim3(256), 0, s, a);
  return check_launch("head_lm_fwd");
}

int launch_head_lm_bwd(const void* z, const float* W, const float* bias, const void* tgt, int64_t tgt_sn, const void* lab,
                       int64_t lab_sn, const float* saved, const float* cls_weight, const float* reg_weight, const float* dcls,
                       const float* dreg, float eps, void* dz, const void* gn_y, int gn_act, float* gn_partial, float* dw, float* db,
                       int n, size_t spatial, int nh, int ncls, int kind, int sigmoid, int ignore, void* ws, size_t ws_bytes,
                       hipStream_t s) {
  HlmArgs a = {};
  a.z = (const elt*)z; a.W = W; a.bias = bias; a.tgt = (const uint8_t*)tgt; a.lab = (const uint8_t*)lab;
  a.tgt_sn = tgt_sn; a.lab_sn = lab_sn; a.spatial = spatial; a.nh = nh; a.ncls = ncls;
  head_lm_plan(spatial, a.runs, a.chunk_runs, a.chunks);
  a.kind = kind; a.sigmoid = sigmoid; a.ignore = ignore;
  a.saved = saved; a.cls_weight = cls_weight; a.reg_weight = reg_weight; a.dcls = dcls; a.dreg = dreg;
  a.eps = eps; a.inv_count = (float)(1.0 / ((double)n * (double)spatial));
  a.dz = (elt*)dz; a.gn_y = (const elt*)gn_y; a.gn_act = gn_act; a.gn_partial = gn_partial;
  MEDNET_REQUIRE(tgt_sn % 4 == 0 && lab_sn % 4 == 0 && ((uintptr_t)tgt & 3) == 0 && ((uintptr_t)lab & 3) == 0, MEDNET_E_SHAPE,
                 "head_lm: targets and labels must be 4-byte aligned per sample");
  MEDNET_REQUIRE((gn_y == nullptr) == (gn_partial == nullptr), MEDNET_E_SHAPE, "head_lm_bwd: gn_y and gn_partial go together");
  MEDNET_REQUIRE(ws_bytes >= head_lm_ws_bytes(n, spatial, nh, ncls), MEDNET_E_WORKSPACE, "head_lm_bwd: workspace too small");
  a.wpart = (float*)ws;
  hipLaunchKernelGGL(head_lm_kernel<true>, dim3(a.chunks, n), dim3(256), 0, s, a);
  int rc = check_launch("head_lm_bwd");
  if (rc) return rc;
  const int total = (nh + ncls) * 33;
  hipLaunchKernelGGL(head_lm_wfinal_kernel, dim3((total + 15) / 16), dim3(256), 0, s, a.wpart, n * a.chunks, nh, ncls, dw, db);
  return check_launch("head_lm_wfinal");
}

}  // namespace mednet
