// Split-bf16 matrix-core kernels for the 3x3x3 convolution family in the fp32 STORAGE mode (config.precision "fp32"): the
// mode that meets the reference within 1e-3, at bf16 matrix-core speed instead of the fp32 MFMA's (1/16 of it).
//
// Every fp32 operand is split where it enters LDS:   v = hi + lo,  hi = bf16(v),  lo = bf16(v - hi)   (16 significant bits)
// and a product is contracted as three bf16 MFMAs with fp32 accumulation,
//        x * w  ~  x_hi * w_hi  +  x_hi * w_lo  +  x_lo * w_hi                 (the dropped x_lo * w_lo term is 2^-16 relative)
// so the result differs from the exact fp32 fmaf chain of conv_f32_mfma.hip by ~2^-16 per product -- 256x below bf16 storage's
// error, an order of magnitude inside the 1e-3 budget after 21 layers and their backward (measured: tests/test_gpu_network.py,
// the cfg2 / cfg4 128^3 golden tests run in this mode).  Activations, gradients, GroupNorm and the losses stay fp32.
// Caveat: an infinite input gives NaN, not +-inf (inf * w_lo with w_lo == 0); finite data is unaffected.
//
//   conv_x3_kernel<1>      nn.Conv3d 3^3 forward and data gradient (components.py:8-9,44), Cin and Cout multiples of 16
//   conv_x3_kernel<2>      nn.ConvTranspose3d(k3,s2,p1,op1) data gradient (in = 2*out - 1 + tap)
//   conv_c1_x3_kernel      the first layer (Cin = 1): contraction over the 27 taps
//   convt_x3_kernel        nn.ConvTranspose3d forward + bias + skip (components.py:259-264,283-284), output-parity classes
//   wgrad_x3_kernel        conv weight gradient (contraction over voxels, both operands through ds_read_b64_tr_b16)
//   convt_wgrad_x3_kernel  ConvTranspose3d weight gradient (output-parity planes of dy, 64-channel blocks of x)
//
// Weights: the packed buffer's bf16 fragment images [cb][kc][tap][k-half][32 co][8 ci] (conv_mfma.hip, pack_mfma_body) hold the
// high halves; mednet_conv3d_pack_elt(MEDNET_F32) also writes the low halves `lo_delta` bytes behind them, same order, so a
// weight slice of one 16-channel chunk is two linear 27 KB copies.
// Forward-type kernels are persistent (one workgroup per CU, each XCD walks a contiguous range of bricks); a step = one
// 16-channel chunk of one brick: the chunk's input halo (fp32 rows -> hi / lo planes [hl][k-half][voxel] of 16-byte pieces)
// and weight slice are committed to LDS, then 27 taps x N-tiles x 3 MFMAs run while the NEXT step's global loads -- dealt out
// one per tap, the next brick's first chunk included -- are in flight into registers.
#include "conv.h"

namespace mednet {

typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

#define X3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

struct HiLo {
  bf16x8 hi, lo;
};
// 8 fp32 (two 16-byte loads) -> 8 high + 8 low bf16
__device__ __forceinline__ HiLo split8(u32x4 ua, u32x4 ub) {
  const f4 a = __builtin_bit_cast(f4, ua), b = __builtin_bit_cast(f4, ub);
  HiLo r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bf16 ha = (bf16)a[j], hb = (bf16)b[j];
    r.hi[j] = ha;
    r.hi[j + 4] = hb;
    r.lo[j] = (bf16)(a[j] - (float)ha);
    r.lo[j + 4] = (bf16)(b[j] - (float)hb);
  }
  return r;
}

constexpr unsigned X3_OOB = 0xFFFFFF00u;  // a buffer-load offset past every resource: the hardware returns zeros

// ================================================================================================== forward / data gradient
template <int STRIDE>
struct X3Tile;
template <>
struct X3Tile<1> {
  static constexpr int TZ = 4, TY = 8, TX = 16, NWAVES = 8;
};
template <>
struct X3Tile<2> {
  static constexpr int TZ = 2, TY = 4, TX = 16, NWAVES = 4;
};

struct X3Args {
  const float* x;     // N x (id,ih,iw) x K, channels last, fp32
  const bf16* w_hi;   // image [ncb][nkc][27][2][32][8]; the low image lo_delta bytes behind it
  unsigned lo_delta, w_bytes;  // w_bytes: lo_delta + bytes of one image (one buffer resource spans both)
  const float* bias;  // nullable, M
  float* y;           // N x (od,oh,ow) x M
  float* stats;       // nullable: partial rows [n][rows][M][2] (see conv_x3_stats_rows): GroupNorm statistics {sum y, sum y^2}
                      // of the stored output, or -- with gn_y -- the first pass of a GroupNorm backward {sum du, sum du * gn_y}
  const float* add;   // nullable: a second gradient of the output tensor, summed before the store (data-gradient form)
  const float* gn_y;  // nullable (data-gradient form): the conv output the GroupNorm in FRONT of this layer normalised (shape of
                      // y); du = y_stored * act'(ca * gn_y + cb) with gn_coef[n][M] = {ca, cb}
  const float* gn_coef;
  int gn_act;
  int n, od, oh, ow, id, ih, iw, k, m;
  int tiles_z, tiles_y, tiles_x, tps, nkc, ncb, nitems, per_xcd, stats_rows, xps;  // xps: item ranges ("XCDs") per sample
  unsigned bytes_in;  // bytes of ONE sample of x (buffer resources are per sample: 32-bit offsets)
  int yb;             // > 0: z-walk of the bricks in columns of yb x tiles_x (see decode); 0: x fastest, then y, then z
};

template <int STRIDE>
__global__ __launch_bounds__(X3Tile<STRIDE>::NWAVES * 64) void conv_x3_kernel(X3Args a) {
  using G = X3Tile<STRIDE>;
  constexpr int TZ = G::TZ, TY = G::TY, TX = G::TX, NWAVES = G::NWAVES, NTHR = NWAVES * 64;
  constexpr int HZ = STRIDE * (TZ - 1) + 3, HY = STRIDE * (TY - 1) + 3, HX = STRIDE * (TX - 1) + 3;
  constexpr int NV = HZ * HY * HX;
  constexpr int NTW = TZ * TY * TX / 32 / NWAVES;
  constexpr int IN_PIECES = 2 * NV, IN_ROUNDS = (IN_PIECES + NTHR - 1) / NTHR;
  constexpr int W_SLICE = 27 * 2 * 32;  // 16-byte pieces of one image slice
  constexpr int W_PIECES = 2 * W_SLICE, W_ROUNDS = (W_PIECES + NTHR - 1) / NTHR;
  static_assert(IN_ROUNDS + W_ROUNDS <= 27, "one staging round per tap");
  static_assert(NTW >= 1 && NTW * NWAVES * 32 == TZ * TY * TX, "whole N-tiles per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* in_lds = reinterpret_cast<bf16x8*>(smem);  // [hl][k-half][NV]
  bf16x8* w_lds = in_lds + 4 * NV;                    // [hl][27][k-half][32 co]

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- staging plan of this thread (independent of the brick): halo piece p = it * NTHR + tid -> (voxel p >> 1, k-half p & 1)
  int rel[IN_ROUNDS], hzyx[IN_ROUNDS];
#pragma unroll
  for (int it = 0; it < IN_ROUNDS; ++it) {
    const int p = it * NTHR + tid;
    const int v = p >> 1;
    const int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
    rel[it] = ((hz * a.ih + hy) * a.iw + hx) * a.k * 4 + (p & 1) * 32;
    hzyx[it] = p < IN_PIECES ? (hz << 16) | (hy << 8) | hx : (0x4000 << 16);  // (past the halo: never inside a volume)
  }
  int lv[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int g = wv * NTW + t;
    const int lz = g / (TY / 2), ly = (g % (TY / 2)) * 2 + (r >> 4), lx = r & 15;
    lv[t] = ((STRIDE * lz) * HY + STRIDE * ly) * HX + STRIDE * lx + h * NV;
  }

  struct Item {
    int tz0, ty0, tx0, n, cb, base, valid;
    __amdgpu_buffer_rsrc_t rs;
  };
  // the i-th item of this workgroup: XCD (blockIdx & 7 under round-robin placement) x walks items [x * per_xcd, (x+1) * per_xcd)
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, slot_step = gridDim.x >> 3;
  auto decode = [&](int i) {
    Item it;
    const int slot = slot0 + i * slot_step;
    const int item = xcd * a.per_xcd + slot;
    it.valid = slot < a.per_xcd && item < a.nitems;
    // items are ordered (sample, channel block, brick): a workgroup's consecutive items stay in one (sample, channel block)
    // group as long as possible, which is what lets a wave keep per-channel sums in registers (conv_x3_stats_rows)
    const int iv = it.valid ? item : 0;
    const int grp = iv / a.tps;
    int tile = iv - grp * a.tps;
    it.n = grp / a.ncb;
    it.cb = grp - it.n * a.ncb;
    if (a.yb > 0) {
      // z-walk: the bricks an item range works on AT THE SAME TIME (its 32 workgroups take consecutive items) are one z-plane of a
      // column of a.yb x tiles_x bricks, and the next round is the plane above it: the two z-planes of halo that neighbours in z
      // share are then one round old (4 MB of input per round and XCD against 4 MB of L2) instead of tiles_y / yb rounds.  With
      // x fastest, y, then z the 32 -> 32 @128^3 launches fetched 1.8 - 2.0 x their input (rocprofv3 FETCH_SIZE).
      const int plane = a.yb * a.tiles_x, col = tile / (plane * a.tiles_z), rem = tile - col * plane * a.tiles_z;
      const int tz = rem / plane, within = rem - tz * plane;
      it.tx0 = (within % a.tiles_x) * TX;
      it.ty0 = (col * a.yb + within / a.tiles_x) * TY;
      it.tz0 = tz * TZ;
    } else {
      it.tx0 = (tile % a.tiles_x) * TX;
      tile /= a.tiles_x;
      it.ty0 = (tile % a.tiles_y) * TY;
      it.tz0 = (tile / a.tiles_y) * TZ;
    }
    it.base = (((STRIDE * it.tz0 - 1) * a.ih + (STRIDE * it.ty0 - 1)) * a.iw + (STRIDE * it.tx0 - 1)) * a.k * 4;
    it.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)it.n * a.id * a.ih * a.iw * a.k), 0, a.bytes_in, 0x00020000);
    return it;
  };

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_hi, 0, a.w_bytes, 0x00020000);
  u32x4 in_reg[IN_ROUNDS][2], w_reg[W_ROUNDS];
  auto fetch_round = [&](int j, const Item& it, int kc) {
    if (j < IN_ROUNDS) {
      const int gz = STRIDE * it.tz0 - 1 + (hzyx[j] >> 16), gy = STRIDE * it.ty0 - 1 + ((hzyx[j] >> 8) & 255),
                gx = STRIDE * it.tx0 - 1 + (hzyx[j] & 255);
      const bool ok = ((unsigned)gz < (unsigned)a.id) & ((unsigned)gy < (unsigned)a.ih) & ((unsigned)gx < (unsigned)a.iw) & (it.valid != 0);
      const unsigned off = ok ? (unsigned)(it.base + rel[j] + kc * 64) : X3_OOB;
      in_reg[j][0] = __builtin_amdgcn_raw_buffer_load_b128(it.rs, off, 0, 0);
      in_reg[j][1] = __builtin_amdgcn_raw_buffer_load_b128(it.rs, off + 16u, 0, 0);
    } else if (j < IN_ROUNDS + W_ROUNDS) {  // (branch-free: one resource spans the high and the low image)
      const int jw = j - IN_ROUNDS;
      const int q = jw * NTHR + tid;
      const int hl = q >= W_SLICE;
      const unsigned off = (unsigned)((it.cb * a.nkc + kc) * W_SLICE + (q - hl * W_SLICE)) * 16u + (hl ? a.lo_delta : 0u);
      w_reg[jw] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, ((q < W_PIECES) & (it.valid != 0)) ? off : X3_OOB, 0, 0);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) {
      const int p = it * NTHR + tid;
      if (p < IN_PIECES) {
        const HiLo s = split8(in_reg[it][0], in_reg[it][1]);
        in_lds[(p & 1) * NV + (p >> 1)] = s.hi;
        in_lds[(2 + (p & 1)) * NV + (p >> 1)] = s.lo;
      }
    }
#pragma unroll
    for (int jw = 0; jw < W_ROUNDS; ++jw) {
      const int q = jw * NTHR + tid;
      if (q < W_PIECES) w_lds[q] = __builtin_bit_cast(bf16x8, w_reg[jw]);
    }
  };

  f32x16 acc[NTW];
  auto init_acc = [&]() {  // (the bias is added at the store)
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  };
  // ---- epilogue.  The accumulator layout gives a lane four 16-byte pieces (channels 8q + 4h ..) of ITS voxel's row: stored as
  // they stand, one instruction touches 64 rows with 16 bytes each (measured: the store-bound first-layer kernel wrote at
  // 0.85 TB/s that way).  Stride 1: a tile goes through 4 KB of LDS private to the wave (behind the images; 16-byte pieces
  // XOR-swizzled by voxel, so the writes and the reads are conflict-free and no barrier is needed: a wave's LDS operations
  // execute in order) and leaves as whole rows -- 8 lanes per voxel, 8 voxels = 1 KB contiguous per instruction.  A lane then
  // owns the SAME 4 channels (piece c = lane & 7) of every voxel it stores, which is also what the per-channel sums want:
  // 8 registers instead of 32, reduced over the lanes of a class with DPP (lane_class_sum<8>), no LDS-queue shuffles.
  // (-DMEDNET_X3_EPI_LDS=0 restores the direct stores: profiles/r03_ab.md has the A/B, 3.4 ms per step.)
#ifndef MEDNET_X3_EPI_LDS
#define MEDNET_X3_EPI_LDS 1
#endif
#ifndef MEDNET_X3_EPI_LDS_S2
#define MEDNET_X3_EPI_LDS_S2 1
#endif
  constexpr bool EPI_LDS = MEDNET_X3_EPI_LDS && (STRIDE == 1 || MEDNET_X3_EPI_LDS_S2);
  // stride 1: the areas sit behind the images; stride 2 (halo 5x9x33: 150 KB of images) has no room: there they alias the input
  // planes, behind one more barrier per item (EPI_ALIAS)
  constexpr bool EPI_ALIAS = EPI_LDS && STRIDE == 2;
  [[maybe_unused]] f4* epi = reinterpret_cast<f4*>(smem + (EPI_ALIAS ? (size_t)0 : (size_t)(4 * NV + 2 * W_SLICE) * 16)) + wv * 256;
  constexpr int NS = EPI_LDS ? 4 : 16;  // channels a lane keeps sums of
  float ssum[NS], ssq[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) ssum[i] = ssq[i] = 0.f;
  int stats_n = -1, stats_cb = 0;
  auto flush_stats = [&]() {
    // one row per wave of the workgroups that work on this (sample, channel block) group (conv_x3_stats_rows)
    const int row = ((xcd % (a.xps > 0 ? a.xps : 1)) * slot_step + slot0) * NWAVES + wv;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      float s1, s2;
      int co;
      bool writer;
      if constexpr (EPI_LDS) {
        s1 = lane_class_sum<8>(ssum[i]);
        s2 = lane_class_sum<8>(ssq[i]);
        co = stats_cb * 32 + 4 * (lane & 7) + i;
        writer = lane < 8;
      } else {
        s1 = ssum[i];
        s2 = ssq[i];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          s1 += __shfl_xor(s1, o);
          s2 += __shfl_xor(s2, o);
        }
        co = stats_cb * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
        writer = r == 0;
      }
      if (writer && stats_n >= 0 && co < a.m) {
        float* dst = a.stats + (((size_t)stats_n * a.stats_rows + row) * a.m + co) * 2;
        dst[0] = s1;
        dst[1] = s2;
      }
      ssum[i] = ssq[i] = 0.f;
    }
  };
  // what happens to a 16-byte piece `o` of the output row at element offset `eo` (channels co0 .. co0 + 3): bias, summed second
  // gradient, store, and the sums (GroupNorm statistics, or the first pass of the previous GroupNorm's backward)
  // (`pre`: the piece's second operands are already in registers -- see store())
  auto finish_piece = [&](f4 o, size_t eo, int co0, const Item& it, int si, bool pre = false, f4 pre_add = f4{0.f, 0.f, 0.f, 0.f},
                          f4 pre_gy = f4{0.f, 0.f, 0.f, 0.f}) {
    if (a.bias) o += *reinterpret_cast<const f4*>(a.bias + co0);
    if (a.add) o += pre ? pre_add : __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.add + eo));
    __builtin_nontemporal_store(o, reinterpret_cast<f4*>(a.y + eo));
    if (a.gn_y) {
      const f4 gy = pre ? pre_gy : *reinterpret_cast<const f4*>(a.gn_y + eo);
      const float* cf = a.gn_coef + ((size_t)it.n * a.m + co0) * 2;
      const f4 c01 = *reinterpret_cast<const f4*>(cf), c23 = *reinterpret_cast<const f4*>(cf + 4);
      float u[4] = {fmaf(c01[0], gy[0], c01[1]), fmaf(c01[2], gy[1], c01[3]), fmaf(c23[0], gy[2], c23[1]), fmaf(c23[2], gy[3], c23[3])};
      float du[4] = {o[0], o[1], o[2], o[3]};
      act_grad_pre_n<4>(du, u, a.gn_act);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ssum[si + j] += du[j];
        ssq[si + j] = fmaf(du[j], gy[j], ssq[si + j]);
      }
    } else if (a.stats) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ssum[si + j] += o[j];
        ssq[si + j] = fmaf(o[j], o[j], ssq[si + j]);
      }
    }
  };
  auto store = [&](const Item& it) {
    const size_t ovol = (size_t)a.od * a.oh * a.ow;
    if (a.stats && (it.n != stats_n || it.cb != stats_cb)) {
      if (stats_n >= 0) flush_stats();
      stats_n = it.n;
      stats_cb = it.cb;
    }
    // Data-gradient forms: the rows of the second operands (summed second gradient, the GroupNorm's input) that this wave's
    // tiles need are requested HERE, before the first tile goes through LDS: the wave fences below pin every later load
    // behind them, and a load issued where it is used costs a whole HBM latency per tile with all 8 waves of the workgroup in
    // their epilogues at the same time (measured: 1.18 ms plain, 1.62 / 1.85 ms with one / both second operands at 32->32 @128^3).
    [[maybe_unused]] f4 pre_add[NTW][4], pre_gy[NTW][4];
    if constexpr (EPI_LDS) {
      if (a.add || a.gn_y) {
        const int c = lane & 7, co0 = it.cb * 32 + 4 * c;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const int g = wv * NTW + t;
          const int oz = it.tz0 + g / (TY / 2);
#pragma unroll
          for (int rd = 0; rd < 4; ++rd) {
            const int v = rd * 8 + (lane >> 3);
            const int oy = it.ty0 + (g % (TY / 2)) * 2 + (v >> 4), ox = it.tx0 + (v & 15);
            const bool ok = oz < a.od && oy < a.oh && ox < a.ow && co0 < a.m;
            const size_t eo = ((size_t)it.n * ovol + ((size_t)oz * a.oh + oy) * a.ow + ox) * a.m + co0;
            pre_add[t][rd] = pre_gy[t][rd] = f4{0.f, 0.f, 0.f, 0.f};
            if (ok && a.add) pre_add[t][rd] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.add + eo));
            if (ok && a.gn_y) pre_gy[t][rd] = *reinterpret_cast<const f4*>(a.gn_y + eo);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int g = wv * NTW + t;
      const int oz = it.tz0 + g / (TY / 2);
      if constexpr (EPI_LDS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // lane (voxel r of the tile, k-half h) -> piece 2q + h of row r
          const f4 o = {acc[t][q * 4], acc[t][q * 4 + 1], acc[t][q * 4 + 2], acc[t][q * 4 + 3]};
          epi[r * 8 + ((2 * q + h) ^ (r & 7))] = o;
        }
        wave_lds_fence();
        const int c = lane & 7, co0 = it.cb * 32 + 4 * c;
        f4 rows4[4];
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) rows4[rd] = epi[(rd * 8 + (lane >> 3)) * 8 + (c ^ ((rd * 8 + (lane >> 3)) & 7))];
        wave_lds_fence();
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {  // 8 voxels per instruction: voxel v of the tile = two x-rows of 16
          const int v = rd * 8 + (lane >> 3);
          const f4 o = rows4[rd];
          const int oy = it.ty0 + (g % (TY / 2)) * 2 + (v >> 4), ox = it.tx0 + (v & 15);
          if (oz < a.od && oy < a.oh && ox < a.ow && co0 < a.m)
            finish_piece(o, ((size_t)it.n * ovol + ((size_t)oz * a.oh + oy) * a.ow + ox) * a.m + co0, co0, it, 0, true, pre_add[t][rd],
                         pre_gy[t][rd]);
        }
      } else {
        const int oy = it.ty0 + (g % (TY / 2)) * 2 + (r >> 4), ox = it.tx0 + (r & 15);
        if (oz < a.od && oy < a.oh && ox < a.ow) {
          const size_t base = ((size_t)it.n * ovol + ((size_t)oz * a.oh + oy) * a.ow + ox) * a.m;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int co0 = it.cb * 32 + 8 * q + 4 * h;
            if (co0 < a.m) {
              const f4 o = {acc[t][q * 4], acc[t][q * 4 + 1], acc[t][q * 4 + 2], acc[t][q * 4 + 3]};
              finish_piece(o, base + co0, co0, it, q * 4);
            }
          }
        }
      }
    }
  };

  int idx = 0;
  Item cur = decode(0);
  if (!cur.valid) return;
  int kc = 0;
#pragma unroll
  for (int j = 0; j < IN_ROUNDS + W_ROUNDS; ++j) fetch_round(j, cur, 0);
  init_acc();
  for (;;) {
    const bool last_chunk = kc + 1 == a.nkc;
    Item nxt = cur;
    int nkc_ = kc + 1;
    if (last_chunk) {
      nxt = decode(++idx);
      nkc_ = 0;
    }
    __syncthreads();  // the previous step's operand reads are done
    commit();
    __syncthreads();
    // Two operand sets: the LDS reads of tap t+1 are in flight while the MFMAs of tap t run; the fence at the end of a tap keeps
    // the compiler from hoisting later taps' reads as well (it then spills: 27 taps x 6 reads are all independent).
    bf16x8 wa_hi[2], wa_lo[2], xh[2][NTW], xl[2][NTW];
    auto load_tap = [&](int tap, int b) {
      const int toff = ((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3;
      wa_hi[b] = w_lds[tap * 64 + h * 32 + r];
      wa_lo[b] = w_lds[W_SLICE + tap * 64 + h * 32 + r];
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        xh[b][t] = in_lds[lv[t] + toff];
        xl[b][t] = in_lds[2 * NV + lv[t] + toff];
      }
    };
    load_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int b = tap & 1;
      if (tap + 1 < 27) load_tap(tap + 1, b ^ 1);
      fetch_round(tap, nxt, nkc_);  // (rounds past IN_ROUNDS + W_ROUNDS are empty)
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        acc[t] = X3_MFMA(wa_lo[b], xh[b][t], acc[t]);
        acc[t] = X3_MFMA(wa_hi[b], xl[b][t], acc[t]);
        acc[t] = X3_MFMA(wa_hi[b], xh[b][t], acc[t]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (last_chunk) {
      if constexpr (EPI_ALIAS) __syncthreads();  // every wave is done with the images before the epilogue areas overwrite them
      store(cur);
      if (!nxt.valid) break;
      init_acc();
    }
    cur = nxt;
    kc = nkc_;
  }
  if (a.stats && stats_n >= 0) flush_stats();
}

static int x3_grid() {
  int cus = ::mednet_internal_cu_count();
  if (cus <= 0) cus = 256;
  return (cus + 7) / 8 * 8;
}
// workgroups of a weight-gradient launch: about one per CU, or what the caller asks for (`workgroups` > 0: the trainer, when the
// launch runs beside the main stream; see wgrad2_plan in conv_mfma.hip)
static int x3_wgrad_target(int workgroups) { return workgroups > 0 ? workgroups : x3_grid(); }

bool conv_x3_enabled() { return tuning_option("x3", 1) != 0; }
bool conv_x3_supported(int cin, int cout, int ksize) { return ksize == 3 && cin % 16 == 0 && cout % 16 == 0; }
// the kernels address one sample through a buffer resource with 32-bit byte offsets
bool conv_x3_fits(int d, int h, int w, int c) { return (double)d * h * w * c * 4.0 < 4294960000.0; }

// Partial rows per sample written by the forward-type kernel (0: not for this shape).  Each wave keeps per-channel sums of
// everything it stores for one (sample, channel block) group and writes ONE row when the group changes / at the end: row =
// (index of the workgroup among those that work on the group) * waves + wave, filled for the group's 32 channels.  Every cell
// [n][row][channel] must be written exactly once, so every workgroup of a group's item range(s) needs at least one brick of the
// group: with G = n * ncb groups over the 8 item ranges, either G divides 8 (a group spans 8 / G ranges, each workgroup one
// group) or 8 divides G with >= grid / 8 bricks per group (a range holds G / 8 whole groups, each workgroup touches them all).
static int x3_stats_xps(int n, int ncb, int tps, int grid) {  // ranges per group (>= 1), 0: not applicable
  const int G = n * ncb, per_range = grid / 8;
  if ((long long)G * tps % 8 != 0) return 0;
  if (G <= 8 && 8 % G == 0) return (long long)G * tps / 8 >= per_range ? 8 / G : 0;
  if (G % 8 == 0) return tps >= per_range ? 1 : 0;
  return 0;
}
int conv_x3_stats_rows(int n, int d, int h, int w, int cout) {
  using G = X3Tile<1>;
  if (!tuning_option("x3_stats", 1)) return 0;
  const int tps = ((d + G::TZ - 1) / G::TZ) * ((h + G::TY - 1) / G::TY) * ((w + G::TX - 1) / G::TX);
  const int grid = x3_grid();
  const int xps = x3_stats_xps(n, (cout + 31) / 32, tps, grid);
  return xps * (grid / 8) * G::NWAVES;
}

template <int STRIDE>
static int launch_x3(const void* x, const void* sec_hi, size_t lo_delta, const float* bias, void* y, int n, int od, int oh, int ow,
                     int id, int ih, int iw, int k, int m, float* stats, hipStream_t s, const void* add = nullptr,
                     const void* gn_y = nullptr, const float* gn_coef = nullptr, int gn_act = MEDNET_ACT_NONE) {
  using G = X3Tile<STRIDE>;
  constexpr int HZ = STRIDE * (G::TZ - 1) + 3, HY = STRIDE * (G::TY - 1) + 3, HX = STRIDE * (G::TX - 1) + 3;
  constexpr size_t lds = ((size_t)4 * HZ * HY * HX + 2 * 27 * 2 * 32) * 16 + (STRIDE == 1 ? (size_t)G::NWAVES * 4096 : 0);  // + epilogue areas
  static_assert(lds <= 160 * 1024, "one workgroup per CU");
  MEDNET_REQUIRE(k % 16 == 0 && m % 16 == 0 && lo_delta != 0, MEDNET_E_UNSUPPORTED, "conv_x3: channels %d -> %d", k, m);
  X3Args a;
  a.x = (const float*)x;
  a.w_hi = (const bf16*)sec_hi;
  const size_t img = (size_t)27 * ((m + 31) / 32 * 32) * k * 2;
  MEDNET_REQUIRE(lo_delta + img < 4294960000.0, MEDNET_E_UNSUPPORTED, "conv_x3: weight images too large");
  a.lo_delta = (unsigned)lo_delta;
  a.w_bytes = (unsigned)(lo_delta + img);
  a.bias = bias;
  a.y = (float*)y;
  a.stats = stats;
  a.n = n; a.od = od; a.oh = oh; a.ow = ow; a.id = id; a.ih = ih; a.iw = iw; a.k = k; a.m = m;
  a.tiles_z = (od + G::TZ - 1) / G::TZ;
  a.tiles_y = (oh + G::TY - 1) / G::TY;
  a.tiles_x = (ow + G::TX - 1) / G::TX;
  a.tps = a.tiles_z * a.tiles_y * a.tiles_x;
  a.nkc = k / 16;
  a.ncb = (m + 31) / 32;
  MEDNET_REQUIRE((double)n * a.tps * a.ncb < 2147483647.0, MEDNET_E_UNSUPPORTED, "conv_x3: too many bricks");
  a.nitems = n * a.tps * a.ncb;
  a.per_xcd = (a.nitems + 7) / 8;
  a.stats_rows = stats ? conv_x3_stats_rows(n, od, oh, ow, m) : 0;
  a.xps = stats ? x3_stats_xps(n, a.ncb, a.tps, x3_grid()) : 1;
  MEDNET_REQUIRE(!stats || (STRIDE == 1 && a.stats_rows > 0), MEDNET_E_UNSUPPORTED, "conv_x3: no fused sums for this shape");
  MEDNET_REQUIRE(!gn_y || (stats && gn_coef), MEDNET_E_SHAPE, "conv_x3: the GroupNorm-backward form needs gn_coef and the partial rows");
  a.add = (const float*)add;
  a.gn_y = (const float*)gn_y;
  a.gn_coef = gn_coef;
  a.gn_act = gn_act;
  a.bytes_in = (unsigned)((size_t)id * ih * iw * k * 4);
  // z-walk when a plane of (32 / tiles_x) x tiles_x bricks is what an item range's 32 workgroups hold at a time
  a.yb = 0;
  if (tuning_option("x3_zwalk", 1) && a.tiles_z > 1 && a.tiles_x <= 32 && 32 % a.tiles_x == 0) {
    const int yb = 32 / a.tiles_x < a.tiles_y ? 32 / a.tiles_x : a.tiles_y;
    if (a.tiles_y % yb == 0) a.yb = yb;
  }
  static bool attr_set[3] = {false, false, false};
  if (!attr_set[STRIDE]) {
    if (hipFuncSetAttribute((const void*)conv_x3_kernel<STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(MEDNET_E_HIP, "conv_x3: cannot raise dynamic LDS to %zu", lds);
    attr_set[STRIDE] = true;
  }
  hipLaunchKernelGGL((conv_x3_kernel<STRIDE>), dim3(x3_grid()), dim3(G::NWAVES * 64), lds, s, a);
  return check_launch("conv_x3");
}

int launch_conv_x3(const void* x, const void* sec_hi, size_t lo_delta, const float* bias, void* y, int n, int d, int h, int w, int k,
                   int m, float* stats, hipStream_t s) {
  return launch_x3<1>(x, sec_hi, lo_delta, bias, y, n, d, h, w, d, h, w, k, m, stats, s);
}
// data gradient (x = dy with K = the layer's Cout channels, y = dx with M = its Cin) + a second gradient of dx's tensor summed in
// (nullable) + the first pass of the backward of the GroupNorm in front of the layer (gn_y nullable; partial rows as above)
int launch_conv_x3_dgrad(const void* dy, const void* sec_hi, size_t lo_delta, void* dx, int n, int d, int h, int w, int k, int m,
                         const void* add, const void* gn_y, const float* gn_coef, int gn_act, float* gn_partial, hipStream_t s) {
  return launch_x3<1>(dy, sec_hi, lo_delta, nullptr, dx, n, d, h, w, d, h, w, k, m, gn_partial, s, add, gn_y, gn_coef, gn_act);
}
int launch_convt_dgrad_x3(const void* dy, const void* sec_hi, size_t lo_delta, void* dx, int n, int d, int h, int w, int cin,
                          int cout, hipStream_t s) {
  // dx (d,h,w; Cin) <- dy (2d,2h,2w; Cout): contraction over Cout
  return launch_x3<2>(dy, sec_hi, lo_delta, nullptr, dx, n, d, h, w, 2 * d, 2 * h, 2 * w, cout, cin, nullptr, s);
}

// ================================================================================================== first layer (Cin = 1)
// model.py:171-174: the network input has ONE channel, so the contraction runs over the 27 taps (padded to 32 = two k-steps):
//   D[co][voxel] = sum_tap W[tap][co] * x[voxel + tap - 1].   A = the weights (hi / lo fragments in registers for the whole
// workgroup), B = each lane's own 16 input values gathered from the fp32 halo brick in LDS and split on the fly.  The kernel is
// bound by writing y once (fp32); one statistics row per wave and brick when the GroupNorm partials are asked for.
struct C1X3Args {
  const float* x;  // N x (d,h,w), one channel
  const float* wt;  // Pf[27][1][cout]
  const float* bias;
  float* y;        // N x (d,h,w) x cout
  float* stats;    // nullable: [n][4 * bricks per sample][cout][2]
  int n, d, h, w, cout, tiles_z, tiles_y, tiles_x, tps, ncb;
};

__global__ __launch_bounds__(256) void conv_c1_x3_kernel(C1X3Args a) {
  constexpr int TZ = 4, TY = 8, TX = 16, HZ = 6, HY = 10, HX = 18, NV = HZ * HY * HX, NTW = 4;
  __shared__ float in_lds[NV + 8];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x / a.ncb, cb = blockIdx.x % a.ncb;
  const int n = tile / a.tps;
  int tt = tile - n * a.tps;
  const int tx0 = (tt % a.tiles_x) * TX;
  tt /= a.tiles_x;
  const int ty0 = (tt % a.tiles_y) * TY;
  const int tz0 = (tt / a.tiles_y) * TZ;
  const float* xs = a.x + (size_t)n * a.d * a.h * a.w;
  for (int v = tid; v < NV; v += 256) {
    const int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
    const int gz = tz0 - 1 + hz, gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    const bool ok = ((unsigned)gz < (unsigned)a.d) & ((unsigned)gy < (unsigned)a.h) & ((unsigned)gx < (unsigned)a.w);
    in_lds[v] = ok ? xs[((size_t)gz * a.h + gy) * a.w + gx] : 0.f;
  }
  // weights: k-step s, this lane's taps 16 s + 8 h + j of output channel cb * 32 + r
  bf16x8 wa_hi[2], wa_lo[2];
  int toff[2][8];  // LDS offsets of those taps (taps 27..31 do not exist: weight zero, offset 0)
  const int co = cb * 32 + r;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tap = 16 * s2 + 8 * h + j;
      const float wv_f = (tap < 27 && co < a.cout) ? a.wt[(size_t)tap * a.cout + co] : 0.f;
      const bf16 hi = (bf16)wv_f;
      wa_hi[s2][j] = hi;
      wa_lo[s2][j] = (bf16)(wv_f - (float)hi);
      toff[s2][j] = tap < 27 ? ((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3 : 0;
    }
  __syncthreads();
  // epilogue through 4 KB of LDS private to the wave, as in conv_x3_kernel: whole rows leave, a lane owns 4 channels
  __shared__ f4 epi_all[4 * 256];
  f4* epi = epi_all + wv * 256;
  float ssum[4], ssq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ssum[i] = ssq[i] = 0.f;
  const size_t ovol = (size_t)a.d * a.h * a.w;
  const int c = lane & 7, co0 = cb * 32 + 4 * c;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int g = wv * NTW + t;
    const int lz = g / (TY / 2), ly0 = (g % (TY / 2)) * 2;
    const int lvx = (lz * HY + ly0 + (r >> 4)) * HX + (r & 15);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 xh, xl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = in_lds[lvx + toff[s2][j]];
        const bf16 hi = (bf16)v;
        xh[j] = hi;
        xl[j] = (bf16)(v - (float)hi);
      }
      acc = X3_MFMA(wa_lo[s2], xh, acc);
      acc = X3_MFMA(wa_hi[s2], xl, acc);
      acc = X3_MFMA(wa_hi[s2], xh, acc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f4 o = {acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
      epi[r * 8 + ((2 * q + h) ^ (r & 7))] = o;
    }
    wave_lds_fence();
    const int oz = tz0 + lz;
    f4 rows4[4];
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) rows4[rd] = epi[(rd * 8 + (lane >> 3)) * 8 + (c ^ ((rd * 8 + (lane >> 3)) & 7))];
    wave_lds_fence();
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      const int v = rd * 8 + (lane >> 3);
      f4 o = rows4[rd];
      const int oy = ty0 + ly0 + (v >> 4), ox = tx0 + (v & 15);
      if (oz < a.d && oy < a.h && ox < a.w && co0 < a.cout) {
        if (a.bias) o += *reinterpret_cast<const f4*>(a.bias + co0);
        __builtin_nontemporal_store(o, reinterpret_cast<f4*>(a.y + ((size_t)n * ovol + ((size_t)oz * a.h + oy) * a.w + ox) * a.cout + co0));
        if (a.stats) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            ssum[j] += o[j];
            ssq[j] = fmaf(o[j], o[j], ssq[j]);
          }
        }
      }
    }
  }
  if (a.stats) {  // one row per wave and brick: the lanes of a class (lane & 7) hold the same 4 channels
    const int row = (tile - n * a.tps) * 4 + wv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float s1 = lane_class_sum<8>(ssum[i]), s2 = lane_class_sum<8>(ssq[i]);
      if (lane < 8 && co0 + i < a.cout) {
        float* dst = a.stats + (((size_t)n * (4 * a.tps) + row) * a.cout + co0 + i) * 2;
        dst[0] = s1;
        dst[1] = s2;
      }
    }
  }
}

bool conv_c1_x3_supported(int cin, int cout, int ksize) { return cin == 1 && ksize == 3 && cout % 4 == 0; }
int conv_c1_x3_stats_rows(int d, int h, int w) { return 4 * ((d + 3) / 4) * ((h + 7) / 8) * ((w + 15) / 16); }
int launch_conv_c1_x3(const void* x, const float* w_pf, const float* bias, void* y, int n, int d, int h, int w, int cout,
                      float* stats, hipStream_t s) {
  C1X3Args a;
  a.x = (const float*)x; a.wt = w_pf; a.bias = bias; a.y = (float*)y; a.stats = stats;
  a.n = n; a.d = d; a.h = h; a.w = w; a.cout = cout;
  a.tiles_z = (d + 3) / 4; a.tiles_y = (h + 7) / 8; a.tiles_x = (w + 15) / 16;
  a.tps = a.tiles_z * a.tiles_y * a.tiles_x;
  a.ncb = (cout + 31) / 32;
  MEDNET_REQUIRE((double)n * a.tps * a.ncb < 2147483647.0, MEDNET_E_UNSUPPORTED, "conv_c1_x3: grid too large");
  hipLaunchKernelGGL(conv_c1_x3_kernel, dim3((unsigned)(n * a.tps * a.ncb)), dim3(256), 0, s, a);
  return check_launch("conv_c1_x3");
}

// ================================================================================================== ConvTranspose3d forward
// out[2j + p] = bias + skip + sum over the taps k of parity class p of W[k] * x[j + delta_k]   (per dim: k=1 -> p=0,d=0;
// k=0 -> p=1,d=1; k=2 -> p=1,d=0).  A workgroup owns a 2x4x16 brick of INPUT voxels (4x8x32 outputs), a wave one N-tile of 32
// input voxels with all 8 output parity classes in registers (8 x 16 accumulators), as convt_f32_mfma_kernel; operands split
// as above.  Not persistent: the launch has thousands of short workgroups, two per CU.
struct CtX3Args {
  const float* x;
  const bf16* w_hi;  // forward image of the layer (mode 2: Weff[co][ci][tap] = Wt[ci][co][tap]); low image lo_delta behind
  unsigned lo_delta, w_bytes;
  const float* bias;
  const float* skip;
  float* y;
  int n, id, ih, iw, k, m;
  int tiles_z, tiles_y, tiles_x, ntiles, nkc, ncb;
};

__global__ __launch_bounds__(256, 2) void convt_x3_kernel(CtX3Args a) {
  constexpr int TZ = 2, TY = 4, TX = 16, HZ = TZ + 1, HY = TY + 1, HX = TX + 1, NV = HZ * HY * HX;
  constexpr int IN_PIECES = 2 * NV, IN_ROUNDS = (IN_PIECES + 255) / 256;
  constexpr int W_SLICE = 27 * 2 * 32, W_PIECES = 2 * W_SLICE, W_ROUNDS = (W_PIECES + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* in_lds = reinterpret_cast<bf16x8*>(smem);  // [hl][k-half][NV]
  bf16x8* w_lds = in_lds + 4 * NV;                    // [hl][27][k-half][32 co]
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x / a.ncb, cb = blockIdx.x % a.ncb;
  int tt = tile;
  const int tx0 = (tt % a.tiles_x) * TX;
  tt /= a.tiles_x;
  const int ty0 = (tt % a.tiles_y) * TY;
  tt /= a.tiles_y;
  const int tz0 = (tt % a.tiles_z) * TZ;
  const int n = tt / a.tiles_z;
  const float* xs = a.x + (size_t)n * a.id * a.ih * a.iw * a.k;
  long long goff[IN_ROUNDS];  // element offset of the piece's 8 channels (chunk 0), -1 outside the volume / past the halo
#pragma unroll
  for (int it = 0; it < IN_ROUNDS; ++it) {
    const int p = it * 256 + tid;
    const int v = p >> 1;
    long long off = -1;
    if (p < IN_PIECES) {
      const int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
      const int gz = tz0 + hz, gy = ty0 + hy, gx = tx0 + hx;
      if (gz < a.id && gy < a.ih && gx < a.iw) off = (((long long)gz * a.ih + gy) * a.iw + gx) * a.k + (p & 1) * 8;
    }
    goff[it] = off;
  }
  const int lz = wv / (TY / 2), ly = (wv % (TY / 2)) * 2 + (r >> 4), lx = r & 15;
  const int lbase = (lz * HY + ly) * HX + lx + h * NV;
  f32x16 acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_hi, 0, a.w_bytes, 0x00020000);
  u32x4 in_reg[IN_ROUNDS][2], w_reg[W_ROUNDS];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      in_reg[it][0] = in_reg[it][1] = z;
      if (goff[it] >= 0) {
        const float* src = xs + goff[it] + kc * 16;
        in_reg[it][0] = *reinterpret_cast<const u32x4*>(src);
        in_reg[it][1] = *reinterpret_cast<const u32x4*>(src + 4);
      }
    }
  };
  fetch(0);
  for (int kc = 0; kc < a.nkc; ++kc) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) {
      const int p = it * 256 + tid;
      if (p < IN_PIECES) {
        const HiLo s = split8(in_reg[it][0], in_reg[it][1]);
        in_lds[(p & 1) * NV + (p >> 1)] = s.hi;
        in_lds[(2 + (p & 1)) * NV + (p >> 1)] = s.lo;
      }
    }
    // the weight slice is copied here, not prefetched through registers (8 x 16 accumulators leave no room for it); the
    // CU's second workgroup covers the latency
#pragma unroll
    for (int jw = 0; jw < W_ROUNDS; ++jw) {
      const int q = jw * 256 + tid;
      const int hl = q >= W_SLICE;
      const unsigned off = (unsigned)((cb * a.nkc + kc) * W_SLICE + (q - hl * W_SLICE)) * 16u + (hl ? a.lo_delta : 0u);
      w_reg[jw] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, q < W_PIECES ? off : X3_OOB, 0, 0);
    }
#pragma unroll
    for (int jw = 0; jw < W_ROUNDS; ++jw) {
      const int q = jw * 256 + tid;
      if (q < W_PIECES) w_lds[q] = __builtin_bit_cast(bf16x8, w_reg[jw]);
    }
    __syncthreads();
    if (kc + 1 < a.nkc) fetch(kc + 1);
    bf16x8 xh[8], xl[8];
#pragma unroll
    for (int dl = 0; dl < 8; ++dl) {
      const int o = lbase + (((dl >> 2) & 1) * HY + ((dl >> 1) & 1)) * HX + (dl & 1);
      xh[dl] = in_lds[o];
      xl[dl] = in_lds[2 * NV + o];
    }
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
      const int pc = (kz != 1) * 4 + (ky != 1) * 2 + (kx != 1);  // output parity class of this tap
      const int dl = (kz == 0) * 4 + (ky == 0) * 2 + (kx == 0);  // input offset of this tap
      const bf16x8 wa_hi = w_lds[tap * 64 + h * 32 + r];
      const bf16x8 wa_lo = w_lds[W_SLICE + tap * 64 + h * 32 + r];
      acc[pc] = X3_MFMA(wa_lo, xh[dl], acc[pc]);
      acc[pc] = X3_MFMA(wa_hi, xl[dl], acc[pc]);
      acc[pc] = X3_MFMA(wa_hi, xh[dl], acc[pc]);
    }
  }
  // ---- epilogue: whole output rows through LDS (the images are dead: one barrier, then 4 KB per wave).  For an output parity
  // (pz, py) and one of the tile's two low-resolution y-rows the classes px = 0, 1 of 16 lanes interleave to 32 consecutive
  // output voxels of ONE output x-row: 4 KB contiguous per (pz, py, row) when Cout = 32, whole 128-byte segments otherwise.
  const int od = 2 * a.id, oh = 2 * a.ih, ow = 2 * a.iw;
  __syncthreads();
  f4* epi = reinterpret_cast<f4*>(smem) + wv * 256;
  const int c = lane & 7, co0 = cb * 32 + 4 * c;
  const int jz = tz0 + lz;
  // The encoder-feature rows of a (pzy, yy) group are requested ONE GROUP AHEAD, in front of the previous group's wave fences
  // (which pin every later load behind them): requested where they are added, each of the 8 groups waited a whole HBM latency.
  f4 sk_next[4];
  auto request_skip = [&](int pzy, int yy) {
    const int oz = 2 * jz + (pzy >> 1), oy = 2 * (ty0 + (wv % (TY / 2)) * 2 + yy) + (pzy & 1);
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      const int ox = 2 * tx0 + rd * 8 + (lane >> 3);
      sk_next[rd] = f4{0.f, 0.f, 0.f, 0.f};
      if (a.skip && oz < od && oy < oh && ox < ow && co0 < a.m)
        sk_next[rd] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.skip + ((((size_t)n * od + oz) * oh + oy) * ow + ox) * a.m + co0));
    }
  };
  request_skip(0, 0);
#pragma unroll
  for (int pzy = 0; pzy < 4; ++pzy) {
#pragma unroll
    for (int yy = 0; yy < 2; ++yy) {
      const f4 sk[4] = {sk_next[0], sk_next[1], sk_next[2], sk_next[3]};
      if (pzy * 2 + yy + 1 < 8) request_skip((pzy * 2 + yy + 1) >> 1, (pzy * 2 + yy + 1) & 1);
      if ((r >> 4) == yy) {
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const int ov = 2 * (r & 15) + px;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f4 o = {acc[pzy * 2 + px][q * 4], acc[pzy * 2 + px][q * 4 + 1], acc[pzy * 2 + px][q * 4 + 2], acc[pzy * 2 + px][q * 4 + 3]};
            epi[ov * 8 + ((2 * q + h) ^ (ov & 7))] = o;
          }
        }
      }
      wave_lds_fence();
      const int oz = 2 * jz + (pzy >> 1), oy = 2 * (ty0 + (wv % (TY / 2)) * 2 + yy) + (pzy & 1);
      f4 rows4[4];
#pragma unroll
      for (int rd = 0; rd < 4; ++rd) rows4[rd] = epi[(rd * 8 + (lane >> 3)) * 8 + (c ^ ((rd * 8 + (lane >> 3)) & 7))];
      wave_lds_fence();
#pragma unroll
      for (int rd = 0; rd < 4; ++rd) {
        const int v = rd * 8 + (lane >> 3);
        f4 o = rows4[rd];
        const int ox = 2 * tx0 + v;
        if (oz < od && oy < oh && ox < ow && co0 < a.m) {
          const size_t eo = ((((size_t)n * od + oz) * oh + oy) * ow + ox) * a.m + co0;
          if (a.bias) o += *reinterpret_cast<const f4*>(a.bias + co0);
          o += sk[rd];
          *reinterpret_cast<f4*>(a.y + eo) = o;
        }
      }
    }
  }
}

int launch_convt_fwd_x3(const void* x, const void* sec_hi, size_t lo_delta, const float* bias, const void* skip, void* y, int n,
                        int d, int h, int w, int cin, int cout, hipStream_t s) {
  constexpr size_t lds = ((size_t)4 * 3 * 5 * 17 + 2 * 27 * 2 * 32) * 16;
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  MEDNET_REQUIRE(cin % 16 == 0 && cout % 16 == 0 && lo_delta != 0, MEDNET_E_UNSUPPORTED, "convt_x3: channels %d -> %d", cin, cout);
  CtX3Args a;
  a.x = (const float*)x;
  a.w_hi = (const bf16*)sec_hi;
  const size_t img = (size_t)27 * ((cout + 31) / 32 * 32) * cin * 2;
  MEDNET_REQUIRE(lo_delta + img < 4294960000.0, MEDNET_E_UNSUPPORTED, "convt_x3: weight images too large");
  a.lo_delta = (unsigned)lo_delta;
  a.w_bytes = (unsigned)(lo_delta + img);
  a.bias = bias; a.skip = (const float*)skip; a.y = (float*)y;
  a.n = n; a.id = d; a.ih = h; a.iw = w; a.k = cin; a.m = cout;
  a.tiles_z = (d + 1) / 2; a.tiles_y = (h + 3) / 4; a.tiles_x = (w + 15) / 16;
  a.ntiles = n * a.tiles_z * a.tiles_y * a.tiles_x;
  a.nkc = cin / 16;
  a.ncb = (cout + 31) / 32;
  MEDNET_REQUIRE((double)a.ntiles * a.ncb < 2147483647.0, MEDNET_E_UNSUPPORTED, "convt_x3: grid too large");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)convt_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(MEDNET_E_HIP, "convt_x3: cannot raise dynamic LDS to %zu", lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(convt_x3_kernel, dim3((unsigned)(a.ntiles * a.ncb)), dim3(256), lds, s, a);
  return check_launch("convt_x3");
}

// ================================================================================================== weight gradient
//   R[tap][a][b] = sum_v A[v][a] * B[v + tap - 1][b]      A = dy (Cout), B = x (Cin);   dw[(a*KB + b)*27 + tap]
// Brick 4x4x16 (halo 6x6x18); LDS rows [voxel][32 channels] bf16, a high and a low plane per operand; a k-step = 16
// x-consecutive voxels, both MFMA operands through the transposing LDS read (lane mapping: conv_mfma.hip tr_operand,
// tools/probes/tr_probe.hip).  Eight waves, two per SIMD (the partner covers a wave's LDS latency: with one wave per SIMD and two
// operand sets the wait for the older set also waited for most of the younger one -- lgkmcnt counts to 15); a wave owns 3-4 of
// the 27 taps (4 x 16 accumulators; waves 3..7 skip their 4th slot) and walks every k-step; the next
// brick's 8 staging rounds are dealt out one per k-step.  Per-workgroup slabs, fixed-order reduce (no atomics).
struct WgX3Args {
  const float* A;
  const float* B;
  float* part;  // [workgroup][27][32][32]
  int n, d, h, w, ka, kb;
  int tiles_z, tiles_y, tiles_x, tps, ntiles, nab, nbb, splits;
  unsigned bytesA, bytesB;  // per sample
};

__device__ __forceinline__ bf16x8 x3_tr_operand(const char* base, int second) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + second));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(512) void wgrad_x3_kernel(WgX3Args a) {
  constexpr int TZ = 4, TY = 4, TX = 16, HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
  constexpr int NA = TZ * TY * TX, NB = HZ * HY * HX;
  constexpr int NTHR = 512, TAPS = 4;  // 8 waves, two per SIMD; wave w owns taps w, w + 8, w + 16, w + 24
  constexpr int A_ROUNDS = NA * 4 / NTHR, B_ROUNDS = (NB * 4 + NTHR - 1) / NTHR;
  constexpr int KSTEPS = NA / 16;
  static_assert(A_ROUNDS + B_ROUNDS <= KSTEPS, "one staging round per k-step");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // bytes: A_hi [NA][64] | A_lo | B_hi [NB][64] | B_lo
  constexpr int A_LO = NA * 64, B_HI = 2 * NA * 64, B_LO = 2 * NA * 64 + NB * 64;

  const int pair = blockIdx.x / a.splits, split = blockIdx.x % a.splits;
  const int ab = pair / a.nbb, bb = pair % a.nbb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int tw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, hk = lane >> 5;
  const int coloff = (16 * (g & 1) + 4 * p) * 2;

  u32x4 regA[A_ROUNDS][2], regB[B_ROUNDS][2];
  struct Next {
    int tz0, ty0, tx0, valid;
    __amdgpu_buffer_rsrc_t rA, rB;
  };
  auto plan = [&](int tile) {
    Next nx;
    nx.valid = tile < a.ntiles;
    int tt = nx.valid ? tile : 0;
    const int n = tt / a.tps;
    tt -= n * a.tps;
    nx.tx0 = (tt % a.tiles_x) * TX;
    tt /= a.tiles_x;
    nx.ty0 = (tt % a.tiles_y) * TY;
    nx.tz0 = (tt / a.tiles_y) * TZ;
    const size_t svox = (size_t)n * a.d * a.h * a.w;
    nx.rA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + svox * a.ka), 0, a.bytesA, 0x00020000);
    nx.rB = __builtin_amdgcn_make_buffer_rsrc((void*)(a.B + svox * a.kb), 0, a.bytesB, 0x00020000);
    return nx;
  };
  auto fetch_round = [&](int j, const Next& nx) {
    if (j < A_ROUNDS) {
      const int c = j * NTHR + tid;
      const int v = c >> 2, part = c & 3;
      const int gz = nx.tz0 + v / (TX * TY), gy = nx.ty0 + (v / TX) % TY, gx = nx.tx0 + v % TX;
      const bool ok = (gz < a.d) & (gy < a.h) & (gx < a.w) & (ab * 32 + part * 8 < a.ka) & (nx.valid != 0);
      const unsigned off = ok ? ((unsigned)((gz * a.h + gy) * a.w + gx) * (unsigned)a.ka + ab * 32 + part * 8) * 4u : X3_OOB;
      regA[j][0] = __builtin_amdgcn_raw_buffer_load_b128(nx.rA, off, 0, 0);
      regA[j][1] = __builtin_amdgcn_raw_buffer_load_b128(nx.rA, off + 16u, 0, 0);
    } else if (j < A_ROUNDS + B_ROUNDS) {
      const int it = j - A_ROUNDS;
      const int c = it * NTHR + tid;
      const int v = c >> 2, part = c & 3;
      const int gz = nx.tz0 - 1 + v / (HX * HY), gy = nx.ty0 - 1 + (v / HX) % HY, gx = nx.tx0 - 1 + v % HX;
      const bool ok = (c < NB * 4) & ((unsigned)gz < (unsigned)a.d) & ((unsigned)gy < (unsigned)a.h) &
                      ((unsigned)gx < (unsigned)a.w) & (bb * 32 + part * 8 < a.kb) & (nx.valid != 0);
      const unsigned off = ok ? ((unsigned)((gz * a.h + gy) * a.w + gx) * (unsigned)a.kb + bb * 32 + part * 8) * 4u : X3_OOB;
      regB[it][0] = __builtin_amdgcn_raw_buffer_load_b128(nx.rB, off, 0, 0);
      regB[it][1] = __builtin_amdgcn_raw_buffer_load_b128(nx.rB, off + 16u, 0, 0);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < A_ROUNDS; ++it) {
      const int c = it * NTHR + tid;
      const HiLo s = split8(regA[it][0], regA[it][1]);
      *reinterpret_cast<bf16x8*>(smem + c * 16) = s.hi;
      *reinterpret_cast<bf16x8*>(smem + A_LO + c * 16) = s.lo;
    }
#pragma unroll
    for (int it = 0; it < B_ROUNDS; ++it) {
      const int c = it * NTHR + tid;
      if (c < NB * 4) {
        const HiLo s = split8(regB[it][0], regB[it][1]);
        *reinterpret_cast<bf16x8*>(smem + B_HI + c * 16) = s.hi;
        *reinterpret_cast<bf16x8*>(smem + B_LO + c * 16) = s.lo;
      }
    }
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int i = 0; i < TAPS; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const bool has4 = tw < 3;  // waves 0..2 own four taps, waves 3..7 three (per SIMD pair w, w + 4: 7, 7, 7, 6)
  int toff[TAPS];
#pragma unroll
  for (int i = 0; i < TAPS; ++i) {
    const int tap = tw + 8 * i < 27 ? tw + 8 * i : 26;
    toff[i] = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * 64;
  }
  const char* Ab = smem + coloff;
  const char* Bb = smem + B_HI + coloff;

  int tile = split;
  if (tile < a.ntiles) {
    const Next first = plan(tile);
#pragma unroll
    for (int j = 0; j < A_ROUNDS + B_ROUNDS; ++j) fetch_round(j, first);
  }
  for (; tile < a.ntiles; tile += a.splits) {
    __syncthreads();  // the previous brick is consumed
    commit();
    __syncthreads();
    const Next nx = plan(tile + a.splits);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const char* arow = Ab + (ks * TX + 8 * hk + q) * 64;
      const bf16x8 ah = x3_tr_operand(arow, 4 * 64);
      const bf16x8 al = x3_tr_operand(arow + A_LO, 4 * 64);
      const char* brow = Bb + (((ks / TY) * HY + ks % TY) * HX + 8 * hk + q) * 64;
      bf16x8 bh[TAPS], bl[TAPS];
#pragma unroll
      for (int i = 0; i < TAPS; ++i) {
        if (i < 3 || has4) {  // (wave-uniform: the 4th slot of waves 3..7 has no tap)
          bh[i] = x3_tr_operand(brow + toff[i], 4 * 64);
          bl[i] = x3_tr_operand(brow + (B_LO - B_HI) + toff[i], 4 * 64);
        }
      }
      fetch_round(ks, nx);  // the next brick's staging rounds, one per k-step
#pragma unroll
      for (int i = 0; i < TAPS; ++i) {
        if (i < 3 || has4) {
          acc[i] = X3_MFMA(al, bh[i], acc[i]);
          acc[i] = X3_MFMA(ah, bl[i], acc[i]);
          acc[i] = X3_MFMA(ah, bh[i], acc[i]);
        }
      }
    }
  }
  float* out = a.part + (size_t)blockIdx.x * 27 * 1024;
  const int col = lane & 31;
#pragma unroll
  for (int i = 0; i < TAPS; ++i) {
    const int tap = tw + 8 * i;
    if (tap < 27) {
#pragma unroll
      for (int j = 0; j < 16; ++j) out[((size_t)tap * 32 + (j & 3) + 8 * (j >> 2) + 4 * hk) * 32 + col] = acc[i][j];
    }
  }
}

// dw[(a*KB + b)*27 + tap] = sum over the splits of part[(pair*splits + split)][tap][a%32][b%32], fixed order
// (few slabs per output -- deep layers, where the channel-block pairs alone fill the chip: one thread per output)
__global__ __launch_bounds__(256) void wgrad_x3_reduce_kernel_few(const float* __restrict__ part, float* __restrict__ dw, int ka, int kb, int nbb,
                                               int splits) {
  const size_t total = (size_t)((ka + 31) / 32) * nbb * 1024 * 27;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int b32 = (int)(e % 32), a32 = (int)((e / 32) % 32), tap = (int)((e / 1024) % 27);
  const int pair = (int)(e / (1024 * 27));
  const int ab = pair / nbb, bb = pair % nbb;
  const float* src = part + ((size_t)pair * splits) * 27 * 1024 + (size_t)tap * 1024 + a32 * 32 + b32;
  float s0 = 0.f;
  for (int k = 0; k < splits; ++k) s0 += src[(size_t)k * 27 * 1024];
  if (ab * 32 + a32 < ka && bb * 32 + b32 < kb) dw[((size_t)(ab * 32 + a32) * kb + bb * 32 + b32) * 27 + tap] = s0;
}
__global__ __launch_bounds__(256) void wgrad_x3_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int ka, int kb,
                                                              int nbb, int splits) {
  // 64 outputs per workgroup, the slabs dealt to its 4 waves (wave g: slabs g, g + 4, ..., four loads in flight), the four sums
  // combined through LDS in a fixed order: one thread per output walking all `splits` slabs was a chain of up to 64 dependent
  // round trips on a third of the CUs (12.5 us per launch, 23 launches per step).
  __shared__ float sh[256];
  const size_t total = (size_t)((ka + 31) / 32) * nbb * 1024 * 27;
  const int g = threadIdx.x >> 6;
  const size_t e = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b32 = 0, a32 = 0, tap = 0, ab = 0, bb = 0;
  if (e < total) {
    // thread index enumerates [pair][tap][a32][b32] so reads are coalesced
    b32 = (int)(e % 32), a32 = (int)((e / 32) % 32), tap = (int)((e / 1024) % 27);
    const int pair = (int)(e / (1024 * 27));
    ab = pair / nbb, bb = pair % nbb;
    const float* src = part + ((size_t)pair * splits) * 27 * 1024 + (size_t)tap * 1024 + a32 * 32 + b32;
    const size_t slab = (size_t)27 * 1024;
    int k = g;
    for (; k + 12 < splits; k += 16) {
      s0 += src[(size_t)k * slab];
      s1 += src[(size_t)(k + 4) * slab];
      s2 += src[(size_t)(k + 8) * slab];
      s3 += src[(size_t)(k + 12) * slab];
    }
    for (; k < splits; k += 4) s0 += src[(size_t)k * slab];
  }
  sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && e < total && ab * 32 + a32 < ka && bb * 32 + b32 < kb)
    dw[((size_t)(ab * 32 + a32) * kb + bb * 32 + b32) * 27 + tap] = (sh[threadIdx.x] + sh[64 + threadIdx.x]) + (sh[128 + threadIdx.x] + sh[192 + threadIdx.x]);
}

static void wgx3_plan(int n, int d, int h, int w, int ka, int kb, int workgroups, WgX3Args& a) {
  a.tiles_z = (d + 3) / 4;
  a.tiles_y = (h + 3) / 4;
  a.tiles_x = (w + 15) / 16;
  a.tps = a.tiles_z * a.tiles_y * a.tiles_x;
  a.ntiles = n * a.tps;
  a.nab = (ka + 31) / 32;
  a.nbb = (kb + 31) / 32;
  const int pairs = a.nab * a.nbb;
  int splits = (x3_wgrad_target(workgroups) + pairs - 1) / pairs;  // about one workgroup per CU (or per second CU beside the main stream)
  if (splits > a.ntiles) splits = a.ntiles;
  if (splits < 1) splits = 1;
  a.splits = splits;
}

size_t wgrad_x3_ws_bytes(int n, int d, int h, int w, int cin, int cout, int workgroups) {
  WgX3Args a;
  wgx3_plan(n, d, h, w, cout, cin, workgroups, a);
  return (size_t)a.nab * a.nbb * a.splits * 27 * 1024 * sizeof(float);
}

// ---- first-layer weight gradient (Cin = 1) of the fp32 storage mode ------------------------------------------------------
//   dW[co][tap] = sum_v x[v + tap - 1] * dy[v][co]:  D[tap (27 of 32 rows)][co] += A[tap][k = voxel] * B[k = voxel][co],
// as wgrad_c1_mfma_kernel (conv_mfma.hip) with both operands split: A gathered from the fp32 halo brick of x (lane = tap row: 8
// x-consecutive values of its shifted row), B = the fp32 dy brick committed to a high and a low bf16 plane and read through the
// transposing LDS read.  Bound by reading dy once (1.07 GB at config 2); the VALU kernel it replaces took 0.49 ms.
struct Wc1X3Args {
  const float* x;   // N x D x H x W
  const float* dy;  // N x D x H x W x 32
  float* part;      // [workgroup][32][27]
  int n, d, h, w;
  int tiles_z, tiles_y, tiles_x, ntiles;
  unsigned bytes_x, bytes_dy;  // per sample
};
__global__ __launch_bounds__(256, 2) void wgrad_c1_x3_kernel(Wc1X3Args a) {
  constexpr int TZ = 4, TY = 8, TX = 16, HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
  constexpr int NJ = TZ * TY * TX, NH = HZ * HY * HX;
  constexpr int XH_BYTES = (NH * 4 + 255) / 256 * 256;  // fp32 halo brick of x
  constexpr int DY_LO = NJ * 64;                        // bytes between the high and the low plane of dy
  constexpr int DY_ROUNDS = NJ * 4 / 256, X_ROUNDS = (NH + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xh = reinterpret_cast<float*>(smem);
  char* dyl = smem + XH_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hk = lane >> 5;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int coloff = (16 * (g & 1) + 4 * p) * 2;
  const int tapc = r < 27 ? r : 26;  // rows 27..31 duplicate tap 26 and are dropped at write-out
  const int abase = ((tapc / 9) * HY + (tapc / 3) % 3) * HX + tapc % 3 + 8 * hk;

  u32x4 rdy[DY_ROUNDS][2];
  float rx[X_ROUNDS];
  auto fetch = [&](int tile) {
    int tt = tile;
    const int tx0 = (tt % a.tiles_x) * TX;
    tt /= a.tiles_x;
    const int ty0 = (tt % a.tiles_y) * TY;
    tt /= a.tiles_y;
    const int tz0 = (tt % a.tiles_z) * TZ;
    const size_t svox = (size_t)(tt / a.tiles_z) * a.d * a.h * a.w;
    const auto rD = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + svox * 32), 0, a.bytes_dy, 0x00020000);
    const auto rX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + svox), 0, a.bytes_x, 0x00020000);
#pragma unroll
    for (int it = 0; it < DY_ROUNDS; ++it) {
      const int c = it * 256 + tid;
      const int part = c & 3, v = c >> 2;
      const int gz = tz0 + v / (TX * TY), gy = ty0 + (v / TX) % TY, gx = tx0 + v % TX;
      const bool in_vol = (gz < a.d) & (gy < a.h) & (gx < a.w);
      const unsigned off = in_vol ? ((unsigned)((gz * a.h + gy) * a.w + gx) * 32u + part * 8) * 4u : X3_OOB;
      rdy[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rD, off, 0, 0);
      rdy[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rD, off + 16u, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < X_ROUNDS; ++it) {
      const int v = it * 256 + tid;
      const int gz = tz0 - 1 + v / (HX * HY), gy = ty0 - 1 + (v / HX) % HY, gx = tx0 - 1 + v % HX;
      const bool in_vol = (v < NH) & ((unsigned)gz < (unsigned)a.d) & ((unsigned)gy < (unsigned)a.h) & ((unsigned)gx < (unsigned)a.w);
      rx[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, in_vol ? (unsigned)((gz * a.h + gy) * a.w + gx) * 4u : X3_OOB, 0, 0));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < DY_ROUNDS; ++it) {
      const int c = it * 256 + tid;
      const HiLo sp = split8(rdy[it][0], rdy[it][1]);
      *reinterpret_cast<bf16x8*>(dyl + c * 16) = sp.hi;
      *reinterpret_cast<bf16x8*>(dyl + DY_LO + c * 16) = sp.lo;
    }
#pragma unroll
    for (int it = 0; it < X_ROUNDS; ++it) {
      const int v = it * 256 + tid;
      if (v < NH) xh[v] = rx[it];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  int tile = blockIdx.x;
  if (tile < a.ntiles) fetch(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();  // previous brick fully consumed
    commit();
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);  // flies while this brick is worked on
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int row = wv * 8 + s8;  // (lz, ly) = (row / TY, row % TY): 16 x-consecutive voxels = one MFMA k-step
      const float* px = xh + abase + ((row / TY) * HY + row % TY) * HX;
      bf16x8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = px[j];
        hi[j] = (__bf16)xv;
        lo[j] = (__bf16)(xv - (float)hi[j]);
      }
      const char* brow = dyl + (row * TX + 8 * hk + q) * 64 + coloff;
      const bf16x8 bh = x3_tr_operand(brow, 4 * 64), bl = x3_tr_operand(brow + DY_LO, 4 * 64);
      acc = X3_MFMA(lo, bh, acc);
      acc = X3_MFMA(hi, bl, acc);
      acc = X3_MFMA(hi, bh, acc);
    }
  }
  // ---- sum the 4 waves in LDS (fixed order), write the workgroup's partial in dW layout [co][27]
  float* red = reinterpret_cast<float*>(smem);  // [4 waves][16][64 lanes]
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) red[(wv * 16 + j) * 64 + lane] = acc[j];
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float sum = (red[(0 * 16 + j) * 64 + lane] + red[(1 * 16 + j) * 64 + lane]) + (red[(2 * 16 + j) * 64 + lane] + red[(3 * 16 + j) * 64 + lane]);
      const int tap = (j & 3) + 8 * (j >> 2) + 4 * hk, co = lane & 31;
      if (tap < 27) a.part[((size_t)blockIdx.x * 32 + co) * 27 + tap] = sum;
    }
  }
}
bool wgrad_c1_x3_supported(int cout, int x_dtype, int dy_dtype) { return cout == 32 && x_dtype == MEDNET_F32 && dy_dtype == MEDNET_F32; }
int wgrad_c1_x3_blocks(int n, int d, int h, int w) {
  const int nt = n * ((d + 3) / 4) * ((h + 7) / 8) * ((w + 15) / 16);
  return nt < 1024 ? nt : 1024;
}
int launch_wgrad_c1_x3(const void* x, const void* dy, float* part, int n, int d, int h, int w, hipStream_t s) {
  Wc1X3Args a;
  a.x = (const float*)x;
  a.dy = (const float*)dy;
  a.part = part;
  a.n = n; a.d = d; a.h = h; a.w = w;
  a.tiles_z = (d + 3) / 4; a.tiles_y = (h + 7) / 8; a.tiles_x = (w + 15) / 16;
  a.ntiles = n * a.tiles_z * a.tiles_y * a.tiles_x;
  MEDNET_REQUIRE((double)d * h * w * 32 * 4.0 < 4294960000.0, MEDNET_E_UNSUPPORTED, "wgrad_c1_x3: one sample must stay below 4 GB");
  a.bytes_x = (unsigned)((size_t)d * h * w * 4);
  a.bytes_dy = (unsigned)((size_t)d * h * w * 32 * 4);
  constexpr size_t lds = 4352 + (size_t)2 * 512 * 64;  // halo of x + the two planes of dy (> the 16 KB of the final reduction)
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wgrad_c1_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(MEDNET_E_HIP, "wgrad_c1_x3: cannot raise dynamic LDS to %zu", lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_c1_x3_kernel, dim3(wgrad_c1_x3_blocks(n, d, h, w)), dim3(256), lds, s, a);
  return check_launch("wgrad_c1_x3");
}

int launch_wgrad_x3(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cin, int cout, void* ws,
                    size_t ws_bytes, hipStream_t s, int workgroups) {
  constexpr size_t lds = ((size_t)4 * 4 * 16 + 6 * 6 * 18) * 64 * 2;
  static_assert(lds <= 160 * 1024, "one workgroup per CU");
  MEDNET_REQUIRE(cin % 16 == 0 && cout % 16 == 0, MEDNET_E_UNSUPPORTED, "wgrad_x3: channels %d -> %d", cin, cout);
  WgX3Args a;
  a.A = (const float*)dy;  // A = dy (Cout rows), B = x (Cin cols)
  a.B = (const float*)x;
  a.part = (float*)ws;
  a.n = n; a.d = d; a.h = h; a.w = w; a.ka = cout; a.kb = cin;
  wgx3_plan(n, d, h, w, cout, cin, workgroups, a);
  a.bytesA = (unsigned)((size_t)d * h * w * cout * 4);
  a.bytesB = (unsigned)((size_t)d * h * w * cin * 4);
  const size_t need = (size_t)a.nab * a.nbb * a.splits * 27 * 1024 * sizeof(float);
  MEDNET_REQUIRE(ws_bytes >= need, MEDNET_E_WORKSPACE, "wgrad_x3: workspace %zu < %zu", ws_bytes, need);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wgrad_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(MEDNET_E_HIP, "wgrad_x3: cannot raise dynamic LDS to %zu", lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_x3_kernel, dim3(a.nab * a.nbb * a.splits), dim3(512), lds, s, a);
  int rc = check_launch("wgrad_x3");
  if (rc) return rc;
  const size_t total = (size_t)a.nab * a.nbb * 1024 * 27;
  if (a.splits < 4) hipLaunchKernelGGL(wgrad_x3_reduce_kernel_few, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.part, dw, cout, cin, a.nbb, a.splits);
  else hipLaunchKernelGGL(wgrad_x3_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, a.part, dw, cout, cin, a.nbb, a.splits);
  return check_launch("wgrad_x3_reduce");
}

// ================================================================================================== ConvTranspose3d weight gradient
//   dW[ci][co][tap] = sum_{n,j} x[n, j + delta][ci] * dy[n, 2 j + p][co]     per dim: tap k=1 -> (p=0, delta=0); k=0 -> (p=1, delta=1);
//                                                                              k=2 -> (p=1, delta=0)
// The contraction runs over the LOW-resolution voxels j.  A workgroup owns a 2x2x16 brick of them: it needs exactly the 4x4x32
// block of dy that starts at 2*j0 -- NO halo on the big tensor -- which goes to LDS sorted by output parity class (8 planes of
// 2x2x16 rows), and a 3x3x17 halo of x.  Every staged dy row feeds only 27/8 taps (a stride-1 weight gradient's rows feed 27),
// so the kernel pairs one 32-channel block of dy with CA = 2 blocks (64 channels) of x: 54 (tap, x-block) accumulators over 8
// waves (7 each), both operands through the transposing LDS read, high / low planes as everywhere in this file.  Still bound
// by staging (about twice the MFMA time), but 5-6x faster than the exact-fp32 kernel that does the same job with a halo on dy.
struct CtWgX3Args {
  const float* x;   // N x (d,h,w) x Cin
  const float* dy;  // N x (2d,2h,2w) x Cout
  float* part;      // [workgroup][27][CA * 32][32]
  int n, d, h, w, cin, cout;
  int tiles_z, tiles_y, tiles_x, tps, ntiles, nab, nbb, splits;
  unsigned bytes_x, bytes_dy;  // per sample
};

template <int CA>
__global__ __launch_bounds__(512) void convt_wgrad_x3_kernel(CtWgX3Args a) {
  constexpr int TZ = 2, TY = 2, TX = 16, HZ = 3, HY = 3, HX = 17;
  constexpr int NJ = TZ * TY * TX, NX = HZ * HY * HX, NDY = 8 * NJ;  // 64 low-res voxels, 153 halo voxels of x, 512 voxels of dy
  constexpr int NTHR = 512, ITEMS = 27 * CA, IPW = (ITEMS + 7) / 8;
  constexpr int X_PIECES = NX * 4 * CA, DY_PIECES = NDY * 4;
  constexpr int X_ROUNDS = (X_PIECES + NTHR - 1) / NTHR, DY_ROUNDS = DY_PIECES / NTHR;
  static_assert(DY_PIECES % NTHR == 0, "whole dy rounds");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // bytes: x_hi [CA][NX][64] | x_lo | dy_hi [8 planes][NJ][64] | dy_lo
  constexpr int X_LO = CA * NX * 64, DY_HI = 2 * CA * NX * 64, DY_LO = DY_HI + NDY * 64;

  const int pair = blockIdx.x / a.splits, split = blockIdx.x % a.splits;
  const int ab = pair / a.nbb, bb = pair % a.nbb;  // ab: block of CA * 32 input channels, bb: block of 32 output channels
  const int tid = threadIdx.x, lane = tid & 63;
  const int tw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, hk = lane >> 5;
  const int coloff = (16 * (g & 1) + 4 * p) * 2;

  u32x4 regX[X_ROUNDS][2], regD[DY_ROUNDS][2];
  struct Next {
    int jz0, jy0, jx0, valid;
    __amdgpu_buffer_rsrc_t rX, rD;
  };
  auto plan = [&](int tile) {
    Next nx;
    nx.valid = tile < a.ntiles;
    int tt = nx.valid ? tile : 0;
    const int n = tt / a.tps;
    tt -= n * a.tps;
    nx.jx0 = (tt % a.tiles_x) * TX;
    tt /= a.tiles_x;
    nx.jy0 = (tt % a.tiles_y) * TY;
    nx.jz0 = (tt / a.tiles_y) * TZ;
    nx.rX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)n * a.d * a.h * a.w * a.cin), 0, a.bytes_x, 0x00020000);
    nx.rD = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + (size_t)n * a.d * a.h * a.w * 8 * a.cout), 0, a.bytes_dy, 0x00020000);
    return nx;
  };
  auto fetch_round = [&](int j, const Next& nx) {
    if (j < X_ROUNDS) {  // piece c -> (x-block ca, halo voxel v, 8-channel part)
      const int c = j * NTHR + tid;
      const int part = c & 3, v = (c >> 2) % NX, ca = (c >> 2) / NX;
      const int gz = nx.jz0 + v / (HX * HY), gy = nx.jy0 + (v / HX) % HY, gx = nx.jx0 + v % HX;
      const int ch = (ab * CA + ca) * 32 + part * 8;
      const bool ok = (c < X_PIECES) & (gz < a.d) & (gy < a.h) & (gx < a.w) & (ch < a.cin) & (nx.valid != 0);
      const unsigned off = ok ? ((unsigned)((gz * a.h + gy) * a.w + gx) * (unsigned)a.cin + ch) * 4u : X3_OOB;
      regX[j][0] = __builtin_amdgcn_raw_buffer_load_b128(nx.rX, off, 0, 0);
      regX[j][1] = __builtin_amdgcn_raw_buffer_load_b128(nx.rX, off + 16u, 0, 0);
    } else if (j < X_ROUNDS + DY_ROUNDS) {  // piece c -> (hi-res voxel of the 4x4x32 block in x-fastest order, part)
      const int it = j - X_ROUNDS;
      const int c = it * NTHR + tid;
      const int part = c & 3, v = c >> 2;
      const int oz = 2 * nx.jz0 + v / (4 * 32), oy = 2 * nx.jy0 + (v / 32) % 4, ox = 2 * nx.jx0 + v % 32;
      const int ch = bb * 32 + part * 8;
      const bool ok = (oz < 2 * a.d) & (oy < 2 * a.h) & (ox < 2 * a.w) & (ch < a.cout) & (nx.valid != 0);
      const unsigned off = ok ? ((unsigned)((oz * 2 * a.h + oy) * 2 * a.w + ox) * (unsigned)a.cout + ch) * 4u : X3_OOB;
      regD[it][0] = __builtin_amdgcn_raw_buffer_load_b128(nx.rD, off, 0, 0);
      regD[it][1] = __builtin_amdgcn_raw_buffer_load_b128(nx.rD, off + 16u, 0, 0);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < X_ROUNDS; ++it) {
      const int c = it * NTHR + tid;
      if (c < X_PIECES) {
        const HiLo s = split8(regX[it][0], regX[it][1]);
        *reinterpret_cast<bf16x8*>(smem + c * 16) = s.hi;  // ((ca * NX + v) * 4 + part) * 16 = c * 16
        *reinterpret_cast<bf16x8*>(smem + X_LO + c * 16) = s.lo;
      }
    }
#pragma unroll
    for (int it = 0; it < DY_ROUNDS; ++it) {
      const int c = it * NTHR + tid;
      const int part = c & 3, v = c >> 2;
      const int lz = v / (4 * 32), ly = (v / 32) % 4, lx = v % 32;
      const int plane = (lz & 1) * 4 + (ly & 1) * 2 + (lx & 1);
      const int row = (plane * TZ + (lz >> 1)) * TY * TX + (ly >> 1) * TX + (lx >> 1);
      const HiLo s = split8(regD[it][0], regD[it][1]);
      *reinterpret_cast<bf16x8*>(smem + DY_HI + row * 64 + part * 16) = s.hi;
      *reinterpret_cast<bf16x8*>(smem + DY_LO + row * 64 + part * 16) = s.lo;
    }
  };

  // this wave's items: i = tw + 8 * s  ->  (tap = i / CA, x-block = i % CA); LDS offsets of their operand rows for k-step 0
  f32x16 acc[IPW];
  int aoff[IPW], boff[IPW];
#pragma unroll
  for (int s2 = 0; s2 < IPW; ++s2) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[s2][j] = 0.f;
    const int i = tw + 8 * s2 < ITEMS ? tw + 8 * s2 : ITEMS - 1;
    const int tap = i / CA, ca = i % CA;
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    const int dz = kz == 0, dyy = ky == 0, dx = kx == 0;                      // delta per dim
    const int plane = (kz != 1) * 4 + (ky != 1) * 2 + (kx != 1);                // output parity class
    aoff[s2] = ((ca * NX) + (dz * HY + dyy) * HX + dx) * 64;
    boff[s2] = plane * NJ * 64;
  }
  const char* Ab = smem + coloff;
  const char* Bb = smem + DY_HI + coloff;

  int tile = split;
  if (tile < a.ntiles) {
    const Next first = plan(tile);
#pragma unroll
    for (int j = 0; j < X_ROUNDS + DY_ROUNDS; ++j) fetch_round(j, first);
  }
  for (; tile < a.ntiles; tile += a.splits) {
    __syncthreads();  // the previous brick is consumed
    commit();
    __syncthreads();
    const Next nx = plan(tile + a.splits);
#pragma unroll
    for (int j = 0; j < X_ROUNDS + DY_ROUNDS; ++j) fetch_round(j, nx);  // (in flight during the brick's k-steps)
#pragma unroll
    for (int ks = 0; ks < TZ * TY; ++ks) {  // k-step = one x-row of 16 low-res voxels
      const int kzz = ks / TY, kyy = ks % TY;
      const int arow = ((kzz * HY + kyy) * HX + 8 * hk + q) * 64, brow = ((kzz * TY + kyy) * TX + 8 * hk + q) * 64;
#pragma unroll
      for (int s2 = 0; s2 < IPW; ++s2) {
        if (tw + 8 * s2 < ITEMS) {  // (wave-uniform: the last slot of some waves is empty)
          const bf16x8 ah = x3_tr_operand(Ab + aoff[s2] + arow, 4 * 64), al = x3_tr_operand(Ab + X_LO + aoff[s2] + arow, 4 * 64);
          const bf16x8 bh = x3_tr_operand(Bb + boff[s2] + brow, 4 * 64), bl = x3_tr_operand(Bb + (DY_LO - DY_HI) + boff[s2] + brow, 4 * 64);
          acc[s2] = X3_MFMA(al, bh, acc[s2]);
          acc[s2] = X3_MFMA(ah, bl, acc[s2]);
          acc[s2] = X3_MFMA(ah, bh, acc[s2]);
        }
      }
    }
  }
  float* out = a.part + (size_t)blockIdx.x * 27 * CA * 1024;
  const int col = lane & 31;
#pragma unroll
  for (int s2 = 0; s2 < IPW; ++s2) {
    const int i = tw + 8 * s2;
    if (i < ITEMS) {
      const int tap = i / CA, ca = i % CA;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        out[((size_t)tap * CA * 32 + ca * 32 + (j & 3) + 8 * (j >> 2) + 4 * hk) * 32 + col] = acc[s2][j];
    }
  }
}

// dw[(ci * Cout + co) * 27 + tap] = sum over the splits of part[(pair * splits + split)][tap][ci % (CA*32)][co % 32]
__global__ __launch_bounds__(256) void convt_wgrad_x3_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int cin,
                                                                    int cout, int nbb, int splits, int ca32) {
  const size_t slab = (size_t)27 * ca32 * 32;
  const size_t total = (size_t)((cin + ca32 - 1) / ca32) * nbb * slab;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int b32 = (int)(e % 32), arow = (int)((e / 32) % ca32), tap = (int)((e / (32 * (size_t)ca32)) % 27);
  const int pair = (int)(e / slab);
  const int ab = pair / nbb, bb = pair % nbb;
  const float* src = part + ((size_t)pair * splits) * slab + ((size_t)tap * ca32 + arow) * 32 + b32;
  float s0 = 0.f, s1 = 0.f;
  int k = 0;
  for (; k + 2 <= splits; k += 2) {
    s0 += src[(size_t)k * slab];
    s1 += src[(size_t)(k + 1) * slab];
  }
  if (k < splits) s0 += src[(size_t)k * slab];
  const int ci = ab * ca32 + arow, co = bb * 32 + b32;
  if (ci < cin && co < cout) dw[((size_t)ci * cout + co) * 27 + tap] = s0 + s1;
}

static int ctwg_ca(int cin) { return cin % 64 == 0 ? 2 : 1; }
static void ctwg_plan(int n, int d, int h, int w, int cin, int cout, int workgroups, CtWgX3Args& a) {
  const int ca32 = ctwg_ca(cin) * 32;
  a.tiles_z = (d + 1) / 2;
  a.tiles_y = (h + 1) / 2;
  a.tiles_x = (w + 15) / 16;
  a.tps = a.tiles_z * a.tiles_y * a.tiles_x;
  a.ntiles = n * a.tps;
  a.nab = (cin + ca32 - 1) / ca32;
  a.nbb = (cout + 31) / 32;
  const int pairs = a.nab * a.nbb;
  int splits = (x3_wgrad_target(workgroups) + pairs - 1) / pairs;
  if (splits > a.ntiles) splits = a.ntiles;
  if (splits < 1) splits = 1;
  a.splits = splits;
}
size_t convt_wgrad_x3_ws_bytes(int n, int d, int h, int w, int cin, int cout, int workgroups) {
  CtWgX3Args a;
  ctwg_plan(n, d, h, w, cin, cout, workgroups, a);
  return (size_t)a.nab * a.nbb * a.splits * 27 * ctwg_ca(cin) * 1024 * sizeof(float);
}
int launch_convt_wgrad_x3(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cin, int cout, void* ws,
                          size_t ws_bytes, hipStream_t s, int workgroups) {
  MEDNET_REQUIRE(cin % 16 == 0 && cout % 16 == 0, MEDNET_E_UNSUPPORTED, "convt_wgrad_x3: channels %d -> %d", cin, cout);
  CtWgX3Args a;
  a.x = (const float*)x; a.dy = (const float*)dy; a.part = (float*)ws;
  a.n = n; a.d = d; a.h = h; a.w = w; a.cin = cin; a.cout = cout;
  ctwg_plan(n, d, h, w, cin, cout, workgroups, a);
  a.bytes_x = (unsigned)((size_t)d * h * w * cin * 4);
  a.bytes_dy = (unsigned)((size_t)d * h * w * 8 * cout * 4);
  const int ca = ctwg_ca(cin);
  const size_t need = (size_t)a.nab * a.nbb * a.splits * 27 * ca * 1024 * sizeof(float);
  MEDNET_REQUIRE(ws_bytes >= need, MEDNET_E_WORKSPACE, "convt_wgrad_x3: workspace %zu < %zu", ws_bytes, need);
  const size_t lds = ((size_t)2 * ca * 153 + 2 * 512) * 64;
  static bool attr_set[3] = {false, false, false};
  if (!attr_set[ca]) {
    const hipError_t e = ca == 2 ? hipFuncSetAttribute((const void*)convt_wgrad_x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                 : hipFuncSetAttribute((const void*)convt_wgrad_x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(MEDNET_E_HIP, "convt_wgrad_x3: cannot raise dynamic LDS to %zu", lds);
    attr_set[ca] = true;
  }
  const dim3 grid(a.nab * a.nbb * a.splits);
  if (ca == 2) hipLaunchKernelGGL(convt_wgrad_x3_kernel<2>, grid, dim3(512), lds, s, a);
  else hipLaunchKernelGGL(convt_wgrad_x3_kernel<1>, grid, dim3(512), lds, s, a);
  int rc = check_launch("convt_wgrad_x3");
  if (rc) return rc;
  const size_t total = (size_t)a.nab * a.nbb * 27 * ca * 1024;
  hipLaunchKernelGGL(convt_wgrad_x3_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.part, dw, cin, cout, a.nbb,
                     a.splits, ca * 32);
  return check_launch("convt_wgrad_x3_reduce");
}

}  // namespace mednet
