// 16-bit (bf16 / fp16) MFMA (matrix-core) kernels for the 3x3x3 convolutions of the U-Net: forward / data-gradient (one kernel, two
// weight images) and weight-gradient.  gfx950 only: v_mfma_f32_32x32x16_bf16, ds_read_b128, ds_read_b64_tr_b16.
//
// ---- forward / data gradient: implicit GEMM, im2col-free ---------------------------------------------------------
//   D[co][voxel] += sum_tap sum_ci W[tap][co][ci] * X[voxel + tap][ci]       (M = 32 output channels, N = 32 voxels)
// A workgroup (4 waves) owns a TZ x TY x 16 brick of output voxels and one block of 32 output channels.  The K loop
// walks the input channels 16 at a time (one MFMA k-step).  Per chunk the brick's input halo (6x10x18 voxels x 16 ch)
// and the 27x32x16 weight slice are staged in LDS ONCE and reused by all 27 taps: every MFMA B operand is a single
// conflict-free ds_read_b128 (8 channels of one voxel), every A operand one ds_read_b128 shared by 4 N-tiles.
// Staging is software-pipelined through registers (global loads of chunk k+1 are in flight while chunk k is on the
// matrix cores) and two workgroups share a CU (62 KB LDS each) so one brick's LDS fill overlaps the other's MFMAs.
// LDS images are split by k-half:  in[h][voxel][8 ch],  w[tap][h][co][8 ch]  so that the 16 lanes a ds_read_b128
// services together hit 16 distinct 16-byte slots; N-tile rows are rotated by the halo row pitch for the same reason.
//
// ---- weight gradient ----------------------------------------------------------------------------------------------
//   R[tap][a][b] = sum_voxel A[voxel][a] * B[map(voxel, tap)][b]     (conv: A = dy, B = x;  convT: A = x, B = dy)
// The contraction index is the voxel, which is the slow axis of both channels-last operands, so both MFMA operands
// are read with the hardware transposing LDS read (ds_read_b64_tr_b16): no transposed copy is ever materialised.
// A wave keeps 7 of the 27 taps (7 x 16 accumulator registers) and walks its workgroup's bricks; per-workgroup
// partials are summed in a fixed order by a second kernel (deterministic, no atomics).
#include <type_traits>
#include "conv.h"

#include "elt16.inc"

namespace mednet {

// ================================================================================================== geometry
// Brick KINDS of the forward / data-gradient kernel (the template parameter of conv_mfma_kernel and launch_fwd):
template <int KIND>
struct FwdTile;
template <>
struct FwdTile<1> {  // conv forward / data gradient
  static constexpr int TZ = 4, TY = 8, TX = 16, STRIDE = 1;
};
template <>
struct FwdTile<2> {  // ConvTranspose3d data gradient: in = 2*out - 1 + tap
  static constexpr int TZ = 2, TY = 4, TX = 16, STRIDE = 2;
};
template <>
struct FwdTile<3> {  // conv forward / data gradient on NARROW volumes (round 5): half as wide.  The deep levels of BASELINE config 5
  // (f_maps [64 .. 1024] on 160 x 160 x 96 patches: ... 20 x 20 x 12, 10 x 10 x 6 voxels) fill 16-wide bricks badly in x -- 6 of 16
  // columns at the deepest level -- and every padded column costs its MFMAs; with 8-wide bricks a 10 x 10 x 6 volume is 39 % real
  // voxels instead of 10 %.  A wave then has two N-tiles (4 x-rows of 8 voxels each) per weight fragment instead of four.
  static constexpr int TZ = 4, TY = 8, TX = 8, STRIDE = 1;
};
// 8-wide bricks for volumes at most 8 voxels wide.  (Measured, profiles/r05_ab.md: for wider volumes whose last 16-wide brick is at
// most half full -- 24 = 16 + 8 columns, config 5's level 2 -- 8-wide bricks do 25 % fewer MFMAs and are 10 % SLOWER: with two N-tiles
// per weight fragment instead of four the kernel is LDS-bound; option conv_narrow=2 selects them there too.)
static bool narrow_bricks(int w) {
  const int mode = tuning_option("conv_narrow", 1);
  return mode != 0 && (w <= 8 || (mode == 2 && (w + 7) / 8 < 2 * ((w + 15) / 16)));
}

// Order in which the forward / data-gradient kernels of this file visit the 27 taps of a K chunk: position s = (kz, kx, ky) with ky
// FASTEST, i.e. tap index tap_at(s) = kz * 9 + ky * 3 + kx.  Round 6: conv32_mfma_kernel serves the three ky taps of a (kz, kx)
// from the same six operand fragments (its N-tiles are row pairs (t, t + 4), so a shift in y is a register renaming), which
// needs them adjacent; every kernel accumulates in this one order so that their outputs stay bit-identical to each other.
__host__ __device__ constexpr int tap_at(int s) { return (s / 9) * 9 + (s % 3) * 3 + (s / 3) % 3; }

// x / d for launch constants d: the host passes ceil(2^32 / d); exact while x * d < 2^32 (checked by the launcher).
// d == 1 has no 32-bit reciprocal and is passed through.
__device__ __forceinline__ int fastdiv(int x, int d, unsigned rcp) { return d == 1 ? x : (int)__umulhi((unsigned)x, rcp); }

// ================================================================================================== forward kernel
struct FwdArgs {
  const elt* x;
  const elt* wpk;  // [cb][kc][27][2][32][8]
  elt* y;
  int n, od, oh, ow;  // output grid
  int id, ih, iw;     // input grid
  int cin, cout;
  int tiles_z, tiles_y, tiles_x, ntiles;  // per sample * n
  int nkc, ncb;
  int nitems;  // ((ntiles + 7) / 8) * 8 * ncb work items (the last brick group may hold padding items)
  unsigned rcp_tiles_x, rcp_tiles_y, rcp_tiles_z, rcp_ncb;  // ceil(2^32 / d): x / d == umulhi(x, rcp) for x * d < 2^32
  unsigned bytes_x;  // size of ONE SAMPLE of x for the buffer resource (< 4 GB)
  unsigned bytes_y;  // same for y
  const elt* add;    // nullable, shape of y: summed into the output in the epilogue (fp32 add, one rounding) -- the residual
                      // branch's gradient joining the data gradient of ExtResNetBlock's second conv (components.py:170-178)
  int act;            // MEDNET_ACT_*: applied to the fp32 accumulators before the elt store (conv -> ReLU/LeakyReLU/ELU of the
                      // 'gcr' orders, components.py:36-40); the fused statistics are then those of the ACTIVATED output
  float* gn_partial;  // nullable: [n][stats_rows][cout][2] = {sum y, sum y^2} of the STORED (rounded) outputs, see stats_accum
  int stats_accum;    // 1: a wave keeps its sums over all its items of a sample and writes ONE row per sample (row =
                      //    4 * workgroup-group + wave); 0: one row per wave and brick (row = 4 * brick + wave)
  int stats_rows;     // rows per sample of gn_partial
  // GNB instantiation (data gradient whose output dz is the input gradient of a GroupNorm + activation, components.py:57,
  // 36-40): the first pass of that GroupNorm's backward -- du = dz * act'(ca * y + cb), sums of du and du * y per channel --
  // is taken in the epilogue from the STORED (rounded) dz rows, so no stand-alone pass re-reads dz and y.  gn_partial then
  // holds [n][stats_rows][cout][2] = {sum du, sum du * y} per CHANNEL (the affine gradients need single channels).
  const elt* gnb_y;      // conv output y of the layer in front (shape of this kernel's output)
  const float* gnb_coef;  // [n][cout][2] = {ca, cb}: its GroupNorm's forward affine, pre-activation = ca * y + cb
  int gnb_act;            // MEDNET_ACT_*
  const elt* gnb_z;       // nullable: the layer in front is the RESIDUAL layer of an ExtResNetBlock (GroupNorm-3, components.py:
                          // 170-178): this kernel's output is the block's output gradient, the activation derivative comes from
                          // the block OUTPUT z (shape of this kernel's output) and gnb_coef is not used
  int xcd_chunk;          // conv32_mfma_kernel: bricks per XCD when the brick count divides by 8 (each XCD then works through a
                          // CONTIGUOUS part of the volume, so neighbouring bricks' halos meet in its L2), else 0
  int zslab;              // conv32_mfma_kernel: z-layers of bricks per XCD (> 0: the x-z-y walk of origin(); implies xcd_chunk)
  unsigned rcp_zslab;
  // Split weights (round 6, MEDNET_ALGO_SPLITW_BIT): the layer's LOW image -- elt(w - elt(w)), what 16-bit rounding takes from a
  // weight -- lies lo_delta bytes behind the high one and is multiplied too, into the same fp32 accumulators (fp16 MFMAs honour
  // subnormal inputs: tools/probes/f16_denorm_probe.py).  conv_mfma_kernel runs the low image as nkc_in more K chunks (nkc = 2
  // nkc_in loop chunks over the same nkc_in input chunks); conv2b_mfma_kernel<V | C32_SPLIT> takes (high, low) of ONE channel
  // block as its two "blocks".  0 = off.
  unsigned lo_delta;
  int nkc_in;  // input K chunks (cin / 16)
};

template <int KIND, bool GNB = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(FwdArgs a) {
  using G = FwdTile<KIND>;
  constexpr int TZ = G::TZ, TY = G::TY, TX = G::TX, STRIDE = G::STRIDE;
  constexpr int RPT = 32 / TX;  // x-rows of an N-tile (32 voxels): 2 rows of 16 or 4 rows of 8
  constexpr int HZ = STRIDE * (TZ - 1) + 3, HY = STRIDE * (TY - 1) + 3, HX = STRIDE * (TX - 1) + 3;
  constexpr int NV = HZ * HY * HX;
  constexpr int NTW = TZ * TY * TX / 32 / 4;  // N-tiles per wave
  constexpr int IN_ROUNDS = (2 * NV + 255) / 256;
  constexpr int W_CHUNKS = 27 * 2 * 32;  // 16-byte pieces of one weight slice
  constexpr int W_ROUNDS = (W_CHUNKS + 255) / 256;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  typedef __attribute__((ext_vector_type(2))) elt eltx2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* in_lds = reinterpret_cast<u32x4*>(smem);                  // [2][NV] 16-byte pieces (8 elt channels)
  u32x4* w_lds = reinterpret_cast<u32x4*>(smem) + 2 * NV;          // [27][2][32]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  // ---- work items: (brick, channel block), numbered so that all channel blocks of a brick run on the same XCD (ids 8
  //      apart share an L2).  A workgroup takes items blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8, so it
  //      stays on its XCD).  The launcher gives either every item its own workgroup, or 2 workgroups per CU that stream
  //      through the items with the NEXT item's first K chunk already in flight while the last chunk of the current one
  //      is on the matrix cores (no exposed load latency per brick).
  // (divisions by launch constants go through host-made reciprocals: a runtime scalar division costs ~30 dependent
  //  instructions)
  const int tiles_per_sample = a.tiles_x * a.tiles_y * a.tiles_z;
  auto decode = [&](int bid, int& tile, int& cb) {
    const int local = bid >> 3;
    const int lq = fastdiv(local, a.ncb, a.rcp_ncb);
    tile = lq * 8 + (bid & 7);
    cb = local - lq * a.ncb;
  };
  auto origin = [&](int tile, int& n, int& tz0, int& ty0, int& tx0) {
    int tt = tile;
    int qd = fastdiv(tt, a.tiles_x, a.rcp_tiles_x);
    tx0 = (tt - qd * a.tiles_x) * TX;
    tt = qd;
    qd = fastdiv(tt, a.tiles_y, a.rcp_tiles_y);
    ty0 = (tt - qd * a.tiles_y) * TY;
    tt = qd;
    qd = fastdiv(tt, a.tiles_z, a.rcp_tiles_z);
    tz0 = (tt - qd * a.tiles_z) * TZ;
    n = qd;
  };
  int cur_bid = blockIdx.x, ctile, ccb;
  decode(cur_bid, ctile, ccb);
  if (ctile >= a.ntiles) return;  // (padding items of the last brick group; workgroup-uniform)

  // ---- staging plan: this thread's halo pieces are the same for every brick (hpos); per brick they become 32-bit BYTE
  //      offsets for buffer loads whose resource descriptor (one SAMPLE, so only a sample has to stay below 4 GB) sits in
  //      SGPRs.  Pieces outside the volume get an offset beyond num_records: the hardware range check returns zeros for
  //      them, so zero padding costs neither a branch nor a select.
  constexpr unsigned OOB = 0xFFFFFF00u;
  int hpos[IN_ROUNDS];  // hz << 16 | hy << 8 | hx; slots past the halo get coordinates that fail every range check
#pragma unroll
  for (int it = 0; it < IN_ROUNDS; ++it) {
    const int v = (it * 256 + tid) >> 1;
    hpos[it] = v < NV ? ((v / (HX * HY)) << 16) | (((v / HX) % HY) << 8) | (v % HX) : 0x7FFF0000;
  }
  const int hh = tid & 1;
  unsigned goff[IN_ROUNDS];
  int pn = 0;  // sample of the brick the plan (and the loads in flight) belong to
  auto plan = [&](int tile) {
    int tz0, ty0, tx0;
    origin(tile, pn, tz0, ty0, tx0);
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) {
      const int gz = STRIDE * tz0 - 1 + (hpos[it] >> 16), gy = STRIDE * ty0 - 1 + ((hpos[it] >> 8) & 255),
                gx = STRIDE * tx0 - 1 + (hpos[it] & 255);
      // branch-free: one unsigned compare per axis covers both bounds ('&', not '&&': no branch ladder)
      const bool in_vol = ((unsigned)gz < (unsigned)a.id) & ((unsigned)gy < (unsigned)a.ih) & ((unsigned)gx < (unsigned)a.iw);
      const unsigned off = ((unsigned)((gz * a.ih + gy) * a.iw + gx) * (unsigned)a.cin + hh * 8) * 2u;
      goff[it] = in_vol ? off : OOB;
    }
  };

  u32x4 in_reg[IN_ROUNDS], w_reg[W_ROUNDS];
  // loop chunk kc: input chunk kc mod nkc_in; weight slice of the high image, or (kc >= nkc_in: split weights) of the low one
  auto w_slice = [&](int cb, int kc) {
    const bool low = kc >= a.nkc_in;
    return reinterpret_cast<const char*>(a.wpk) + (low ? (size_t)a.lo_delta : 0) + ((size_t)cb * a.nkc_in + (low ? kc - a.nkc_in : kc)) * (W_CHUNKS * 16);
  };
  auto in_chunk = [&](int kc) { return kc >= a.nkc_in ? kc - a.nkc_in : kc; };
  auto prefetch = [&](int cb, int kc) {  // 15 loads issued back to back, nothing waits on them until commit()
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)pn * a.id * a.ih * a.iw * a.cin), 0, a.bytes_x, 0x00020000);
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) in_reg[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, goff[it], in_chunk(kc) * 32, 0);
    // the weight slice of (cb, kc) is one linear 27 KB run: thread t takes pieces t, t+256, ...; the last round is cut
    // off by the resource's size (the round is part of voffset: the hardware range check does not see soffset)
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_slice(cb, kc), 0, W_CHUNKS * 16, 0x00020000);
#pragma unroll
    for (int it = 0; it < W_ROUNDS; ++it) w_reg[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, tid * 16 + it * 4096, 0, 0);
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) {
      const int p = it * 256 + tid;
      if (p < 2 * NV) in_lds[(p & 1) * NV + (p >> 1)] = in_reg[it];
    }
#pragma unroll
    for (int it = 0; it < W_ROUNDS; ++it) {
      const int c = it * 256 + tid;
      if (c < W_CHUNKS) w_lds[c] = w_reg[it];
    }
  };

  // ---- this lane's voxel in each of the wave's N-tiles (2 rows x 16 voxels), row-rotated for conflict-free reads
  int lbase[NTW];  // LDS piece index of tap (0,0,0) for this lane
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int g = wv * NTW + t;
    const int lz = g / (TY / RPT), ly = (g % (TY / RPT)) * RPT + r / TX;
    const int lx = TX == 8 ? (r & 7) : (STRIDE == 1 ? (((r & 15) - (r >> 4) * HX) & 15) : (r & 15));
    lbase[t] = ((STRIDE * lz) * HY + STRIDE * ly) * HX + STRIDE * lx + h * NV;
  }
  const eltx2 ones = {(elt)1.0f, (elt)1.0f};
  const size_t ovol = (size_t)a.od * a.oh * a.ow;

  // fused GroupNorm statistics (see the epilogue): per channel PAIR, kept across the items of one sample when stats_accum
  float gs[4], gq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) gs[k] = gq[k] = 0.f;
  int acc_n = 0;  // sample the running sums belong to
  auto flush_stats = [&](int nn, int row, int cbk) {  // reduce over the wave, lanes 0..3 write {sum, sumsq, 0, 0} per pair
    float rs[4], rq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rs[k] = gs[k];
      rq[k] = gq[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // the 16 lanes with the same piece (lane & 3): DPP / permlane, no LDS-queue shuffles
      rs[k] = lane_class_sum<4>(rs[k]);
      rq[k] = lane_class_sum<4>(rq[k]);
    }
    const int pjl = lane & 3;
    if (lane < 4 && cbk * 32 + pjl * 8 < a.cout) {
      float* dst = a.gn_partial + (((size_t)nn * a.stats_rows + row) * a.cout + cbk * 32 + pjl * 8) * 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 o = {rs[k], rq[k], 0.f, 0.f};
        *reinterpret_cast<f32x4*>(dst + k * 4) = o;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) gs[k] = gq[k] = 0.f;
  };
  // GNB: this lane's 8 channels (piece pj = lane & 3 of the channel block), {sum du, sum du * y}.  In the stride-1 form the 16
  // running sums are PARKED in LDS between epilogues ([k][thread]: conflict-free 4-byte accesses of the owning thread, no
  // barrier): held in registers across the tap loop they pushed the kernel over 256 registers (20 spilled).
  constexpr bool PARK = GNB && STRIDE == 1;
  auto park_of = [&]() {  // [16][256] floats behind the images (recomputed from an opaque thread id where it is used: as a
    int t = tid;           // loop-invariant address it would be kept in a register across the tap loop)
    asm volatile("" : "+v"(t));
    return reinterpret_cast<float*>(smem + (size_t)(2 * NV + W_CHUNKS) * 16) + t;
  };
  float bs[GNB && !PARK ? 8 : 1], bq[GNB && !PARK ? 8 : 1];
#pragma unroll
  for (int k = 0; k < (GNB && !PARK ? 8 : 1); ++k) bs[k] = bq[k] = 0.f;
  if constexpr (PARK) {
    float* park = park_of();
#pragma unroll
    for (int k = 0; k < 16; ++k) park[k * 256] = 0.f;
  }
  auto flush_bwd = [&](int nn, int row, int cbk) {  // sum over the 16 lanes that share a piece; lanes 0..3 write 64 bytes
    if constexpr (GNB) {
      float rs[8], rq[8];
      float* park = park_of();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if constexpr (PARK) {
          rs[k] = park[k * 256];
          rq[k] = park[(8 + k) * 256];
          park[k * 256] = 0.f;
          park[(8 + k) * 256] = 0.f;
        } else {
          rs[k] = bs[k];
          rq[k] = bq[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        rs[k] = lane_class_sum<4>(rs[k]);
        rq[k] = lane_class_sum<4>(rq[k]);
      }
      const int pjl = lane & 3;
      if (lane < 4 && cbk * 32 + pjl * 8 < a.cout) {
        float* dst = a.gn_partial + (((size_t)nn * a.stats_rows + row) * a.cout + cbk * 32 + pjl * 8) * 2;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          const f32x4 o = {rs[k], rq[k], rs[k + 1], rq[k + 1]};
          *reinterpret_cast<f32x4*>(dst + k * 2) = o;
        }
      }
      if constexpr (!PARK) {
#pragma unroll
        for (int k = 0; k < 8; ++k) bs[k] = bq[k] = 0.f;
      }
    }
  };
  auto flush_any = [&](int nn, int row, int cbk) {
    if constexpr (GNB) flush_bwd(nn, row, cbk);
    else flush_stats(nn, row, cbk);
  };
  // row of this wave in accumulate mode: workgroups that share (id >> 3) / ncb and the XCD form one row group; they differ
  // in the channel block (which is constant per workgroup when ncb divides 64: the launcher checks that)
  const int acc_row = ((((int)blockIdx.x >> 3) / a.ncb) * 8 + ((int)blockIdx.x & 7)) * 4 + wv;

  plan(ctile);
  prefetch(ccb, 0);
  while (true) {
    int n, tz0, ty0, tx0;
    origin(ctile, n, tz0, ty0, tx0);
    const int tis = ctile - n * tiles_per_sample;
    const int cb = ccb;
    // the item after this one
    const int nbid = cur_bid + (int)gridDim.x;
    int ntile = 0, ncb2 = 0;
    bool has_next = false;
    if (nbid < a.nitems) {
      decode(nbid, ntile, ncb2);
      has_next = ntile < a.ntiles;
    }

    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (int kc = 0; kc < a.nkc; ++kc) {
      __syncthreads();  // every wave is done reading the previous LDS image (MFMA operands or the epilogue's rows)
      commit();
      __syncthreads();
      // What flies while chunk kc is on the matrix cores: the next chunk of this item, or the first chunk of the next
      // item.  Its 15 loads are NOT issued in one burst (measured: a burst blocks the wave's issue for ~3000 cycles
      // behind the CU's one texture-address path, which both workgroups share); they are dealt out one per tap, so the
      // wave keeps feeding the matrix cores.  No branch inside the tap loop (it would end the scheduling region): when
      // there is nothing to fetch the offsets are pushed out of range and the loads return zeros without touching memory.
      bool do_pf = true;
      int pf_cb = cb, pf_kc = kc + 1;
      if (kc + 1 >= a.nkc) {
        pf_kc = 0;
        pf_cb = ncb2;
        do_pf = has_next;
        if (has_next) plan(ntile);
      }
      const unsigned kill = do_pf ? 0u : OOB;
      const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)pn * a.id * a.ih * a.iw * a.cin), 0, a.bytes_x, 0x00020000);
      const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_slice(pf_cb, pf_kc), 0, W_CHUNKS * 16, 0x00020000);
      const int pf_in = in_chunk(pf_kc) * 32;
      // Software-pipelined over the 27 taps: the fragments of tap t+1 (1 weight + NTW input ds_read_b128) are in flight
      // while tap t is on the matrix cores; sched_group_barrier pins that interleave (hipcc otherwise sinks each read to
      // just in front of its MFMA and exposes the LDS latency 108 times per chunk).
      constexpr int PD = 1;  // prefetch distance in taps (PD + 1 operand sets in registers); 2 measured no faster (+20 VGPRs)
      eltx8 wa[PD + 1], xb[PD + 1][NTW];
      auto tap_off = [&](int t1) { return ((t1 / 9) * HY + (t1 / 3) % 3) * HX + t1 % 3; };
#pragma unroll
      for (int p = 0; p < PD; ++p) {  // (loop positions are visited in the order tap_at(): see there)
        wa[p] = __builtin_bit_cast(eltx8, w_lds[(tap_at(p) * 2 + h) * 32 + r]);
#pragma unroll
        for (int t = 0; t < NTW; ++t) xb[p][t] = __builtin_bit_cast(eltx8, in_lds[lbase[t] + tap_off(tap_at(p))]);
      }
      static_assert(IN_ROUNDS + W_ROUNDS <= 27, "one staging load per tap");
      // The wave in its MFMA loop outranks its SIMD partner (the other workgroup's wave, which is then staging, in its
      // epilogue or in its own tap loop): measured 0.493 -> 0.472 ms on the 32 -> 32 @128^3 launches, 2 x 3 interleaved runs
      // (priority 3 the same); without it the older wave wins every arbitration whatever it is doing.
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        const int cur = tap % (PD + 1), nxt = (tap + PD) % (PD + 1);
        if (tap + PD < 27) {
          const int t1 = tap_at(tap + PD);
          const int toff = tap_off(t1);
          wa[nxt] = __builtin_bit_cast(eltx8, w_lds[(t1 * 2 + h) * 32 + r]);
#pragma unroll
          for (int t = 0; t < NTW; ++t) xb[nxt][t] = __builtin_bit_cast(eltx8, in_lds[lbase[t] + toff]);
        }
        if (tap < IN_ROUNDS)
          in_reg[tap] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, goff[tap] | kill, pf_in, 0);
        else if (tap < IN_ROUNDS + W_ROUNDS)
          w_reg[tap - IN_ROUNDS] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (unsigned)(tid * 16 + (tap - IN_ROUNDS) * 4096) | kill, 0, 0);
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = MEDNET_MFMA_32x32x16(wa[cur], xb[cur][t], acc[t], 0, 0, 0);
        if (tap + PD < 27) __builtin_amdgcn_sched_group_barrier(0x100, NTW + 1, 0);  // DS reads of tap+PD first ...
        if (tap < IN_ROUNDS + W_ROUNDS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // ... one staging load ...
        __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);                          // ... then the MFMAs of tap
      }
      __builtin_amdgcn_s_setprio(0);
    }

    // ---- epilogue through LDS.  D[row = co][col = voxel]: the accumulator layout gives every lane four 8-byte pieces of
    // its voxel's 64-byte channel row, so a direct store instruction would touch 32 rows with 8 bytes each.  Instead the
    // brick's output is assembled in LDS (8-byte chunks XOR-swizzled by voxel so both the ds_write_b64 and the
    // ds_read_b128 are conflict-free) and written out as whole rows: 4 lanes per voxel, 16 voxels = 1 KB contiguous per
    // wave instruction when Cout = 32.  The GroupNorm partial sums come from the same LDS image (8 channels per lane).
    elt* out_lds = reinterpret_cast<elt*>(smem);  // [TZ*TY*TX voxels][32 co], reuses the input image
    __syncthreads();                                // every wave is done with the MFMA reads of the last chunk
    // (an opaque copy of the thread id: without it the compiler hoists the epilogue's ~30 addresses, which are the same
    //  for every item, out of the item loop and pays for that with spills in the tap loop)
    int etid = tid;
    asm volatile("" : "+v"(etid));
    const int e_lane = etid & 63, e_wv = etid >> 6, e_r = e_lane & 31, e_h = e_lane >> 5;
#pragma unroll
    for (int t = 0; t < NTW; ++t) act_apply_v16(acc[t], a.act);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int g = e_wv * NTW + t;
      const int lz = g / (TY / RPT), ly = (g % (TY / RPT)) * RPT + e_r / TX;
      const int lx = TX == 8 ? (e_r & 7) : (STRIDE == 1 ? (((e_r & 15) - (e_r >> 4) * HX) & 15) : (e_r & 15));
      const int vl = (lz * TY + ly) * TX + lx;
      const int sw = (vl >> 1) & 7;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        eltx4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (elt)acc[t][q * 4 + j];  // co = 8q + 4h + j
        *reinterpret_cast<eltx4*>(out_lds + vl * 32 + ((2 * q + e_h) ^ sw) * 4) = o;
      }
    }
    __syncthreads();
    // GroupNorm partials are kept per channel PAIR (v_dot2c_f32_bf16: two exact elt products + fp32 add per
    // instruction, 8 instructions per 8-channel piece instead of 24): entry 2j of the partial row gets the sums of
    // channels 2j and 2j+1, entry 2j+1 is zero.  GroupNorm only ever adds the channels of a group, so this is exact
    // whenever the channels per group are even (the host asks for fused partials only then).
    if (a.gn_partial && a.stats_accum) {
      while (acc_n < n) {  // the sample changed: write the finished sample's row (zeros for samples this wave skipped)
        flush_any(acc_n, acc_row, cb);
        ++acc_n;
      }
    }
    const int pj = etid & 3;
    // Row stores: lane = (voxel et of 64, 16-byte piece pj of 4).  The 64 voxels of round `it` are 4 x-rows of 16, so
    // the per-round part of the address is a SCALAR (soffset of a buffer store whose resource is this sample) and the
    // lane part is computed once; lanes outside the volume / past Cout get an out-of-range voffset and the hardware
    // drops the store.
    constexpr int RPR = 64 / TX;  // x-rows per round of 64 voxels
    static_assert((TX == 16 || TX == 8) && TY % RPR == 0, "row decomposition of the epilogue");
    const int et = etid >> 2, ex = et & (TX - 1), ey = et / TX;
    const bool lane_ok = (tx0 + ex < a.ow) & (cb * 32 + pj * 8 < a.cout);  // (a 16-channel layer fills half a block)
    const unsigned vbase = (unsigned)((ey * a.ow + ex) * a.cout + pj * 8) * 2u;
    const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (size_t)n * ovol * a.cout), 0, a.bytes_y, 0x00020000);
    const auto rsrc_add = __builtin_amdgcn_make_buffer_rsrc((void*)((a.add ? a.add : a.y) + (size_t)n * ovol * a.cout), 0, a.bytes_y, 0x00020000);
    const int swl = (et >> 1) & 7;  // (vl >> 1) & 7 does not depend on the round: 64 voxels per round
    const elt* rd = out_lds + et * 32 + (pj ^ (swl >> 1)) * 8;
    constexpr int ROUNDS = (TZ * TY * TX * 4) / 256;
    // The GNB form walks the rounds in groups of at most 4: with all 8 rounds of the stride-1 brick in flight at once (8 output
    // rows + 8 y rows + 8 z rows = 96 registers) on top of the next item's staged chunk, the kernel spilled 49 registers (200
    // bytes of scratch per lane, in the epilogue AND around the tap loop).
    constexpr int RG = (GNB && ROUNDS > 2) ? 2 : ROUNDS;
    float ca[GNB ? 8 : 1], cbf[GNB ? 8 : 1];
    const auto rsrc_gy = __builtin_amdgcn_make_buffer_rsrc((void*)((GNB ? a.gnb_y : a.y) + (size_t)n * ovol * a.cout), 0, a.bytes_y, 0x00020000);
    const auto rsrc_gz = __builtin_amdgcn_make_buffer_rsrc((void*)((GNB && a.gnb_z ? a.gnb_z : a.y) + (size_t)n * ovol * a.cout), 0, a.bytes_y, 0x00020000);
    auto load_coef = [&]() {  // the GroupNorm affine of this lane's 8 channels (64 bytes, cache resident)
      if constexpr (GNB) {
        if (!a.gnb_z) {
          const auto rsrc_cf = __builtin_amdgcn_make_buffer_rsrc((void*)(a.gnb_coef + (size_t)n * a.cout * 2), 0, (unsigned)a.cout * 8u, 0x00020000);
          const unsigned coff = (unsigned)(cb * 32 + pj * 8) * 8u;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 c4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_cf, coff + q * 16, 0, 0));
            ca[2 * q] = c4[0];
            cbf[2 * q] = c4[1];
            ca[2 * q + 1] = c4[2];
            cbf[2 * q + 1] = c4[3];
          }
        }
      }
    };
    if constexpr (!PARK) load_coef();
    float ebs[PARK ? 8 : 1], ebq[PARK ? 8 : 1];  // (the parked sums, in registers for the epilogue only)
    float* park = park_of();
    if constexpr (PARK) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        ebs[k] = park[k * 256];
        ebq[k] = park[(8 + k) * 256];
      }
    }
#pragma unroll
    for (int it0 = 0; it0 < ROUNDS; it0 += RG) {
    if (it0 > 0) __builtin_amdgcn_sched_barrier(0);  // (keeps the next group's loads behind this group's stores)
    eltx8 rows[RG];  // all LDS reads of the group first (the accumulators' registers are free now), then the stores back to
#pragma unroll        // back: read -> wait -> store per round exposed the LDS latency eight times
    for (int it = 0; it < RG; ++it) rows[it] = *reinterpret_cast<const eltx8*>(rd + (it0 + it) * 64 * 32);
    // GNB: the rows of y (and of the block output z) at the positions this lane stores, through buffer resources (positions /
    // channels outside the tensor read zeros), all of the group in flight before the first use
    eltx8 yrow[GNB ? RG : 1], zrow[GNB ? RG : 1];
    if constexpr (PARK) load_coef();  // (per group: 16 registers that need not live through the whole epilogue)
    if constexpr (GNB) {
#pragma unroll
      for (int itl = 0; itl < RG; ++itl) {
        const int it = it0 + itl;
        const int oz = tz0 + (it * RPR) / TY, oyb = ty0 + (it * RPR) % TY;
        const bool ok = lane_ok & (oz < a.od) & (oyb + ey < a.oh);
        const unsigned soff = (unsigned)(((oz * a.oh + oyb) * a.ow + tx0) * a.cout + cb * 32) * 2u;
        yrow[itl] = __builtin_bit_cast(eltx8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_gy, ok ? vbase : OOB, soff, 0));
        if (a.gnb_z) zrow[itl] = __builtin_bit_cast(eltx8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_gz, ok ? vbase : OOB, soff, 0));
      }
    }
#pragma unroll
    for (int itl = 0; itl < RG; ++itl) {
      const int it = it0 + itl;
      eltx8 v = rows[itl];
      if (swl & 1) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);
      const int oz = tz0 + (it * RPR) / TY, oyb = ty0 + (it * RPR) % TY;
      const bool ok = lane_ok & (oz < a.od) & (oyb + ey < a.oh);
      const unsigned soff = (unsigned)(((oz * a.oh + oyb) * a.ow + tx0) * a.cout + cb * 32) * 2u;
      if (a.add) {  // (wave-uniform) y += add: 16-byte row pieces at the same offsets, out-of-range lanes read zeros
        const eltx8 ad = __builtin_bit_cast(eltx8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_add, ok ? vbase : OOB, soff, 0));
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (elt)((float)v[k] + (float)ad[k]);
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_y, ok ? vbase : OOB, soff, 2);  // aux 2 = nt: streamed output must not push the input rows (read again by the next K chunk) out of L2
      if constexpr (GNB) {
        if (ok) {  // du = dz * act'(pre-activation) from the STORED dz, exactly what the apply pass recomputes
          float g[8], u[8], yy[8];
          if (a.gnb_z) {  // (wave-uniform) residual layer: act' from the block output
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              yy[k] = (float)yrow[itl][k];
              g[k] = (float)v[k];
              u[k] = (float)zrow[itl][k];
            }
            act_grad_n<8>(g, u, a.gnb_act);
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              yy[k] = (float)yrow[itl][k];
              g[k] = (float)v[k];
              u[k] = fmaf(ca[k], yy[k], cbf[k]);
            }
            act_grad_pre_n<8>(g, u, a.gnb_act);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if constexpr (PARK) {
              ebs[k] += g[k];
              ebq[k] = fmaf(g[k], yy[k], ebq[k]);
            } else {
              bs[k] += g[k];
              bq[k] = fmaf(g[k], yy[k], bq[k]);
            }
          }
        }
      } else if (ok && a.gn_partial) {  // statistics of what is stored (the rounded values), exactly like the stand-alone pass
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const eltx2 pr = {v[2 * k], v[2 * k + 1]};
          gs[k] = MEDNET_FDOT2(pr, ones, gs[k], false);
          gq[k] = MEDNET_FDOT2(pr, pr, gq[k], false);
        }
      }
    }
    }
    if constexpr (PARK) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        park[k * 256] = ebs[k];
        park[(8 + k) * 256] = ebq[k];
      }
    }
    // GroupNorm statistics fused into the producer (components.py:57 follows every conv of the 'c g .' orders): one row
    // per wave and brick, or -- accumulate mode -- nothing here: the sums stay in registers until the sample changes
    if (a.gn_partial && !a.stats_accum) flush_any(n, tis * 4 + e_wv, cb);
    if (!has_next) {
      if (a.gn_partial && a.stats_accum) {
        while (acc_n < a.n) {  // the last sample of this wave, then zero rows for the samples after it
          flush_any(acc_n, acc_row, cb);
          ++acc_n;
        }
      }
      break;
    }
    cur_bid = nbid;
    ctile = ntile;
    ccb = ncb2;
  }
}

// ================================================================================================== 32 -> 32 channels
// The layers that hold half of the network's FLOPs (conv2 / conv3 of the two full-resolution ExtResNetBlocks and their
// data gradients, components.py:168-180) have Cin = Cout = 32: ONE channel block and two K chunks, i.e. every work item
// uses the same 55 KB of weights.  The general kernel above re-stages them per item (14 of its 32 staging loads and LDS
// writes per thread and item, 900 MB of L2 -> LDS traffic per launch) and reads every 64-byte voxel row twice, 32 bytes per
// K chunk, one item period apart (1.57x the algorithmic HBM bytes: the second half often comes back from beyond L2).
// This kernel is the specialisation: ONE workgroup of 4 waves per CU, one wave per SIMD with the whole 512-register file;
//   * the weights live in REGISTERS for the workgroup's lifetime (54 A fragments x 4 = 216 registers per lane);
//   * LDS holds only input bricks, whole 64-byte rows (both K chunks), DOUBLE buffered (2 x 69 KB): the next brick's 17
//     pieces per thread are fetched during the tap loop (one load every third tap) and committed to the other buffer, so
//     one barrier per item is all the synchronisation there is;
//   * the epilogue goes through a 4 KB LDS area that is PRIVATE to the wave (a wave owns one z-plane of the brick = whole
//     64-byte rows of 128 voxels), two halves of 64 voxels: no barrier, and the row stores are the same 1 KB-per-instruction
//     stores as above.
// B operand reads are the only LDS traffic of the tap loop: 1 KB per MFMA instead of 1.25 KB.
// Variant bits of the specialisation: what the epilogue does is known at compile time (one wave per SIMD: a runtime
// branch per row or value costs its full latency, nobody else is there to hide it)
enum : int { C32_GNB = 1, C32_ADD = 2, C32_STATS = 4, C32_ACT = 8, C32_SPLIT = 16 /* conv2b only: (high, low) weight images as the two blocks */ };
// The MFMA of conv32_mfma_kernel's tap loop, with its A operand (a weight fragment) read from the ACCUMULATION registers where it
// lives: the kernel keeps 54 fragments = 216 registers for its lifetime, more than half of them in AGPRs, and through the builtin
// hipcc copies each one to VGPRs in front of its MFMA (4 v_accvgpr_read + wait states per step: 154 copies per brick in the tap
// loop of a kernel with one wave per SIMD, where every issue slot between two MFMAs is spoken for).  gfx90a+ MFMAs take A / B from
// either file; the constraint "a" pins the fragments there and the copies disappear.  (The accumulators go to VGPRs instead.)
#ifdef MEDNET_ELT_F16
#define MEDNET_MFMA_ASM_OP "v_mfma_f32_32x32x16_f16"
#else
#define MEDNET_MFMA_ASM_OP "v_mfma_f32_32x32x16_bf16"
#endif
__device__ __forceinline__ void mfma_a_acc(f32x16& acc, const eltx8& w_in_agpr, const eltx8& b) {
  asm volatile(MEDNET_MFMA_ASM_OP " %0, %1, %2, %0" : "+v"(acc) : "a"(w_in_agpr), "v"(b));
}
__device__ __forceinline__ void mfma_a_zero(f32x16& acc, const eltx8& w_in_agpr, const eltx8& b) {
  asm volatile(MEDNET_MFMA_ASM_OP " %0, %1, %2, 0" : "=v"(acc) : "a"(w_in_agpr), "v"(b));
}
template <int V>
__global__ __launch_bounds__(256, 1) void conv32_mfma_kernel(FwdArgs a) {
  constexpr bool GNB = (V & C32_GNB) != 0, ADD = (V & C32_ADD) != 0, STATS = (V & C32_STATS) != 0, ACT = (V & C32_ACT) != 0;
  static_assert(!(GNB && (STATS || ACT)), "the data-gradient variant has its own sums and no activation");
  constexpr int TZ = 4, TY = 8, TX = 16, HZ = 6, HY = 10, HX = 18, NV = HZ * HY * HX, NTW = 4;
  constexpr int PIECES = 4 * NV, IN_ROUNDS = (PIECES + 255) / 256;  // 4320 16-byte pieces of a brick: 17 per thread
  constexpr int NVP = NV + 4, BUF_PIECES = 4 * NVP;                 // plane pitch (in pieces) = 12 mod 16
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  typedef __attribute__((ext_vector_type(2))) elt eltx2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* in_lds = reinterpret_cast<u32x4*>(smem);                           // [2 buffers][4 pieces][NVP]
  elt* wlds = reinterpret_cast<elt*>(smem + (size_t)2 * BUF_PIECES * 16) + (threadIdx.x >> 6) * 2048;  // 4 KB per wave
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wv_s = __builtin_amdgcn_readfirstlane(wv);
  // brick of sequence number q of this workgroup's XCD (workgroup ids 8 apart share an XCD and its L2).  With zslab > 0 the
  // XCD owns zslab consecutive z-layers of bricks of one sample and walks them x fastest, then z, then y: the 32 bricks its
  // CUs work on together are one y-row of the slab, whose halos overlap in x and z, and the next y-row finds the shared two
  // voxel rows still in L2 -- 1.17x the input bytes instead of the 2.1x of isolated bricks.  Otherwise bricks are dealt in
  // linear order (x, y, z, sample), a contiguous run per XCD when the count divides by 8, else interleaved.
  const int xcd = (int)(blockIdx.x & 7);
  auto origin = [&](int q, int& n, int& tz0, int& ty0, int& tx0) {
    if (a.zslab) {
      int qd = fastdiv(q, a.tiles_x, a.rcp_tiles_x);
      tx0 = (q - qd * a.tiles_x) * TX;
      const int yy = fastdiv(qd, a.zslab, a.rcp_zslab);
      const int zz = xcd * a.zslab + (qd - yy * a.zslab);
      ty0 = yy * TY;
      n = fastdiv(zz, a.tiles_z, a.rcp_tiles_z);
      tz0 = (zz - n * a.tiles_z) * TZ;
    } else {
      int tt = a.xcd_chunk ? xcd * a.xcd_chunk + q : q;
      int qd = fastdiv(tt, a.tiles_x, a.rcp_tiles_x);
      tx0 = (tt - qd * a.tiles_x) * TX;
      tt = qd;
      qd = fastdiv(tt, a.tiles_y, a.rcp_tiles_y);
      ty0 = (tt - qd * a.tiles_y) * TY;
      tt = qd;
      qd = fastdiv(tt, a.tiles_z, a.rcp_tiles_z);
      tz0 = (tt - qd * a.tiles_z) * TZ;
      n = qd;
    }
  };
  // fused statistics, kept over the wave's items of a sample (row = 4 * workgroup + wave); workgroups beyond the item
  // count still owe their (zero) rows
  float gs[4], gq[4], bs[GNB ? 8 : 1], bq[GNB ? 8 : 1];
#pragma unroll
  for (int k = 0; k < 4; ++k) gs[k] = gq[k] = 0.f;
#pragma unroll
  for (int k = 0; k < (GNB ? 8 : 1); ++k) bs[k] = bq[k] = 0.f;
  int acc_n = 0;
  const int acc_row = (int)blockIdx.x * 4 + wv;
  auto flush = [&](int nn) {
    const int pjl = lane & 3;
    float* dst = a.gn_partial + (((size_t)nn * a.stats_rows + acc_row) * 32 + pjl * 8) * 2;
    if constexpr (GNB) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        bs[k] = lane_class_sum<4>(bs[k]);
        bq[k] = lane_class_sum<4>(bq[k]);
      }
      if (lane < 4) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          const f32x4 o = {bs[k], bq[k], bs[k + 1], bq[k + 1]};
          *reinterpret_cast<f32x4*>(dst + k * 2) = o;
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) bs[k] = bq[k] = 0.f;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        gs[k] = lane_class_sum<4>(gs[k]);
        gq[k] = lane_class_sum<4>(gq[k]);
      }
      if (lane < 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4 o = {gs[k], gq[k], 0.f, 0.f};
          *reinterpret_cast<f32x4*>(dst + k * 4) = o;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) gs[k] = gq[k] = 0.f;
    }
  };
  // brick sequence of this workgroup: seq, seq + step, ... < end (see origin())
  const int seq_step = a.xcd_chunk ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int seq_end = a.xcd_chunk ? a.xcd_chunk : a.ntiles;
  int seq = a.xcd_chunk ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (seq >= seq_end) {  // (workgroup-uniform) nothing to do but the statistics rows
    if constexpr (GNB || STATS)
      for (int nn = 0; nn < a.n; ++nn) flush(nn);
    return;
  }

  // ---- the weights: A fragment of (chunk kc, tap) = piece ((kc*27 + tap)*2 + h)*32 + r of the packed image
  //      Variants whose epilogue has second operands (the summed gradient, the GroupNorm input) want those rows in flight
  //      BEFORE the tap loop ends -- their latency is otherwise exposed, 2 us per brick -- and have no registers for them:
  //      there the fragments of steps [TW0, TW1) are transient, fetched again (from L2) 30+ steps before their use in every
  //      tap loop, and the second operands take over their registers for the last 8 steps.
  //      Round 3: the GroupNorm input rows (GNB) do not go through registers at all any more.  The K-chunk-0 planes of the brick
  //      being read are dead from step 27 on (steps 27..53 read the chunk-1 planes; B operands are read two steps ahead), and
  //      the buffer is not written again before the next item's barrier: the wave's 8 rows (1 KB each, exactly in lane order)
  //      are fetched by LDS-DMA (buffer_load ... lds) into that space at steps 27..34 -- 2 600 to 3 500 cycles before the
  //      epilogue reads them back, instead of 128 to 1 024 -- and only the summed-gradient rows (ADD) still take over fragment
  //      registers.  Option conv32_gnb_lds=0 (compile-time knob MEDNET_C32_GNB_LDS) restores the register form.
#ifndef MEDNET_C32_GNB_LDS
#define MEDNET_C32_GNB_LDS 1
#endif
  constexpr bool GNB_LDS = GNB && MEDNET_C32_GNB_LDS;
  constexpr int NSEC = (ADD ? 8 : 0) + (GNB && !GNB_LDS ? 8 : 0);  // second-operand rows in registers (4 each, as a fragment)
  constexpr int SEC0 = 54 - 8, WT = NSEC, TW1 = SEC0, TW0 = TW1 - WT;
  eltx8 wreg[54];
  const u32x4* wp = reinterpret_cast<const u32x4*>(a.wpk) + h * 32 + r;
#pragma unroll
  for (int i = 0; i < 54; ++i)  // wreg[] is in the order of USE: step i = (K chunk, kz, kx, ky) multiplies tap tap_at(i % 27)
    if (i < TW0 || i >= TW1) wreg[i] = __builtin_bit_cast(eltx8, wp[((i / 27) * 27 + tap_at(i % 27)) * 64]);

  // ---- staging plan of a brick (halo 6 x 10 x 18 voxels = 60 x-rows of 72 16-byte pieces).  One wave per SIMD means
  //      that whatever runs outside the tap loop is exposed in full and that vector work inside it must stay small, so the
  //      plan is made of SCALARS: in the 15 main rounds wave w loads pieces 0..63 (voxels hx = 0..15) of row 4 * round + w
  //      -- row validity is a scalar (an invalid row gets a resource of 0 bytes: the hardware range check returns zeros),
  //      the row's byte offset goes into the load's scalar offset, and the per-lane part (x position, validity in x) is ONE
  //      register per brick.  Two tail rounds fetch the rows' last 8 pieces (hx = 16, 17), a row per 8 threads.
  constexpr unsigned OOB = 0xFFFFFF00u;
  constexpr int MAIN_ROUNDS = HZ * HY / 4, TAIL_ROUNDS = 2;
  static_assert(MAIN_ROUNDS * 4 == HZ * HY && MAIN_ROUNDS + TAIL_ROUNDS == IN_ROUNDS && TAIL_ROUNDS * 32 >= HZ * HY, "round layout");
  const int q4 = lane & 3, hx_main = lane >> 2;
  int tail_pos[TAIL_ROUNDS];  // hz << 8 | hy of the thread's tail row; rows past the halo fail every range check
#pragma unroll
  for (int k = 0; k < TAIL_ROUNDS; ++k) {
    const int row = 32 * k + (tid >> 3);
    tail_pos[k] = row < HZ * HY ? ((row / HY) << 8) | (row % HY) : 0x7F00;
  }
  const int tail_hx = 16 + ((tid >> 2) & 1);
  u32x4 in_reg[IN_ROUNDS];
  // (mask arithmetic, not a select: a scalar branch would split the tap loop's basic block)
  auto rsrc_of = [&](int n, bool ok) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)n * a.id * a.ih * a.iw * 32), 0, a.bytes_x & (0u - (unsigned)ok), 0x00020000);
  };
  unsigned voff_main = 0;
  auto plan_main = [&](int tx0) {  // the per-lane part of a brick's main rounds
    const int gx = tx0 - 1 + hx_main;
    voff_main = (unsigned)gx < (unsigned)a.iw ? (unsigned)gx * 64u + (unsigned)q4 * 16u : OOB;
  };
  auto tail_voff = [&](int k, int tz0, int ty0, int tx0) {
    const int gz = tz0 - 1 + (tail_pos[k] >> 8), gy = ty0 - 1 + (tail_pos[k] & 255), gxt = tx0 - 1 + tail_hx;
    const bool ok = ((unsigned)gz < (unsigned)a.id) & ((unsigned)gy < (unsigned)a.ih) & ((unsigned)gxt < (unsigned)a.iw);
    return ok ? (unsigned)((gz * a.ih + gy) * a.iw + gxt) * 64u + (unsigned)q4 * 16u : OOB;
  };
  // (a main round's scalar part comes in phases so that the tap loop can put a few scalar instructions into each MFMA gap
  //  instead of all of them into one; phase < 0: everything at once)
  int lr_gz = 0, lr_gy = 0;
  unsigned lr_soff = 0, lr_num = 0;
  auto load_round = [&](int it, int phase, int n, int tz0, int ty0, int tx0, bool valid) {
    if (it < MAIN_ROUNDS) {
      if (phase == 0 || phase < 0) {
        const int row = it * 4 + wv_s;
        const int hz = (row * 205) >> 11, hy = row - hz * HY;  // (row / 10 for row < 1029)
        lr_gz = tz0 - 1 + hz;
        lr_gy = ty0 - 1 + hy;
      }
      if (phase == 1 || phase < 0) {
        const bool ok = valid & ((unsigned)lr_gz < (unsigned)a.id) & ((unsigned)lr_gy < (unsigned)a.ih);
        lr_num = a.bytes_x & (0u - (unsigned)ok);
      }
      if (phase == 2 || phase < 0) lr_soff = (unsigned)((lr_gz * a.ih + lr_gy) * a.iw) * 64u;  // (anything for an invalid row)
      if (phase == 3 || phase < 0)
        in_reg[it] = __builtin_amdgcn_raw_buffer_load_b128(
            __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)n * a.id * a.ih * a.iw * 32), 0, lr_num, 0x00020000), voff_main, lr_soff, 0);
    } else if (phase == 3 || phase < 0) {
      in_reg[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(n, valid), tail_voff(it - MAIN_ROUNDS, tz0, ty0, tx0), 0, 0);
    }
  };
  // LDS slot of a piece: plane q (planes NVP pieces apart: the four planes of a voxel on different banks), voxel row * 18 +
  // hx; tail threads without a row write the zeros their out-of-range load returned to a spare slot (a select, not a
  // branch: the tap loop stays one basic block)
  const int slot_main = q4 * NVP + hx_main;
  int slot_tail[TAIL_ROUNDS];
#pragma unroll
  for (int k = 0; k < TAIL_ROUNDS; ++k) {
    const int row = 32 * k + (tid >> 3);
    slot_tail[k] = row < HZ * HY ? q4 * NVP + row * HX + tail_hx : -1;
  }
  const int spare_slot = (2 * BUF_PIECES * 16 + 4 * 4096) / 16 + tid;
  auto commit_one = [&](int b, int it) {
    if (it < MAIN_ROUNDS) {
      in_lds[b * BUF_PIECES + (it * 4 + wv_s) * HX + slot_main] = in_reg[it];
    } else {
      const int sl = slot_tail[it - MAIN_ROUNDS];
      in_lds[sl >= 0 ? b * BUF_PIECES + sl : spare_slot] = in_reg[it];
    }
  };
  // Operand fragments (round 6).  N-tile t of the wave's z-plane = the x-rows t (lanes 0..15 of a k-half) and t + 4 (lanes 16..31),
  // not 2t and 2t + 1: the tap shifted by ky then needs the rows (t + ky, t + ky + 4) -- the fragment of "tile" t + ky.  SIX fragments
  // a = 0..5 (rows (a, a + 4); a = 4, 5 reach into the halo) therefore serve the three ky taps of all four tiles of a (K chunk, kz,
  // kx): 6 ds_read_b128 per 12 MFMAs instead of 12, without a single shuffle.  A bare loop of this shape runs at 1.63 PFLOP/s against
  // 1.11-1.34 with one read per MFMA (tools/probes/mfma_shape_probe.hip, profiles/r06_mfma_feed_probe.log).  Row t + 4 lies 72
  // voxels = 8 mod 16 behind row t: its lanes are rotated by 8 so that every hardware lane group of a ds_read_b128 ({0-3, 12-15,
  // 20-27}, {4-11, 16-19, 28-31}) hits 16 distinct 16-byte slots.
  const int lx6 = ((r & 15) + 8 * (r >> 4)) & 15;
  const int lbase6 = (wv * HY + 4 * (r >> 4)) * HX + lx6;  // fragment a, tap (kz, ., kx): + (kz * HY + a) * HX + kx
  [[maybe_unused]] const eltx2 ones = {(elt)1.0f, (elt)1.0f};
  const size_t ovol = (size_t)a.od * a.oh * a.ow;

  // ---- the first two bricks: loads in bursts.  From then on brick j + 2 is fetched during brick j's tap loop, a round
  //      every third step, into the registers that brick j + 1's round has just left for LDS: a load has a whole brick
  //      period (> 4 us) to arrive, and the memory system sees an even stream instead of 256 CUs bursting in step
  {
    int n0, tz0, ty0, tx0;
    origin(seq, n0, tz0, ty0, tx0);
    plan_main(tx0);
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) load_round(it, -1, n0, tz0, ty0, tx0, true);
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) commit_one(0, it);
    const bool valid1 = seq + seq_step < seq_end;
    origin(valid1 ? seq + seq_step : 0, n0, tz0, ty0, tx0);
    plan_main(tx0);
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) load_round(it, -1, n0, tz0, ty0, tx0, valid1);
  }
  int buf = 0;
  // (GNB) per-wave copy of the current sample's GroupNorm coefficients: 256 bytes inside the spare-slot area, of which only the
  // slots of threads 224..255 are ever written (spare_slot above)
  [[maybe_unused]] int coef_n = -1;
  [[maybe_unused]] char* cf_lds = smem + (size_t)2 * BUF_PIECES * 16 + 4 * 4096 + wv_s * 256;
  static_assert(4 * 256 <= 224 * 16, "the coefficient copies stay clear of the spare slots in use");
  while (true) {
    int n, tz0, ty0, tx0;
    origin(seq, n, tz0, ty0, tx0);
    const bool has_next = seq + seq_step < seq_end;
    const bool has_next2 = seq + 2 * seq_step < seq_end;
    if constexpr (GNB || STATS) {  // a new sample: the sums so far go out (here, where no accumulator is alive)
      while (acc_n < n) {
        flush(acc_n);
        ++acc_n;
      }
    }
    if constexpr (GNB) {
      // ... and the sample's GroupNorm coefficients (32 channels x {a, b}: 256 bytes) come in, into this wave's own copy in LDS.
      // Fetched per brick from global memory, as until round 4, the four loads sat between the tap loop and the epilogue with
      // nothing to hide their latency behind (one wave per SIMD).
      if (n != coef_n) {
        coef_n = n;
        const auto rsrc_cf = __builtin_amdgcn_make_buffer_rsrc((void*)(a.gnb_coef + (size_t)n * 64), 0, 256u, 0x00020000);
        const u32x4 c4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_cf, (unsigned)(lane * 16), 0, 0);  // (lanes >= 16: zeros)
        wave_lds_fence();  // the previous brick's epilogue has read the old copy
        if (lane < 16) *reinterpret_cast<u32x4*>(cf_lds + lane * 16) = c4;
        wave_lds_fence();
      }
    }
    int n2, tz2, ty2, tx2;  // brick j + 2
    origin(has_next2 ? seq + 2 * seq_step : 0, n2, tz2, ty2, tx2);
    __syncthreads();  // the brick is complete, and every wave is done reading the other buffer (the previous brick)

    // ---- output rows of this wave (its z-plane of the brick, 8 x-rows of 16 voxels)
    const int pj = lane & 3, ev = lane >> 2;  // row reads: voxel ev of 16 in an x-row, 16-byte piece pj
    // rows outside the volume get resources of 0 bytes (loads return zeros, stores are dropped): validity in y and z is
    // scalar, validity in x is the lane's offset
    const bool plane_ok = tz0 + wv_s < a.od;
    const bool lane_ok = tx0 + ev < a.ow;
    const unsigned vb = lane_ok ? (unsigned)(ev * 32 + pj * 8) * 2u : OOB;
    // per row: bytes of its resources (0 = outside) and its scalar offset; made one row per step in the MIDDLE of the tap
    // loop (pinned there: left alone, the ~100 scalar instructions end up in one MFMA gap or behind the loop)
    unsigned row_num[8], row_off[8];
    auto plan_row = [&](int j) {
      const bool ok = plane_ok & (ty0 + j < a.oh);
      row_num[j] = a.bytes_y & (0u - (unsigned)ok);
      row_off[j] = (unsigned)((((tz0 + wv_s) * a.oh + ty0 + j) * a.ow + tx0) * 32) * 2u;
      asm volatile("" : "+s"(row_num[j]), "+s"(row_off[j]));
    };
    auto row_rsrc = [&](const elt* base, int j) {
      return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)n * ovol * 32), 0, row_num[j], 0x00020000);
    };
    auto row_soff = [&](int j) { return row_off[j]; };
    [[maybe_unused]] eltx8 adr[ADD ? 8 : 1], yrw[GNB ? 8 : 1];  // second operands: requested in the last 8 steps below
    // (GNB_LDS: this wave's 8 KB inside the chunk-0 planes of the brick's own buffer, see above)
    [[maybe_unused]] char* ydst = reinterpret_cast<char*>(in_lds + (size_t)buf * BUF_PIECES) + wv_s * 8192;
    static_assert(!GNB_LDS || 4 * 8192 <= 2 * NVP * 16, "the four waves' rows fit the two dead planes");

    // ---- 54 steps (K chunk, tap) of 4 MFMAs; B operands are read TWO steps ahead (nobody else hides the LDS latency).
    //      Step 3k: round k of brick j + 1 goes from its registers to the other LDS buffer, then round k of brick j + 2 is
    //      requested into them.
    f32x16 acc[NTW];
    const u32x4* img = in_lds + (size_t)buf * BUF_PIECES;
    // group gq = (K chunk, kz, kx) = steps 3 gq .. 3 gq + 2 (ky = 0, 1, 2): fragment a of the group
    auto b_fragment = [&](int gq, int a6) {
      const int kc1 = gq / 9, kz1 = (gq / 3) % 3, kx1 = gq % 3;
      return __builtin_bit_cast(eltx8, img[(2 * kc1 + h) * NVP + lbase6 + (kz1 * HY + a6) * HX + kx1]);
    };
    eltx8 xf[2][6];  // two sets: the six reads of group g + 1 are dealt out over the twelve MFMAs of group g
#pragma unroll
    for (int a6 = 0; a6 < 6; ++a6) xf[0][a6] = b_fragment(0, a6);
    // issue order written out and fenced (sched_barrier): one MFMA, then the few other instructions of its gap -- a gap
    // hides ~5 issue slots; the scheduler left to itself packs a step's reads and scalar work into one gap and the
    // matrix pipe idles for the rest
#pragma unroll
    for (int s54 = 0; s54 < 54; ++s54) {
      const int gq = s54 / 3, ky = s54 % 3, cur = gq & 1;
      const int round = s54 / 3;
      const bool staging = s54 % 3 == 0 && round < IN_ROUNDS;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        if (s54 == 0) mfma_a_zero(acc[t], wreg[0], xf[cur][t]);
        else mfma_a_acc(acc[t], wreg[s54], xf[cur][t + ky]);
        __builtin_amdgcn_sched_barrier(0);
        if (gq + 1 < 18 && (t & 1) == 0) xf[cur ^ 1][(ky * 4 + t) >> 1] = b_fragment(gq + 1, (ky * 4 + t) >> 1);
        if (s54 == 0 && t == 1) plan_main(tx2);
        if (t == 3 && s54 >= 24 && s54 < 32) plan_row(s54 - 24);
        if (t == 2 && s54 < WT)
          wreg[TW0 + s54] = __builtin_bit_cast(eltx8, __builtin_nontemporal_load(wp + (((TW0 + s54) / 27) * 27 + tap_at((TW0 + s54) % 27)) * 64));
        if (t == 2 && s54 >= SEC0) {  // second operands of row s54 - SEC0
          const int j = s54 - SEC0;
          if constexpr (ADD) adr[j] = __builtin_bit_cast(eltx8, __builtin_amdgcn_raw_buffer_load_b128(row_rsrc(a.add, j), vb, row_soff(j), 0));
          if constexpr (GNB && !GNB_LDS) yrw[j] = __builtin_bit_cast(eltx8, __builtin_amdgcn_raw_buffer_load_b128(row_rsrc(a.gnb_y, j), vb, row_soff(j), 0));
        }
        if constexpr (GNB_LDS) {
          // The DMAs below overwrite chunk-0 planes that OTHER waves read as B operands (a wave's z-plane needs the halo planes
          // of its neighbours) until THEIR step 26.  This wave is past its own (the MFMAs of step 26 have consumed them), but
          // waves of a workgroup are not in lockstep: one barrier here -- every wave arrives within a few hundred cycles of
          // the others, one wave per SIMD -- makes "dead from step 27 on" true for the workgroup, not only for this wave.
          if (t == 2 && s54 == 27) __builtin_amdgcn_s_barrier();
          if (t == 2 && s54 >= 27 && s54 < 35) {  // GroupNorm input row s54 - 27 -> LDS (row planned at step 24 + row)
            const int j = s54 - 27;
            // (the resource is built here, not by the row_rsrc lambda: handed a lambda's return value, hipcc 7.2 silently drops
            //  the kernel's HOST stub -- the library then fails to load with an undefined kernel symbol)
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.gnb_y + (size_t)n * ovol * 32), 0, row_num[j], 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (__attribute__((address_space(3))) void*)(ydst + j * 1024), 16, vb, row_off[j], 0, 0);
          }
        }
        if (staging) {
          if (t == 0) commit_one(buf ^ 1, round);
          load_round(round, t, n2, tz2, ty2, tx2, has_next2);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: two halves of 64 voxels (4 x-rows) through the wave's private 4 KB of LDS -- no barrier
    if constexpr (ACT) {
#pragma unroll
      for (int t = 0; t < NTW; ++t) act_apply_v16(acc[t], a.act);
    }
    const int e_lx = lx6;
    const int e_sw = (e_lx >> 1) & 7;
    eltx8 rows[8];
    // LDS operations of a wave execute in order, so the hardware needs no barrier between a half's writes, the other lanes'
    // reads of them and the next half's writes.  The COMPILER does: to it the lanes are threads, its alias analysis is per
    // thread, and it may move a lane's read above that lane's own write (it did, in the ConvTranspose epilogue of the split-bf16
    // kernels, commit 29d6f66).  wave_lds_fence() emits no instruction; it pins the order of the memory operations.
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      wave_lds_fence();  // (half 1: the reads of half 0 stay above these writes)
      // rows 4 half .. 4 half + 3 of the plane = the lanes (r >> 4) == half of ALL four tiles (tile t holds rows t and t + 4)
      if ((r >> 4) == half) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const int vl = t * 16 + e_lx;  // voxel of the half: row (0..3) * 16 + x
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            eltx4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (elt)acc[t][q * 4 + j];  // co = 8q + 4h + j
            *reinterpret_cast<eltx4*>(wlds + vl * 32 + ((2 * q + h) ^ e_sw) * 4) = o;
          }
        }
      }
      wave_lds_fence();  // the rows below were written by OTHER lanes of this wave
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int vox = j * 16 + ev, swl = (vox >> 1) & 7;
        rows[4 * half + j] = *reinterpret_cast<const eltx8*>(wlds + vox * 32 + (pj ^ (swl >> 1)) * 8);
      }
    }
    if constexpr (GNB_LDS) {
      // The rows' LDS-DMAs were this wave's vector-memory operations number 27..34 of the tap loop; younger than them are the
      // staging loads of rounds 12..16 (steps 36..48) and, in the ADD variant, the 8 summed-gradient rows (steps 46..53), all
      // issued unconditionally: waiting until at most that many are outstanding has the rows in LDS (nothing but the issuing
      // wave's own vmcnt orders a ds_read behind an LDS-DMA).
      if constexpr (ADD) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) yrw[j] = *reinterpret_cast<const eltx8*>(ydst + j * 1024 + lane * 16);
    }
    float ca[GNB ? 8 : 1], cbf[GNB ? 8 : 1];
    if constexpr (GNB) {  // (after the accumulators are gone: the register file is full until then)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(cf_lds + pj * 64 + q * 16);
        ca[2 * q] = c4[0];
        cbf[2 * q] = c4[1];
        ca[2 * q + 1] = c4[2];
        cbf[2 * q + 1] = c4[3];
      }
    }
    const bool swap_halves = (ev >> 1) & 1;  // (= bit 0 of the row swizzle (vox >> 1) & 7, the same for the eight rows)
    [[maybe_unused]] const bool gnb_elu = a.gnb_act == MEDNET_ACT_ELU;
    [[maybe_unused]] const float gnb_neg = a.gnb_act == MEDNET_ACT_LEAKY ? 0.1f : (a.gnb_act == MEDNET_ACT_RELU ? 0.f : 1.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      eltx8 v = rows[j];
      const eltx8 vs = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);
      v = swap_halves ? vs : v;
      if constexpr (ADD) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (elt)((float)v[k] + (float)adr[j][k]);
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), row_rsrc(a.y, j), vb, row_soff(j), 2);
      [[maybe_unused]] const bool ok = lane_ok & (row_num[j] != 0u);
      if constexpr (GNB) {
        // du = dz * act'(u), u = ca * y + cb, branch-free for a runtime activation (control flow inside the row loop makes
        // the compiler hoist all eight rows' conversions above it, which does not fit): the factor for u <= 0 is exp(u)
        // (ELU), 0.1, 0 or 1.  Two channels per instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): the epilogue of a
        // one-wave-per-SIMD kernel is exposed, instruction for instruction.
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        const eltx8 vz = ok ? v : eltx8{};
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
          const f32x2 yy = {(float)yrw[j][2 * p2], (float)yrw[j][2 * p2 + 1]};
          const f32x2 g = {(float)vz[2 * p2], (float)vz[2 * p2 + 1]};
          const f32x2 ca2 = {ca[2 * p2], ca[2 * p2 + 1]}, cb2 = {cbf[2 * p2], cbf[2 * p2 + 1]};
          const f32x2 u = ca2 * yy + cb2;
          const f32x2 ul = u * 1.44269504088896340736f;  // exp(u) = 2^(u * log2 e), as __expf
          const f32x2 e = {__builtin_amdgcn_exp2f(ul[0]), __builtin_amdgcn_exp2f(ul[1])};
          const f32x2 negc = {gnb_neg, gnb_neg};
          const f32x2 gn = g * (gnb_elu ? e : negc);
          const f32x2 du = {u[0] > 0.f ? g[0] : gn[0], u[1] > 0.f ? g[1] : gn[1]};
          f32x2 s2 = {bs[2 * p2], bs[2 * p2 + 1]}, q2 = {bq[2 * p2], bq[2 * p2 + 1]};
          s2 += du;
          q2 = du * yy + q2;
          bs[2 * p2] = s2[0];
          bs[2 * p2 + 1] = s2[1];
          bq[2 * p2] = q2[0];
          bq[2 * p2 + 1] = q2[1];
        }
      } else if constexpr (STATS) {
        const eltx8 vz = ok ? v : eltx8{};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const eltx2 pr = {vz[2 * k], vz[2 * k + 1]};
          gs[k] = MEDNET_FDOT2(pr, ones, gs[k], false);
          gq[k] = MEDNET_FDOT2(pr, pr, gq[k], false);
        }
      }
    }
    if (!has_next) {
      if constexpr (GNB || STATS) {
        while (acc_n < a.n) {
          flush(acc_n);
          ++acc_n;
        }
      }
      break;
    }
    seq += seq_step;
    buf ^= 1;
  }
}

#include "conv2b_mfma.inc"
#include "convt_dgrad32_mfma.inc"

// ================================================================================================== ConvTranspose3d forward
// out[2j + p] = bias + skip + sum_{taps k of parity class p} W[k] * x[j + delta_k]   (per dim: k=1 -> p=0,d=0; k=0 -> p=1,d=1;
// k=2 -> p=1,d=0).  A workgroup owns a 2x4x16 brick of INPUT voxels (4x8x32 outputs); a wave owns one N-tile of 32 input
// voxels and keeps all 8 output parity classes of it in registers (8 x 16 accumulators), so the staged input brick and
// weight slice are used by all 27 taps exactly as in the forward convolution.  Bias and the decoder's skip tensor
// (`x += encoder_features`, components.py:283-284) are added in the epilogue.
struct CtArgs {
  const elt* x;
  const elt* wpk;
  const float* bias;
  const elt* skip;
  elt* y;
  int n, id, ih, iw;  // input grid (output is 2x)
  int cin, cout;
  int tiles_z, tiles_y, tiles_x, ntiles;
  int nkc, ncb;
  int nkc_in;         // input K chunks; nkc = 2 nkc_in with split weights (the low image as nkc_in more chunks, FwdArgs::lo_delta)
  unsigned lo_delta;
};

__global__ __launch_bounds__(256, 2) void convt_fwd_mfma_kernel(CtArgs a) {
  constexpr int TZ = 2, TY = 4, TX = 16;
  constexpr int HZ = TZ + 1, HY = TY + 1, HX = TX + 1;
  constexpr int NV = HZ * HY * HX;
  constexpr int IN_ROUNDS = (2 * NV + 255) / 256;
  constexpr int W_CHUNKS = 27 * 2 * 32;
  constexpr int W_ROUNDS = (W_CHUNKS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  eltx8* in_lds = reinterpret_cast<eltx8*>(smem);
  eltx8* w_lds = reinterpret_cast<eltx8*>(smem) + 2 * NV;

  const int bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3;
  const int tile = (local / a.ncb) * 8 + xcd;
  const int cb = local % a.ncb;
  if (tile >= a.ntiles) return;
  int tt = tile;
  const int tx0 = (tt % a.tiles_x) * TX;
  tt /= a.tiles_x;
  const int ty0 = (tt % a.tiles_y) * TY;
  tt /= a.tiles_y;
  const int tz0 = (tt % a.tiles_z) * TZ;
  const int n = tt / a.tiles_z;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  long long goff[IN_ROUNDS];
#pragma unroll
  for (int it = 0; it < IN_ROUNDS; ++it) {
    const int p = it * 256 + tid;
    const int v = p >> 1, hh = p & 1;
    long long off = -1;
    if (v < NV) {
      const int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
      const int gz = tz0 + hz, gy = ty0 + hy, gx = tx0 + hx;
      off = (gz < a.id && gy < a.ih && gx < a.iw)
                ? ((((long long)n * a.id + gz) * a.ih + gy) * a.iw + gx) * a.cin + hh * 8
                : -2;
    }
    goff[it] = off;
  }
  const elt* wsrc = a.wpk + (size_t)cb * a.nkc_in * (W_CHUNKS * 8);
  eltx8 in_reg[IN_ROUNDS], w_reg[W_ROUNDS];
  // Branch-free: every lane always loads (padding / out-of-image lanes read a valid dummy address); what is invalid
  // is replaced by zeros when the registers are committed to LDS, so the 16 loads issue back to back with no waits.
  auto prefetch = [&](int kc) {
    const bool low = kc >= a.nkc_in;  // (split weights: the low image's chunks over the same input chunks)
    const int kci = low ? kc - a.nkc_in : kc;
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it)
      in_reg[it] = *reinterpret_cast<const eltx8*>(a.x + (goff[it] >= 0 ? goff[it] : 0) + kci * 16);
    const elt* ws = reinterpret_cast<const elt*>(reinterpret_cast<const char*>(wsrc) + (low ? (size_t)a.lo_delta : 0)) + (size_t)kci * (W_CHUNKS * 8);
#pragma unroll
    for (int it = 0; it < W_ROUNDS; ++it) {
      const int c = it * 256 + tid;
      w_reg[it] = *reinterpret_cast<const eltx8*>(ws + (size_t)(c < W_CHUNKS ? c : W_CHUNKS - 1) * 8);
    }
  };
  auto commit = [&]() {
    const eltx8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) {
      const int p = it * 256 + tid;
      if (goff[it] != -1) in_lds[(p & 1) * NV + (p >> 1)] = goff[it] >= 0 ? in_reg[it] : zero;
    }
#pragma unroll
    for (int it = 0; it < W_ROUNDS; ++it) {
      const int c = it * 256 + tid;
      if (c < W_CHUNKS) w_lds[c] = w_reg[it];
    }
  };

  // the wave's N-tile: rows (lz, ly0 + yy), yy = r >> 4, rotated by the row pitch so the 16 lanes of a read group differ
  const int lz = wv / (TY / 2), ly = (wv % (TY / 2)) * 2 + (r >> 4);
  const int lx = ((r & 15) - (r >> 4) * HX) & 15;
  const int lbase = (lz * HY + ly) * HX + lx + h * NV;

  f32x16 acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;

  prefetch(0);
  for (int kc = 0; kc < a.nkc; ++kc) {
    __syncthreads();
    commit();
    __syncthreads();
    if (kc + 1 < a.nkc) prefetch(kc + 1);
    eltx8 xb[8];
#pragma unroll
    for (int dl = 0; dl < 8; ++dl) xb[dl] = in_lds[lbase + (((dl >> 2) & 1) * HY + ((dl >> 1) & 1)) * HX + (dl & 1)];
    eltx8 wa[2];
    wa[0] = w_lds[h * 32 + r];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
      const int pz = kz != 1, py = ky != 1, px = kx != 1;      // output parity of this tap
      const int dz = kz == 0, dy = ky == 0, dx = kx == 0;      // input offset of this tap
      if (tap + 1 < 27) wa[(tap + 1) & 1] = w_lds[((tap + 1) * 2 + h) * 32 + r];  // next weight fragment in flight
      acc[pz * 4 + py * 2 + px] = MEDNET_MFMA_32x32x16(wa[tap & 1], xb[dz * 4 + dy * 2 + dx],
                                                                          acc[pz * 4 + py * 2 + px], 0, 0, 0);
      if (tap + 1 < 27) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  }

  // ---- epilogue: + bias + skip, through LDS.  A lane holds, per parity class, four 4-channel pieces of ONE output voxel,
  // and the voxels of neighbouring lanes are 128 B apart in memory: written straight from the accumulators that is 64
  // eight-byte requests per instruction (and as many 8-byte reads of the skip tensor).  Instead the 4x8x32 output block is
  // assembled as an fp32 image in LDS, one z-parity at a time (512 voxels x 32 ch x 4 B = 64 KB), and streamed out as whole
  // 64-byte channel rows: 16-byte skip load + add + ONE rounding + 16-byte store per lane, 1 KB contiguous per instruction.
  // 16-byte chunks of a row are XOR-swizzled by the input x index so the ds_write_b128 groups (8 lanes) do not collide.
  float* out_lds = reinterpret_cast<float*>(smem);  // [2 lz][8 oy][32 ox][32 co] floats, reuses the staging images
  float bv[16];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[q * 4 + j] = a.bias ? a.bias[cb * 32 + 8 * q + 4 * h + j] : 0.f;
  const int od = 2 * a.id, oh = 2 * a.ih, ow = 2 * a.iw;
  typedef __attribute__((ext_vector_type(4))) float f4;
#pragma unroll
  for (int pz = 0; pz < 2; ++pz) {
    // the encoder-feature rows this thread adds in this half: requested before the two barriers and the LDS image (behind the
    // barriers each of them waited its whole HBM latency where it was used)
    eltx8 skr[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int vox = (it * 256 + tid) >> 2, piece = tid & 3;
      const int oz = 2 * (tz0 + (vox >> 8)) + pz, oy = 2 * ty0 + ((vox >> 5) & 7), ox = 2 * tx0 + (vox & 31);
      skr[it] = eltx8{0, 0, 0, 0, 0, 0, 0, 0};
      if (a.skip && oz < od && oy < oh && ox < ow)
        skr[it] = *reinterpret_cast<const eltx8*>(a.skip + ((((size_t)n * od + oz) * oh + oy) * ow + ox) * a.cout + cb * 32 + piece * 8);
    }
    __syncthreads();  // MFMA operand reads (first half) / the previous half's row reads are done
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int p = pz * 4 + pp;
      const int vox = (lz * 8 + 2 * ly + (pp >> 1)) * 32 + 2 * lx + (pp & 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = acc[p][q * 4 + j] + bv[q * 4 + j];  // co = 8q + 4h + j
        *reinterpret_cast<f4*>(out_lds + vox * 32 + (((2 * q + h) ^ (lx & 7)) * 4)) = o;
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * 256 + tid;
      const int vox = idx >> 2, piece = idx & 3;
      const int ox_l = vox & 31, oy_l = (vox >> 5) & 7, lz2 = vox >> 8;
      const int key = (ox_l >> 1) & 7;
      const f4 lo = *reinterpret_cast<const f4*>(out_lds + vox * 32 + (((2 * piece) ^ key) * 4));
      const f4 hi = *reinterpret_cast<const f4*>(out_lds + vox * 32 + (((2 * piece + 1) ^ key) * 4));
      const int oz = 2 * (tz0 + lz2) + pz, oy = 2 * ty0 + oy_l, ox = 2 * tx0 + ox_l;
      if (oz < od && oy < oh && ox < ow) {
        const size_t o = ((((size_t)n * od + oz) * oh + oy) * ow + ox) * a.cout + cb * 32 + piece * 8;
        const eltx8 sk = skr[it];
        eltx8 ov;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ov[j] = (elt)(lo[j] + (float)sk[j]);
          ov[4 + j] = (elt)(hi[j] + (float)sk[4 + j]);
        }
        *reinterpret_cast<eltx8*>(a.y + o) = ov;
      }
    }
  }
}

int launch_convt_fwd_mfma(const void* x, const void* sec, const float* bias, const void* skip, void* y, int n, int d,
                          int h, int w, int cin, int cout, hipStream_t s, size_t lo_delta) {
  constexpr size_t lds = 512 * 32 * 4;  // the epilogue's fp32 half-block (64 KB) > staging images (36 KB); 2 workgroups per CU
  static_assert(lds >= ((size_t)2 * 3 * 5 * 17 + 27 * 2 * 32) * 16 && lds <= 80 * 1024, "LDS plan of convt_fwd");
  CtArgs a;
  a.x = (const elt*)x;
  a.wpk = (const elt*)sec;
  a.bias = bias;
  a.skip = (const elt*)skip;
  a.y = (elt*)y;
  a.n = n; a.id = d; a.ih = h; a.iw = w; a.cin = cin; a.cout = cout;
  a.tiles_z = (d + 1) / 2;
  a.tiles_y = (h + 3) / 4;
  a.tiles_x = (w + 15) / 16;
  a.ntiles = n * a.tiles_z * a.tiles_y * a.tiles_x;
  a.nkc_in = cin / 16;
  a.nkc = lo_delta ? 2 * a.nkc_in : a.nkc_in;
  a.lo_delta = (unsigned)lo_delta;
  a.ncb = cout / 32;
  const unsigned grid = (unsigned)((a.ntiles + 7) / 8) * 8 * a.ncb;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)convt_fwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(MEDNET_E_HIP, "convt_fwd_mfma: cannot raise dynamic LDS to %zu", lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(convt_fwd_mfma_kernel, dim3(grid), dim3(256), lds, s, a);
  return check_launch("convt_fwd_mfma");
}

// ================================================================================================== first layer (Cin = 1)
// y[vox][co] = sum_tap W[co][tap] * x[vox + tap - 1]: with one input channel the contraction index is the TAP (27, padded
// to 32 = two MFMA k-steps).  The B operand (k = tap, n = voxel) is gathered from an fp32 halo brick of x in LDS -- 8
// scalar LDS reads per fragment -- and split into elt hi + lo parts (x = hi + lo to ~2^-17), so the network input keeps
// fp32-level precision at 4 MFMAs per 32 voxels; the weights (32 x 32 elt) live in registers for the kernel's lifetime.
// The kernel is bound by writing its output (32 channels per input voxel); the VALU formulation it replaces was 4x slower.
// Round 6: PERSISTENT.  With one short-lived workgroup per brick, 63 % of the instruction stream was per-brick fixed cost -- the
// weight fragments and tap offsets (21 %) and the reduction of the 32 statistics sums over the wave (42 %) -- in a kernel that
// is instruction-bound (252 us for 537 MB of output).  Now at most 4 workgroups per CU walk the (brick, channel block) items:
// weights and offsets are made once, the next brick's halo values are in flight (registers) while the current brick is on the
// matrix cores, and a wave keeps its sums over all its bricks of a sample: one partial row per wave, workgroup and sample.
// The output is bit-identical to the one-brick-per-workgroup form (same MFMA sequence per voxel); option conv_c1_persist=0
// launches a workgroup per item.
struct C1Args {
  const float* x;    // N x D x H x W (one channel)
  const float* w;    // packed forward image Pf[tap][co] (fp32)
  elt* y;           // NDHWC
  float* gn_partial; // nullable: [n][4 * gridDim.x / ncb][cout][2] per-wave {sum y, sum y^2} of the stored values
  int n, d, h, w_, cout;
  int tiles_z, tiles_y, tiles_x, ntiles, ncb;
  unsigned rcp_tiles_x, rcp_tiles_y, rcp_tiles_z, rcp_ncb;
  int x16;  // 1: x holds elt values (the 1-channel output of a GroupNorm in the 'gcr' orders), else fp32 (the network input)
  int split;  // split weights (MEDNET_ALGO_SPLITW_BIT): the weights' low parts elt(w - elt(w)) are multiplied too
  unsigned bytes_x;  // one sample of x
};

// SPLIT: the weights' low parts as a third MFMA per k-step (split weights); compiled apart so that the default form keeps no
// registers for them (128 per lane at four workgroups per CU)
template <bool SPLIT>
__global__ __launch_bounds__(256, 4) void conv_c1_mfma_kernel(C1Args a) {
  constexpr int TZ = 4, TY = 8, TX = 16, HZ = TZ + 2, HY = TY + 2, HX = TX + 2, NV = HZ * HY * HX, NTW = 4;
  static_assert(NTW == TY / 2 && TZ == 4, "a wave owns one z-plane of the brick");
  constexpr int IN_ROUNDS = (NV + 255) / 256;
  constexpr unsigned OOB = 0xFFFFFF00u;
  // the halo brick of the input, ALREADY split into its elt high and low parts (x = hi + lo to ~2^-17): every halo value feeds up to 27
  // taps, and split where it is gathered (round 1-5) the two conversions and the subtraction ran once per tap, tile and k-step
  __shared__ elt xs_hi[NV], xs_lo[NV];
  __shared__ __attribute__((aligned(16))) elt epi[4 * 1024];  // per wave: one tile of 32 voxels x 32 channels on its way out
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  // items (brick, channel block) = item / ncb, item % ncb; gridDim.x is a multiple of ncb: a workgroup keeps its channel block
  const int cb = (int)(blockIdx.x % a.ncb);
  const int nitems = a.ntiles * a.ncb;
  // weights: A operand, lane (co = r, h) holds taps 8h..8h+7 (k-step 0) and 16+8h..16+8h+7 (k-step 1); taps >= 27 are 0
  eltx8 wa[2], wl[SPLIT ? 2 : 1];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tap = ks * 16 + 8 * h + j;
      const float wf = tap < 27 && cb * 32 + r < a.cout ? a.w[(size_t)tap * a.cout + cb * 32 + r] : 0.f;  // (a 16-channel layer fills half a block)
      wa[ks][j] = (elt)wf;
      if constexpr (SPLIT) wl[ks][j] = (elt)(wf - (float)wa[ks][j]);
    }
  // LDS offsets of this lane's 8 taps per k-step
  int toff[2][8];  // (both k-halves' offsets are compile-time constants: one select per entry instead of the divisions by 9 and 3)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t0 = ks * 16 + j, t1 = ks * 16 + 8 + j;
      const int o0 = t0 < 27 ? ((t0 / 9) * HY + (t0 / 3) % 3) * HX + t0 % 3 : 0, o1 = t1 < 27 ? ((t1 / 9) * HY + (t1 / 3) % 3) * HX + t1 % 3 : 0;
      toff[ks][j] = h ? o1 : o0;
    }
  // this thread's halo positions (the same for every brick): hz << 16 | hy << 8 | hx; slots past the halo fail every range check
  int hpos[IN_ROUNDS];
#pragma unroll
  for (int k = 0; k < IN_ROUNDS; ++k) {
    const int i = tid + 256 * k;
    hpos[k] = i < NV ? ((i / (HX * HY)) << 16) | (((i / HX) % HY) << 8) | (i % HX) : 0x7FFF0000;
  }
  auto origin = [&](int item, int& n, int& tz0, int& ty0, int& tx0) {
    int tt = fastdiv(item, a.ncb, a.rcp_ncb);
    int qd = fastdiv(tt, a.tiles_x, a.rcp_tiles_x);
    tx0 = (tt - qd * a.tiles_x) * TX;
    tt = qd;
    qd = fastdiv(tt, a.tiles_y, a.rcp_tiles_y);
    ty0 = (tt - qd * a.tiles_y) * TY;
    tt = qd;
    qd = fastdiv(tt, a.tiles_z, a.rcp_tiles_z);
    tz0 = (tt - qd * a.tiles_z) * TZ;
    n = qd;
  };
  // halo values of a brick: buffer loads through a per-sample resource, positions outside the volume get an out-of-range offset
  // and come back as zeros
  float xin[IN_ROUNDS];
  auto fetch = [&](int item, bool valid) {
    int n, tz0, ty0, tx0;
    origin(item, n, tz0, ty0, tx0);
    const unsigned esz = a.x16 ? 2u : 4u;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(a.x) + (size_t)n * a.bytes_x), 0,
                                                        a.bytes_x & (0u - (unsigned)valid), 0x00020000);
#pragma unroll
    for (int k = 0; k < IN_ROUNDS; ++k) {
      const int gz = tz0 - 1 + (hpos[k] >> 16), gy = ty0 - 1 + ((hpos[k] >> 8) & 255), gx = tx0 - 1 + (hpos[k] & 255);
      const bool in_vol = ((unsigned)gz < (unsigned)a.d) & ((unsigned)gy < (unsigned)a.h) & ((unsigned)gx < (unsigned)a.w_);
      const unsigned off = in_vol ? (unsigned)((gz * a.h + gy) * a.w_ + gx) * esz : OOB;
      if (a.x16) {  // (workgroup-uniform)
        const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(rsrc, off, 0, 0);
        xin[k] = (float)__builtin_bit_cast(elt, u);
      } else {
        xin[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < IN_ROUNDS; ++k) {
      const int i = tid + 256 * k;
      if (i < NV) {
        const elt xh = (elt)xin[k];
        xs_hi[i] = xh;
        xs_lo[i] = (elt)(xin[k] - (float)xh);
      }
    }
  };
  const size_t vol = (size_t)a.d * a.h * a.w_;
  // fused GroupNorm statistics, as in conv_mfma_kernel: per channel PAIR (v_dot2c_f32: two exact products + fp32 add per
  // instruction), taken from the stored 64-byte rows -- lane = (voxel, 16-byte piece lane & 3) -- over the wave's tiles of a sample;
  // entry 2j of the partial row gets the sums of channels 2j and 2j + 1, entry 2j + 1 is zero (GroupNorm only adds the channels of a
  // group; the host asks for fused partials only when the channels per group are even).  Until round 6: 32 per-channel sums per
  // lane from the accumulators, reduced over the wave once per BRICK -- 42 % of the kernel's instructions.
  typedef __attribute__((ext_vector_type(2))) elt eltx2;
  const eltx2 ones = {(elt)1.0f, (elt)1.0f};
  float gs[4], gq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) gs[k] = gq[k] = 0.f;
  int acc_n = 0;
  auto flush = [&](int nn) {  // one row per wave: sum over the 16 lanes that share a piece (DPP / v_permlane steps: plain VALU)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gs[k] = lane_class_sum<4>(gs[k]);
      gq[k] = lane_class_sum<4>(gq[k]);
    }
    const int pjl = lane & 3;
    if (lane < 4 && cb * 32 + pjl * 8 < a.cout) {
      const int rows = 4 * (int)(gridDim.x / a.ncb);
      float* dst = a.gn_partial + (((size_t)nn * rows + (blockIdx.x / a.ncb) * 4 + wv) * a.cout + cb * 32 + pjl * 8) * 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 o = {gs[k], gq[k], 0.f, 0.f};
        *reinterpret_cast<f32x4*>(dst + k * 4) = o;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) gs[k] = gq[k] = 0.f;
  };

  int item = blockIdx.x;  // (the launcher never starts more workgroups than items)
  fetch(item, true);
  while (true) {
    int n, tz0, ty0, tx0;
    origin(item, n, tz0, ty0, tx0);
    const int nitem = item + (int)gridDim.x;
    const bool has_next = nitem < nitems;
    if (a.gn_partial) {
      while (acc_n < n) {  // (workgroup-uniform) a new sample: the finished one's row goes out, zero rows for skipped samples
        flush(acc_n);
        ++acc_n;
      }
    }
    __syncthreads();  // every wave is done gathering from the previous brick
    commit();
    __syncthreads();
    fetch(has_next ? nitem : item, has_next);  // in flight while this brick is on the matrix cores
    // (a rolled loop: unrolled inside the item loop, the compiler hoists the 4 x 16 gather addresses, which are the same for every
    //  brick, out of it and spills 181 registers at the 128 that four workgroups per CU leave)
#pragma unroll 1
    for (int t = 0; t < NTW; ++t) {
      const int g = wv * NTW + t;
      const int lz = wv, ly = t * 2 + (r >> 4), lx = r & 15;  // (N-tile g = wv * 4 + t: z-plane g / 4, rows 2 (g % 4), + 1)
      const int oz_t = tz0 + lz;
      const int base = (lz * HY + ly) * HX + lx;
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        eltx8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          hi[j] = xs_hi[base + toff[ks][j]];
          lo[j] = xs_lo[base + toff[ks][j]];
        }
        acc = MEDNET_MFMA_32x32x16(wa[ks], hi, acc, 0, 0, 0);
        acc = MEDNET_MFMA_32x32x16(wa[ks], lo, acc, 0, 0, 0);
        if constexpr (SPLIT) acc = MEDNET_MFMA_32x32x16(wl[ks], hi, acc, 0, 0, 0);
      }
      // The accumulator layout gives a lane four 8-byte pieces (channels 8q + 4h ..) of ITS voxel's 64-byte row: stored as they
      // stand, one instruction touches 64 rows with 8 bytes each, and this kernel does little else than store.
      // The tile goes through 2 KB of LDS private to the wave (8-byte pieces XOR-swizzled by voxel: conflict-free both ways, no
      // barrier -- a wave's LDS operations execute in order, the fence keeps the compiler from reordering them) and leaves as whole
      // rows: 4 lanes per voxel, 16 voxels = one x-row of the brick = 1 KB contiguous per instruction when Cout = 32.
      elt* tile_lds = epi + wv * 1024;
      wave_lds_fence();  // (the previous tile's row reads stay above these writes)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        eltx4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (elt)acc[q * 4 + j];
        *reinterpret_cast<eltx4*>(tile_lds + r * 32 + (((2 * q + h) ^ ((r >> 2) & 7)) * 4)) = o;
      }
      wave_lds_fence();
      eltx8 rows2[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = i * 16 + (lane >> 2), sw = (v >> 2) & 7;
        eltx8 rv = *reinterpret_cast<const eltx8*>(tile_lds + v * 32 + (((lane & 3) ^ (sw >> 1)) * 8));
        if (sw & 1) rv = __builtin_shufflevector(rv, rv, 4, 5, 6, 7, 0, 1, 2, 3);
        rows2[i] = rv;
      }
      wave_lds_fence();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = i * 16 + (lane >> 2);
        const int sy = ty0 + (g % (TY / 2)) * 2 + (v >> 4), sx = tx0 + (v & 15);
        const bool ok = oz_t < a.d && sy < a.h && sx < a.w_ && cb * 32 + (lane & 3) * 8 < a.cout;
        if (ok)
          __builtin_nontemporal_store(rows2[i], reinterpret_cast<eltx8*>(a.y + ((size_t)n * vol + ((size_t)oz_t * a.h + sy) * a.w_ + sx) * a.cout + cb * 32 + (lane & 3) * 8));
        const eltx8 vz = ok ? rows2[i] : eltx8{};  // statistics of what is stored
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const eltx2 pr = {vz[2 * k], vz[2 * k + 1]};
          gs[k] = MEDNET_FDOT2(pr, ones, gs[k], false);
          gq[k] = MEDNET_FDOT2(pr, pr, gq[k], false);
        }
      }
    }
    if (!has_next) break;
    item = nitem;
  }
  if (a.gn_partial) {
    while (acc_n < a.n) {  // the last sample of this workgroup, then zero rows for the samples after it
      flush(acc_n);
      ++acc_n;
    }
  }
}

bool conv_c1_mfma_supported(int cin, int cout, int ksize, int x_dtype, int y_dtype, int y_layout, bool bias) {
  return cin == 1 && ksize == 3 && cout % 16 == 0 && (x_dtype == MEDNET_F32 || x_dtype == ELT_DTYPE) && y_dtype == ELT_DTYPE &&
         y_layout == MEDNET_NDHWC && !bias;
}
// workgroups of a launch: 4, 3 or 2 per CU -- whichever leaves the last round of the walk fullest (the item count of config 5's
// first layer, 14 400, is 14.06 rounds of 1024 workgroups: 15 rounds with the last one 6 % full, measured 6 % slower than 18.75
// rounds of 768) -- rounded down to a multiple of the channel-block count so that a workgroup keeps its block; one per (brick,
// channel block) item when there are no more items than that, or with option conv_c1_persist=0
static int conv_c1_grid(int n, int d, int h, int w, int cout) {
  const int ncb = (cout + 31) / 32;
  const int nitems = n * ((d + 3) / 4) * ((h + 7) / 8) * ((w + 15) / 16) * ncb;
  const int cus = ::mednet_internal_cu_count() > 0 ? ::mednet_internal_cu_count() : 256;
  if (!tuning_option("conv_c1_persist", 1) || nitems <= 4 * cus) return nitems;
  int best = 0;
  double best_fill = 0.0;
  for (int per_cu = 4; per_cu >= 2; --per_cu) {
    const int g = per_cu * cus / ncb * ncb;
    if (g <= 0) continue;
    const double fill = (double)nitems / ((double)((nitems + g - 1) / g) * g);
    if (fill > best_fill + 0.01) {  // (more workgroups per CU hide more latency: fewer only for a clearly fuller last round)
      best = g;
      best_fill = fill;
    }
  }
  return best > 0 ? best : nitems;
}
int conv_c1_stats_chunks(int n, int d, int h, int w, int cout) { return 4 * (conv_c1_grid(n, d, h, w, cout) / ((cout + 31) / 32)); }
int launch_conv_c1_mfma(const void* x, const float* w_pf, void* y, int n, int d, int h, int w, int cout, float* gn_partial,
                        hipStream_t s, int x_dtype, int split) {
  C1Args a;
  a.split = split;
  a.gn_partial = gn_partial;
  a.x = (const float*)x;
  a.x16 = x_dtype != MEDNET_F32;
  a.w = w_pf;
  a.y = (elt*)y;
  a.n = n; a.d = d; a.h = h; a.w_ = w; a.cout = cout;
  a.tiles_z = (d + 3) / 4;
  a.tiles_y = (h + 7) / 8;
  a.tiles_x = (w + 15) / 16;
  a.ntiles = n * a.tiles_z * a.tiles_y * a.tiles_x;
  a.ncb = (cout + 31) / 32;
  auto rcp = [](int dd) { return dd == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)dd - 1) / (unsigned)dd); };
  a.rcp_tiles_x = rcp(a.tiles_x); a.rcp_tiles_y = rcp(a.tiles_y); a.rcp_tiles_z = rcp(a.tiles_z); a.rcp_ncb = rcp(a.ncb);
  MEDNET_REQUIRE((double)a.ntiles * a.ncb * 1024.0 < 4294967296.0, MEDNET_E_UNSUPPORTED, "conv_c1_mfma: grid too large");
  MEDNET_REQUIRE((double)d * h * w * 4.0 < 4294960000.0, MEDNET_E_UNSUPPORTED, "conv_c1_mfma: one input sample must stay below 4 GB");
  a.bytes_x = (unsigned)((size_t)d * h * w * (a.x16 ? 2 : 4));
  const dim3 grid((unsigned)conv_c1_grid(n, d, h, w, cout));
  if (split) hipLaunchKernelGGL(conv_c1_mfma_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(conv_c1_mfma_kernel<false>, grid, dim3(256), 0, s, a);
  return check_launch("conv_c1_mfma");
}

// ================================================================================================== weight packing
// element e of section [cb][kc][tap][h][co][j]  <-  Weff[cb*32+co][kc*16+h*8+j][tap]
// mode 0: conv fwd      Weff[m][k][t] = W[m][k][t]            (W: Cout,Cin,27)  M=Cout K=Cin
// mode 1: conv dgrad    Weff[m][k][t] = W[k][m][26-t]                           M=Cin  K=Cout
// mode 2: convT fwd     Weff[m][k][t] = Wt[k][m][t]           (Wt: Cin,Cout,27) M=Cout K=Cin
// mode 3: convT dgrad   Weff[m][k][t] = Wt[m][k][t]                             M=Cin  K=Cout
// A launch's blocks per layer: one block per (channel block, K chunk) TILE of the forward image, then the tiles of the
// data-gradient image.  A tile is 32 rows x 16 contraction channels x 27 taps = 13 824 elements that are CONTIGUOUS in the
// fragment image; in the PyTorch tensor they are 32 (or 16) runs of 432 (864) consecutive floats.  The block reads the runs in
// source order into LDS and writes the image in image order -- high and, in the fp32 storage mode, low elements from the same
// LDS copy -- so both sides are coalesced.  The forward image's tile also holds what the two fp32 tap-major images of the
// direct / exact-product kernels need of these 32 x 16 channels: they leave as 128-byte (Pf) and 64-byte (Pb) segments.
// (The element-per-thread form gathered every image element with a stride of 27 floats and scattered the fp32 images four bytes
// at a time: 0.21 / 0.30 ms per step for the 20 layers of config 2.)
constexpr int PACK_TILE = 32 * 16 * 27;
__host__ __device__ inline unsigned pack_img_tiles(int cin, int cout) {  // the larger of the two images' tile counts
  const unsigned f = (unsigned)((cout + 31) / 32) * (unsigned)(cin / 16), b = (unsigned)((cin + 31) / 32) * (unsigned)(cout / 16);
  return f > b ? f : b;
}
__device__ __forceinline__ void pack_mfma_body(const float* __restrict__ w, elt* __restrict__ out0, elt* __restrict__ out1,
                                               int M0, int K0, int mode0, int mode1, float* __restrict__ Pf,
                                               float* __restrict__ Pb, int transposed_src, size_t lo_delta, unsigned blk,
                                               float* __restrict__ tile /* LDS, PACK_TILE floats */) {
  const unsigned ntmax = pack_img_tiles(K0, M0);
  const unsigned tq = blk;
  const int img = (int)(tq / ntmax);  // 0: forward image (M0 x K0), 1: data-gradient image (K0 x M0)
  if (img > 1) return;
  const int M = img ? K0 : M0, K = img ? M0 : K0, mode = img ? mode1 : mode0;
  const int nkc = K / 16;
  const unsigned tl = tq % ntmax;
  if (tl >= (unsigned)((M + 31) / 32) * (unsigned)nkc) return;
  const int kc = (int)(tl % nkc), cb = (int)(tl / nkc);
  // source order -> LDS in image order: d = ((tap * 2 + h) * 32 + co) * 8 + j, k = 8 h + j
  // (16-byte loads: a run starts at a multiple of 16 * 27 floats and is 432 or 864 floats long)
  typedef __attribute__((ext_vector_type(4))) float pf4;
  const bool al16 = (reinterpret_cast<size_t>(w) & 15) == 0;  // (a caller's parameter pointer may be only 4-byte aligned)
  if (mode == 0 || mode == 3) {  // w[m][k][t]: per row m a run of 16 * 27 floats
#pragma unroll 2
    for (int s4 = threadIdx.x; s4 < PACK_TILE / 4; s4 += 256) {
      const int ml = s4 / 108, rem0 = (s4 % 108) * 4;
      const int m = cb * 32 + ml;
      pf4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < M) {
        const float* src = w + ((size_t)m * K + kc * 16) * 27 + rem0;
        if (al16) v = *reinterpret_cast<const pf4*>(src);
        else v = pf4{src[0], src[1], src[2], src[3]};
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kl = (rem0 + e) / 27, tap = (rem0 + e) % 27;
        tile[((tap * 2 + (kl >> 3)) * 32 + ml) * 8 + (kl & 7)] = v[e];
      }
    }
  } else {  // w[k][m][t] (mode 1: taps reversed): per k a run of 32 * 27 floats (rows past M: zeros)
#pragma unroll 2
    for (int s4 = threadIdx.x; s4 < PACK_TILE / 4; s4 += 256) {
      const int kl = s4 / 216, rem0 = (s4 % 216) * 4;
      pf4 v = {0.f, 0.f, 0.f, 0.f};
      // (M is a multiple of 16: a run of valid rows ends on a 16-byte boundary)
      if (cb * 32 + (rem0 + 3) / 27 < M) {
        const float* src = w + ((size_t)(kc * 16 + kl) * M + cb * 32) * 27 + rem0;
        if (al16) v = *reinterpret_cast<const pf4*>(src);
        else v = pf4{src[0], src[1], src[2], src[3]};
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ml = (rem0 + e) / 27, t = (rem0 + e) % 27;
        const int tap = mode == 1 ? 26 - t : t;
        tile[((tap * 2 + (kl >> 3)) * 32 + ml) * 8 + (kl & 7)] = cb * 32 + ml < M ? v[e] : 0.f;
      }
    }
  }
  __syncthreads();
  elt* out = (img ? out1 : out0) + (size_t)tl * PACK_TILE;  // (tl = cb * nkc + kc: the image's tile order)
  elt* out_lo = lo_delta ? reinterpret_cast<elt*>(reinterpret_cast<char*>(out) + lo_delta) : nullptr;
  for (int d = threadIdx.x * 8; d < PACK_TILE; d += 256 * 8) {
    eltx8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = tile[d + j];
      hi[j] = (elt)v;
      lo[j] = (elt)(v - (float)hi[j]);
    }
    *reinterpret_cast<eltx8*>(out + d) = hi;
    if (out_lo) *reinterpret_cast<eltx8*>(out_lo + d) = lo;  // the LOW image of the split-bf16 contraction (conv_x3_mfma.hip)
  }
  if (img == 0 && Pf) {  // Pf[t][ci][co] and Pb[t'][co][ci] (t' = 26 - t for a conv, t for a ConvTranspose) of this tile's channels
    const int cin = K0, cout = M0;
    // 16-byte stores (cout and cin are multiples of 16, the images 256-byte aligned inside the pack buffer)
    for (int i4 = threadIdx.x; i4 < PACK_TILE / 4; i4 += 256) {
      const int co = (i4 & 7) * 4, kl = (i4 >> 3) & 15, t = i4 >> 7;
      if (cb * 32 + co < cout) {
        const float* src = tile + ((t * 2 + (kl >> 3)) * 32 + co) * 8 + (kl & 7);
        *reinterpret_cast<pf4*>(Pf + ((size_t)t * cin + kc * 16 + kl) * cout + cb * 32 + co) = pf4{src[0], src[8], src[16], src[24]};
      }
    }
    for (int i4 = threadIdx.x; i4 < PACK_TILE / 4; i4 += 256) {
      const int k4 = (i4 & 3) * 4, co = (i4 >> 2) & 31, t = i4 >> 7;
      if (cb * 32 + co < cout)
        *reinterpret_cast<pf4*>(Pb + ((size_t)(transposed_src ? t : 26 - t) * cout + cb * 32 + co) * cin + kc * 16 + k4) =
            *reinterpret_cast<const pf4*>(tile + ((t * 2 + (k4 >> 3)) * 32 + co) * 8 + (k4 & 7));
    }
  }
}
__global__ __launch_bounds__(256) void pack_mfma_kernel(const float* __restrict__ w, elt* __restrict__ out0,
                                                        elt* __restrict__ out1, int M0, int K0, int mode0, int mode1,
                                                        float* __restrict__ Pf, float* __restrict__ Pb, int transposed_src,
                                                        size_t lo_delta) {
  __shared__ float tile[PACK_TILE];
  pack_mfma_body(w, out0, out1, M0, K0, mode0, mode1, Pf, Pb, transposed_src, lo_delta, blockIdx.x, tile);
}
// all 3x3x3 layers of a network in ONE launch (after the optimizer step every layer's weights have moved): a block finds its
// layer in the device table built once (mednet_conv3d_pack_table), whose entries carry the first block of every layer
__global__ __launch_bounds__(256) void pack_mfma_many_kernel(const PackJobDev* __restrict__ jobs, int njobs, int with_low) {
  __shared__ float tile[PACK_TILE];
  int lo = 0, hi = njobs - 1;  // the last layer whose first block is <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const PackJobDev j = jobs[lo];
  const unsigned blk = blockIdx.x - j.first_block;
  if (blk >= j.nblocks) return;
  // with_low bit 0: the low 16-bit images too; bit 1 (MEDNET_PACK_HIGH_ONLY): no fp32 images
  pack_mfma_body(j.w, (elt*)j.sec_fwd, (elt*)j.sec_bwd, j.cout, j.cin, j.transposed ? 2 : 0, j.transposed ? 3 : 1,
                 (with_low & 2) ? nullptr : j.Pf, j.Pb, j.transposed, (with_low & 1) ? (size_t)j.lo_delta : 0, blk, tile);
}

PackLayout pack_layout(int cin, int cout, int ksize) {
  PackLayout L;
  L.taps = ksize * ksize * ksize;
  const size_t f32 = align_up((size_t)L.taps * cin * cout * sizeof(float), 256);
  L.f32_fwd = 0;
  L.f32_bwd = f32;
  // the contraction side of an image needs a multiple of 16 (one MFMA k-step), its M side is padded to 32 rows
  const bool mfma = ksize == 3 && cin % 16 == 0 && cout % 16 == 0;
  const size_t fwd_bytes = mfma ? align_up((size_t)27 * ((cout + 31) / 32 * 32) * cin * sizeof(elt), 256) : 0;
  const size_t bwd_bytes = mfma ? align_up((size_t)27 * ((cin + 31) / 32 * 32) * cout * sizeof(elt), 256) : 0;
  L.mfma_bytes = fwd_bytes > bwd_bytes ? fwd_bytes : bwd_bytes;
  L.mfma_fwd = 2 * f32;
  L.mfma_bwd = 2 * f32 + fwd_bytes;
  // the low images of the split-bf16 contraction (written only when the pack is asked for them: fp32 storage mode)
  L.lo_delta = fwd_bytes + bwd_bytes;
  L.total = 2 * f32 + 2 * (fwd_bytes + bwd_bytes) + 256;
  return L;
}

unsigned pack_mfma_blocks(int cin, int cout) { return 2 * pack_img_tiles(cin, cout); }
int launch_pack_mfma(const float* w, void* sec_fwd, void* sec_bwd, float* Pf, float* Pb, int cin, int cout, int T,
                     int transposed_src, hipStream_t s, size_t lo_delta) {
  if (T != 27) return MEDNET_OK;
  // forward image: M = cout, K = cin; backward (data-gradient) image: M = cin, K = cout
  hipLaunchKernelGGL(pack_mfma_kernel, dim3(pack_mfma_blocks(cin, cout)), dim3(256), 0, s, w, (elt*)sec_fwd, (elt*)sec_bwd, cout, cin,
                     transposed_src ? 2 : 0, transposed_src ? 3 : 1, Pf, Pb, transposed_src, lo_delta);
  return check_launch("pack_mfma");
}
int launch_pack_mfma_many(const void* table_device, int njobs, unsigned max_blocks, hipStream_t s, int with_low) {
  hipLaunchKernelGGL(pack_mfma_many_kernel, dim3(max_blocks), dim3(256), 0, s, (const PackJobDev*)table_device, njobs, with_low);
  return check_launch("pack_mfma_many");
}

// the kernels address ONE SAMPLE of the activation tensor through a buffer resource with 32-bit byte offsets
bool conv_mfma_fits(int n, int d, int h, int w, int c) {
  (void)n;
  return (double)d * h * w * c * 2.0 < 4294960000.0;
}
bool conv_mfma_supported(int cin, int cout, int ksize, int x_dtype, int y_dtype, int x_layout, int y_layout, bool bias) {
  return ksize == 3 && cin % 16 == 0 && cout % 16 == 0 && x_dtype == ELT_DTYPE && y_dtype == ELT_DTYPE &&
         x_layout == MEDNET_NDHWC && y_layout == MEDNET_NDHWC && !bias;
}

// Rows per sample of the fused GroupNorm partials and whether the persistent kernel accumulates over a workgroup's items.
// Accumulate mode needs: the persistent launch form (more items than the 512 resident workgroups, with margin so that every
// workgroup starts on a valid item), brick count a multiple of 8 (no padding items), and ncb | 64 (a workgroup's channel
// block never changes).
// The 32 -> 32 specialisation (conv32_mfma_kernel): one workgroup per CU, at least two bricks each.
// Its grid is exactly 256 workgroups and conv_stats_plan hands out 256 * 4 statistics rows: that is the MI355X's CU count.
// On a device with another count (a partitioned part) the general kernel takes the call.
static bool conv32_applies(int ntiles, int cin, int cout) {
  return cin == 32 && cout == 32 && ntiles >= 512 && ::mednet_internal_cu_count() == 256 && tuning_option("conv32", 1);
}
// (A/B knob conv32_gnb=0: the data-gradient launches that also take a GroupNorm backward's first pass go to the general kernel,
//  whose second workgroup per CU hides that epilogue; the specialisation's single wave per SIMD cannot)
// (split weights: the specialisation keeps ONE image of the layer's weights in registers -- conv2b's split form takes those calls)
static bool conv32_takes(int ntiles, int cin, int cout, bool gnb, bool split = false) {
  return !split && conv32_applies(ntiles, cin, cout) && (!gnb || tuning_option("conv32_gnb", 1));
}
// The two-block kernel (conv2b_mfma_kernel, round 6): layers whose output channels come in pairs of 32-channel blocks, on 16-wide
// bricks, with enough (brick, block pair) items to give each of the 256 one-per-CU workgroups work in nearly full rounds (the
// last round of a persistent grid runs with whatever is left: 288 items would keep 7/8 of the chip idle for half of the launch).
// Split weights: its (high, low) form takes every layer whose output channels come in whole 32-channel blocks, whatever the item count.
static bool conv2b_takes(int ntiles, int cin, int cout, bool split = false) {
  if (split) return cout % 32 == 0 && cin % 16 == 0 && ::mednet_internal_cu_count() == 256 && tuning_option("conv2b_split", 1);
  // (from 4 K chunks on: with two -- 32 input channels -- an item is too short for its fixed costs, measured 0.95-0.99 of the general kernel)
  if (!(cout % 64 == 0 && cin % 16 == 0 && cin >= tuning_option("conv2b_min_cin", 64) && ::mednet_internal_cu_count() == 256 &&
        tuning_option("conv2b", 1)))
    return false;
  const int nitems = ((ntiles + 7) / 8) * 8 * (cout / 64);
  const int rounds = (nitems + 255) / 256;
  return nitems >= 256 && nitems * 100 >= tuning_option("conv2b_min_fill", 80) * rounds * 256;
}
// Brick kind of a stride-1 launch: 3 (8-wide bricks) where that covers the volume with fewer voxel slots, unless the 32 -> 32
// specialisation (16-wide bricks only) takes the call
static int conv_fwd_kind(int n, int d, int h, int w, int cin, int cout, bool gnb, bool split = false) {
  using G = FwdTile<1>;
  const int ntiles16 = n * ((d + G::TZ - 1) / G::TZ) * ((h + G::TY - 1) / G::TY) * ((w + G::TX - 1) / G::TX);
  return narrow_bricks(w) && !conv32_takes(ntiles16, cin, cout, gnb, split) ? 3 : 1;
}
// workgroups of a conv2b launch: one per CU, fewer (a multiple of 8) when the launch has fewer items
static int conv2b_grid(int nitems) { return nitems < 256 ? nitems : 256; }
// The ConvTranspose3d data-gradient specialisation (convt_dgrad32_mfma_kernel, round 6): 32 gradient channels in, 64 out, one
// workgroup per CU with at least one 2 x 4 x 16 brick each; split weights stay with the general kernel (one image in registers)
static bool convt_dgrad32_takes(int ntiles, int k_dy, int m_dx, bool split) {
  return k_dy == 32 && m_dx == 64 && !split && ntiles >= 256 && ::mednet_internal_cu_count() == 256 && tuning_option("convt_dgrad32", 1);
}
static void conv_stats_plan(int n, int d, int h, int w, int cin, int cout, int& rows, int& accum, bool gnb = false, bool split = false) {
  const int tx = conv_fwd_kind(n, d, h, w, cin, cout, gnb, split) == 3 ? FwdTile<3>::TX : FwdTile<1>::TX;
  using G = FwdTile<1>;
  const int tps = ((d + G::TZ - 1) / G::TZ) * ((h + G::TY - 1) / G::TY) * ((w + tx - 1) / tx);
  const int ncb = (cout + 31) / 32, ntiles = n * tps;
  if (tx == G::TX && conv32_takes(ntiles, cin, cout, gnb, split)) {  // always accumulating: one row per wave of the 256 workgroups
    accum = 1;
    rows = 256 * 4;
    return;
  }
  if (tx == G::TX && conv2b_takes(ntiles, cin, cout, split)) {
    // 256 workgroups stepping by 256 items: a workgroup keeps its block pair (split weights: its block) when the pair count
    // divides 32, and without padding items every workgroup starts on a valid brick: one row per wave of the workgroups that
    // share a pair, else a row per wave and brick
    const int ncbp = split ? cout / 32 : cout / 64;
    const int grid = conv2b_grid(((ntiles + 7) / 8) * 8 * ncbp);
    accum = tuning_option("conv_stats_accum", 1) && ntiles % 8 == 0 && (grid / 8) % ncbp == 0;
    rows = accum ? (grid / 8 / ncbp) * 8 * 4 : 4 * tps;
    return;
  }
  const int nitems = ((ntiles + 7) / 8) * 8 * ncb;
  accum = tuning_option("conv_persist", 1) && tuning_option("conv_stats_accum", 1) && nitems >= 1024 && ntiles % 8 == 0 &&
          64 % ncb == 0;
  rows = accum ? (512 / ncb) * 4 : 4 * tps;
}

// What a launch of the forward / data-gradient family is made of, for audits without a device (mednet_conv3d_stats_plan,
// tests/test_plan_audit.py): the launcher fills it from the SAME code path that launches and returns before the launch.
struct FwdPlanProbe {
  int kind;  // 2: general kernel, one statistics row per wave and brick; 3: general kernel, accumulate mode; 4: conv32_mfma_kernel;
             // 5 / 6: conv2b_mfma_kernel (`ncb` = PAIRS of channel blocks), row per wave and brick / accumulate mode
  int grid, nitems, ncb, ntiles, tiles_per_sample, accum, rows, xcd_chunk, zslab, tiles_x, tiles_y, tiles_z;
};

struct GnbSpec {  // fused first pass of a GroupNorm backward (see FwdArgs::gnb_y); partial goes through gn_partial
  const void* y = nullptr;
  const float* coef = nullptr;
  int act = MEDNET_ACT_NONE;
  const void* z = nullptr;  // residual-layer form (FwdArgs::gnb_z)
};

template <int KIND>
static int launch_fwd(const void* x, const void* sec, void* y, int n, int od, int oh, int ow, int id, int ih, int iw,
                      int cin, int cout, float* gn_partial, hipStream_t s, int act = MEDNET_ACT_NONE,
                      const void* add = nullptr, GnbSpec gnb = GnbSpec(), FwdPlanProbe* probe = nullptr, size_t lo_delta = 0) {
  using G = FwdTile<KIND>;
  constexpr int STRIDE = G::STRIDE;
  constexpr int HZ = STRIDE * (G::TZ - 1) + 3, HY = STRIDE * (G::TY - 1) + 3, HX = STRIDE * (G::TX - 1) + 3;
  constexpr size_t lds = ((size_t)2 * HZ * HY * HX + 27 * 2 * 32) * 16;
  static_assert(lds <= 80 * 1024, "two workgroups must fit one CU");
  FwdArgs a;
  a.gn_partial = gn_partial;
  a.act = act;
  a.add = (const elt*)add;
  a.gnb_y = (const elt*)gnb.y;
  a.gnb_coef = gnb.coef;
  a.gnb_act = gnb.act;
  a.gnb_z = (const elt*)gnb.z;
  const bool use_gnb = gnb.y != nullptr;
  MEDNET_REQUIRE(!use_gnb || (gn_partial && (gnb.z ? STRIDE == 2 : (STRIDE == 1 && gnb.coef))), MEDNET_E_UNSUPPORTED,
                 "conv_mfma: fused GroupNorm-backward sums need a partial buffer and the forward affine (stride 1) or the block "
                 "output (stride 2)");
  a.x = (const elt*)x;
  a.wpk = (const elt*)sec;
  a.y = (elt*)y;
  a.n = n; a.od = od; a.oh = oh; a.ow = ow; a.id = id; a.ih = ih; a.iw = iw;
  a.cin = cin; a.cout = cout;
  a.tiles_z = (od + G::TZ - 1) / G::TZ;
  a.tiles_y = (oh + G::TY - 1) / G::TY;
  a.tiles_x = (ow + G::TX - 1) / G::TX;
  a.ntiles = n * a.tiles_z * a.tiles_y * a.tiles_x;
  const bool split = lo_delta != 0;  // split weights: the low image lies lo_delta bytes behind `sec` (FwdArgs::lo_delta)
  MEDNET_REQUIRE(lo_delta < 0xFFFFFFFFull, MEDNET_E_UNSUPPORTED, "conv_mfma: weight images too large for split weights");
  a.lo_delta = (unsigned)lo_delta;
  a.nkc_in = cin / 16;
  a.nkc = split ? 2 * a.nkc_in : a.nkc_in;  // (the general kernel: the low image as nkc_in more chunks)
  a.ncb = (cout + 31) / 32;
  auto rcp = [](int d) { return d == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
  a.rcp_tiles_x = rcp(a.tiles_x); a.rcp_tiles_y = rcp(a.tiles_y); a.rcp_tiles_z = rcp(a.tiles_z); a.rcp_ncb = rcp(a.ncb);
  MEDNET_REQUIRE((double)a.ntiles * a.ncb * 8.0 * 1024.0 < 4294967296.0, MEDNET_E_UNSUPPORTED, "conv_mfma: grid too large");
  a.bytes_x = (unsigned)((size_t)id * ih * iw * cin * 2);
  MEDNET_REQUIRE((double)od * oh * ow * cout * 2.0 < 4294960000.0, MEDNET_E_UNSUPPORTED, "conv_mfma: one output sample must stay below 4 GB");
  a.bytes_y = (unsigned)((size_t)od * oh * ow * cout * 2);
  a.nitems = ((a.ntiles + 7) / 8) * 8 * a.ncb;
  // persistent form: 2 workgroups per CU stream through the items (option conv_persist=0: one workgroup per item)
  unsigned grid = (unsigned)a.nitems;
  if (tuning_option("conv_persist", 1) && grid > 512u) grid = 512u;
  a.stats_accum = 0;
  a.stats_rows = 4 * a.tiles_z * a.tiles_y * a.tiles_x;
  if (gn_partial && STRIDE == 1) conv_stats_plan(n, od, oh, ow, cin, cout, a.stats_rows, a.stats_accum, use_gnb, split);
  a.xcd_chunk = 0;
  if constexpr (KIND == 2) {
    if (convt_dgrad32_takes(a.ntiles, cin, cout, split) && !add && act == MEDNET_ACT_NONE) {
      Ct32Args c;
      c.dy = a.x; c.wpk = a.wpk; c.dx = a.y;
      c.gn_y = a.gnb_y; c.gn_z = a.gnb_z; c.gn_partial = a.gn_partial; c.gn_act = a.gnb_act;
      c.n = n; c.d = od; c.h = oh; c.w = ow;
      c.tiles_z = a.tiles_z; c.tiles_y = a.tiles_y; c.tiles_x = a.tiles_x; c.ntiles = a.ntiles;
      c.rcp_tiles_x = a.rcp_tiles_x; c.rcp_tiles_y = a.rcp_tiles_y; c.rcp_tiles_z = a.rcp_tiles_z;
      c.bytes_dy = a.bytes_x; c.bytes_dx = a.bytes_y;
      const int rows32 = 2 * 256;  // one row per (workgroup, z-plane of its bricks) and sample: see convt_dgrad_gn_rows
      if (probe) {
        *probe = FwdPlanProbe{7, 256, a.ntiles, 2, a.ntiles, a.tiles_z * a.tiles_y * a.tiles_x, 1, rows32,
                              a.ntiles % 8 == 0 ? a.ntiles / 8 : 0, 0, a.tiles_x, a.tiles_y, a.tiles_z};
        return MEDNET_OK;
      }
      static bool attr_ct32[2] = {false, false};
      auto go = [&](auto kernel) -> int {
        if (!attr_ct32[use_gnb]) {
          if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT32_LDS) != hipSuccess)
            return fail(MEDNET_E_HIP, "convt_dgrad32_mfma: cannot raise dynamic LDS to %zu", CT32_LDS);
          attr_ct32[use_gnb] = true;
        }
        hipLaunchKernelGGL(kernel, dim3(256), dim3(256), CT32_LDS, s, c);
        return MEDNET_OK;
      };
      const int rc = use_gnb ? go(convt_dgrad32_mfma_kernel<true>) : go(convt_dgrad32_mfma_kernel<false>);
      if (rc) return rc;
      return check_launch("convt_dgrad32_mfma");
    }
  }
  if constexpr (KIND == 1) {
    if (conv32_takes(a.ntiles, cin, cout, use_gnb, split)) {
      constexpr size_t lds32 = (size_t)2 * 4 * (HZ * HY * HX + 4) * 16 + 4 * 4096 + 256 * 16;
      static_assert(lds32 <= 160 * 1024, "two bricks of whole rows + the waves' epilogue areas + spare slots");
      a.xcd_chunk = a.ntiles % 8 == 0 ? a.ntiles / 8 : 0;
      a.zslab = 0;
      a.rcp_zslab = 0;
      if ((n * a.tiles_z) % 8 == 0 && a.tiles_z % (n * a.tiles_z / 8) == 0 && tuning_option("conv32_zslab", 1)) {
        a.zslab = n * a.tiles_z / 8;  // (a slab lies inside ONE sample: the statistics rows of a workgroup need that)
        a.rcp_zslab = rcp(a.zslab);
      }
      const int variant = use_gnb ? (C32_GNB | (add ? C32_ADD : 0))
                                  : ((add ? C32_ADD : 0) | (gn_partial ? C32_STATS : 0) | (act != MEDNET_ACT_NONE ? C32_ACT : 0));
      MEDNET_REQUIRE(!use_gnb || act == MEDNET_ACT_NONE, MEDNET_E_UNSUPPORTED, "conv32_mfma: no activation in the data-gradient form");
      if (probe) {
        *probe = FwdPlanProbe{4, 256, a.ntiles, a.ncb, a.ntiles, a.tiles_z * a.tiles_y * a.tiles_x, a.stats_accum, a.stats_rows,
                              a.xcd_chunk, a.zslab, a.tiles_x, a.tiles_y, a.tiles_z};
        return MEDNET_OK;
      }
      static bool attr32[16] = {};
      auto go = [&](auto kernel) -> int {
        if (!attr32[variant]) {
          if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32) != hipSuccess)
            return fail(MEDNET_E_HIP, "conv32_mfma: cannot raise dynamic LDS to %zu", lds32);
          attr32[variant] = true;
        }
        hipLaunchKernelGGL(kernel, dim3(256), dim3(256), lds32, s, a);
        return MEDNET_OK;
      };
      int rc32 = MEDNET_OK;
      switch (variant) {
        case 0: rc32 = go(conv32_mfma_kernel<0>); break;
        case C32_ADD: rc32 = go(conv32_mfma_kernel<C32_ADD>); break;
        case C32_STATS: rc32 = go(conv32_mfma_kernel<C32_STATS>); break;
        case C32_STATS | C32_ADD: rc32 = go(conv32_mfma_kernel<C32_STATS | C32_ADD>); break;
        case C32_ACT: rc32 = go(conv32_mfma_kernel<C32_ACT>); break;
        case C32_ACT | C32_ADD: rc32 = go(conv32_mfma_kernel<C32_ACT | C32_ADD>); break;
        case C32_ACT | C32_STATS: rc32 = go(conv32_mfma_kernel<C32_ACT | C32_STATS>); break;
        case C32_ACT | C32_STATS | C32_ADD: rc32 = go(conv32_mfma_kernel<C32_ACT | C32_STATS | C32_ADD>); break;
        case C32_GNB: rc32 = go(conv32_mfma_kernel<C32_GNB>); break;
        default: rc32 = go(conv32_mfma_kernel<C32_GNB | C32_ADD>); break;
      }
      if (rc32) return rc32;
      return check_launch("conv32_mfma");
    }
    if (conv2b_takes(a.ntiles, cin, cout, split)) {
      FwdArgs b2 = a;
      b2.ncb = split ? cout / 32 : cout / 64;  // PAIRS of channel blocks; split weights: blocks (the pair is (high, low))
      b2.rcp_ncb = rcp(b2.ncb);
      b2.nitems = ((a.ntiles + 7) / 8) * 8 * b2.ncb;
      b2.nkc = a.nkc_in;
      b2.xcd_chunk = (a.ntiles % 8 == 0 && tuning_option("conv2b_xcd_walk", 1)) ? a.ntiles / 8 : 0;
      const int grid2 = conv2b_grid(b2.nitems);
      const int variant = (use_gnb ? (C32_GNB | (add ? C32_ADD : 0))
                                   : ((add ? C32_ADD : 0) | (gn_partial ? C32_STATS : 0) | (act != MEDNET_ACT_NONE ? C32_ACT : 0))) |
                          (split ? C32_SPLIT : 0);
      MEDNET_REQUIRE(!use_gnb || act == MEDNET_ACT_NONE, MEDNET_E_UNSUPPORTED, "conv2b_mfma: no activation in the data-gradient form");
      if (probe) {
        *probe = FwdPlanProbe{b2.stats_accum ? 6 : 5, grid2, b2.nitems, b2.ncb, a.ntiles, a.tiles_z * a.tiles_y * a.tiles_x, b2.stats_accum,
                              b2.stats_rows, b2.xcd_chunk, 0, a.tiles_x, a.tiles_y, a.tiles_z};
        return MEDNET_OK;
      }
      static bool attr2b[32] = {};
      auto go2 = [&](auto kernel) -> int {
        if (!attr2b[variant]) {
          if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C2B_LDS) != hipSuccess)
            return fail(MEDNET_E_HIP, "conv2b_mfma: cannot raise dynamic LDS to %zu", C2B_LDS);
          attr2b[variant] = true;
        }
        hipLaunchKernelGGL(kernel, dim3(grid2), dim3(256), C2B_LDS, s, b2);
        return MEDNET_OK;
      };
      int rc2 = MEDNET_OK;
      switch (variant) {
        case C32_SPLIT: rc2 = go2(conv2b_mfma_kernel<C32_SPLIT>); break;
        case C32_SPLIT | C32_ADD: rc2 = go2(conv2b_mfma_kernel<C32_SPLIT | C32_ADD>); break;
        case C32_SPLIT | C32_STATS: rc2 = go2(conv2b_mfma_kernel<C32_SPLIT | C32_STATS>); break;
        case C32_SPLIT | C32_STATS | C32_ADD: rc2 = go2(conv2b_mfma_kernel<C32_SPLIT | C32_STATS | C32_ADD>); break;
        case C32_SPLIT | C32_ACT: rc2 = go2(conv2b_mfma_kernel<C32_SPLIT | C32_ACT>); break;
        case C32_SPLIT | C32_ACT | C32_ADD: rc2 = go2(conv2b_mfma_kernel<C32_SPLIT | C32_ACT | C32_ADD>); break;
        case C32_SPLIT | C32_ACT | C32_STATS: rc2 = go2(conv2b_mfma_kernel<C32_SPLIT | C32_ACT | C32_STATS>); break;
        case C32_SPLIT | C32_ACT | C32_STATS | C32_ADD: rc2 = go2(conv2b_mfma_kernel<C32_SPLIT | C32_ACT | C32_STATS | C32_ADD>); break;
        case C32_SPLIT | C32_GNB: rc2 = go2(conv2b_mfma_kernel<C32_SPLIT | C32_GNB>); break;
        case C32_SPLIT | C32_GNB | C32_ADD: rc2 = go2(conv2b_mfma_kernel<C32_SPLIT | C32_GNB | C32_ADD>); break;
        case 0: rc2 = go2(conv2b_mfma_kernel<0>); break;
        case C32_ADD: rc2 = go2(conv2b_mfma_kernel<C32_ADD>); break;
        case C32_STATS: rc2 = go2(conv2b_mfma_kernel<C32_STATS>); break;
        case C32_STATS | C32_ADD: rc2 = go2(conv2b_mfma_kernel<C32_STATS | C32_ADD>); break;
        case C32_ACT: rc2 = go2(conv2b_mfma_kernel<C32_ACT>); break;
        case C32_ACT | C32_ADD: rc2 = go2(conv2b_mfma_kernel<C32_ACT | C32_ADD>); break;
        case C32_ACT | C32_STATS: rc2 = go2(conv2b_mfma_kernel<C32_ACT | C32_STATS>); break;
        case C32_ACT | C32_STATS | C32_ADD: rc2 = go2(conv2b_mfma_kernel<C32_ACT | C32_STATS | C32_ADD>); break;
        case C32_GNB: rc2 = go2(conv2b_mfma_kernel<C32_GNB>); break;
        default: rc2 = go2(conv2b_mfma_kernel<C32_GNB | C32_ADD>); break;
      }
      if (rc2) return rc2;
      return check_launch("conv2b_mfma");
    }
  }
  if (probe) {
    *probe = FwdPlanProbe{a.stats_accum ? 3 : 2, (int)grid, a.nitems, a.ncb, a.ntiles, a.tiles_z * a.tiles_y * a.tiles_x,
                          a.stats_accum, a.stats_rows, 0, 0, a.tiles_x, a.tiles_y, a.tiles_z};
    return MEDNET_OK;
  }
  static bool attr_set[4] = {false, false, false, false};
  if (!attr_set[KIND]) {
    if (hipFuncSetAttribute((const void*)conv_mfma_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(MEDNET_E_HIP, "conv_mfma: cannot raise dynamic LDS to %zu", lds);
    attr_set[KIND] = true;
  }
  if constexpr (STRIDE == 1) {
    if (use_gnb) {
      constexpr size_t lds_gnb = lds + 16 * 256 * sizeof(float);  // + the parked GroupNorm-backward sums (16 per thread)
      static_assert(lds_gnb <= 80 * 1024, "two workgroups must fit one CU");
      static bool attr_gnb[4] = {false, false, false, false};
      if (!attr_gnb[KIND]) {
        if (hipFuncSetAttribute((const void*)conv_mfma_kernel<KIND, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_gnb) != hipSuccess)
          return fail(MEDNET_E_HIP, "conv_mfma: cannot raise dynamic LDS to %zu", lds_gnb);
        attr_gnb[KIND] = true;
      }
      hipLaunchKernelGGL((conv_mfma_kernel<KIND, true>), dim3(grid), dim3(256), lds_gnb, s, a);
      return check_launch("conv_mfma(gnb)");
    }
  }
  if constexpr (STRIDE == 2) {
    if (use_gnb) {
      static bool attr_gnb2 = false;
      if (!attr_gnb2) {
        if (hipFuncSetAttribute((const void*)conv_mfma_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
          return fail(MEDNET_E_HIP, "conv_mfma: cannot raise dynamic LDS to %zu", lds);
        attr_gnb2 = true;
      }
      hipLaunchKernelGGL((conv_mfma_kernel<2, true>), dim3(grid), dim3(256), lds, s, a);
      return check_launch("conv_mfma(gnb, stride 2)");
    }
  }
  hipLaunchKernelGGL((conv_mfma_kernel<KIND>), dim3(grid), dim3(256), lds, s, a);
  return check_launch("conv_mfma");
}

int launch_conv_mfma(const void* x, const void* packed_section, void* y, int n, int d, int h, int w, int cin, int cout,
                     int x_dtype, int y_dtype, float* gn_partial, hipStream_t s, int act, const void* add, size_t lo_delta) {
  (void)x_dtype;
  (void)y_dtype;
  if (conv_fwd_kind(n, d, h, w, cin, cout, false, lo_delta != 0) == 3)
    return launch_fwd<3>(x, packed_section, y, n, d, h, w, d, h, w, cin, cout, gn_partial, s, act, add, GnbSpec(), nullptr, lo_delta);
  return launch_fwd<1>(x, packed_section, y, n, d, h, w, d, h, w, cin, cout, gn_partial, s, act, add, GnbSpec(), nullptr, lo_delta);
}
// data gradient + the first pass of the GroupNorm backward its output feeds (gn_partial: per-channel {sum du, sum du*y})
int launch_conv_mfma_gnb(const void* dy, const void* packed_section, void* dx, int n, int d, int h, int w, int cin, int cout,
                         const void* add, const void* gn_y, const float* gn_coef, int gn_act, float* gn_partial, hipStream_t s,
                         size_t lo_delta) {
  GnbSpec g;
  g.y = gn_y;
  g.coef = gn_coef;
  g.act = gn_act;
  if (conv_fwd_kind(n, d, h, w, cin, cout, true, lo_delta != 0) == 3)
    return launch_fwd<3>(dy, packed_section, dx, n, d, h, w, d, h, w, cin, cout, gn_partial, s, MEDNET_ACT_NONE, add, g, nullptr, lo_delta);
  return launch_fwd<1>(dy, packed_section, dx, n, d, h, w, d, h, w, cin, cout, gn_partial, s, MEDNET_ACT_NONE, add, g, nullptr, lo_delta);
}
// plan of launch_conv_mfma / launch_conv_mfma_gnb (stride 1) or launch_convt_dgrad_gn_mfma (stride 2; d, h, w = the LOW-resolution
// grid the data gradient writes), with fused sums: out[13] = the fields of FwdPlanProbe in order
int conv_mfma_plan(int n, int d, int h, int w, int cin, int cout, bool gnb, int stride, int* out, bool split) {
  FwdPlanProbe p{};
  const size_t lo = split ? 256 : 0;  // (any non-zero distance: the probe returns before the launch)
  float dummy_partial;  // (never dereferenced: the probe returns before the launch)
  GnbSpec g;
  int rc;
  if (stride == 2) {
    g.y = g.z = &dummy_partial;
    rc = launch_fwd<2>(nullptr, nullptr, nullptr, n, d, h, w, 2 * d, 2 * h, 2 * w, cout, cin, &dummy_partial, nullptr, MEDNET_ACT_NONE,
                       nullptr, g, &p, lo);
  } else {
    if (gnb) {
      g.y = &dummy_partial;
      g.coef = &dummy_partial;
    }
    rc = conv_fwd_kind(n, d, h, w, cin, cout, gnb, split) == 3
             ? launch_fwd<3>(nullptr, nullptr, nullptr, n, d, h, w, d, h, w, cin, cout, &dummy_partial, nullptr, MEDNET_ACT_NONE, nullptr, g, &p, lo)
             : launch_fwd<1>(nullptr, nullptr, nullptr, n, d, h, w, d, h, w, cin, cout, &dummy_partial, nullptr, MEDNET_ACT_NONE, nullptr, g, &p, lo);
  }
  if (rc) return rc;
  const int v[13] = {p.kind, p.grid, p.nitems, p.ncb, p.ntiles, p.tiles_per_sample, p.accum, p.rows, p.xcd_chunk, p.zslab,
                     p.tiles_x, p.tiles_y, p.tiles_z};
  for (int i = 0; i < 13; ++i) out[i] = v[i];
  return MEDNET_OK;
}
int conv_mfma_stats_chunks(int n, int d, int h, int w, int cin, int cout, bool gnb, bool split) {
  int rows, accum;
  conv_stats_plan(n, d, h, w, cin, cout, rows, accum, gnb, split);
  // (A/B knob conv_fuse_gnb_general=0: only the 32 -> 32 specialisation takes the GroupNorm-backward sums in its epilogue; the
  //  general kernel's data gradients leave them to the stand-alone pass)
  if (gnb && !tuning_option("conv_fuse_gnb_general", 1)) {
    using G = FwdTile<1>;
    const int ntiles16 = n * ((d + G::TZ - 1) / G::TZ) * ((h + G::TY - 1) / G::TY) * ((w + G::TX - 1) / G::TX);
    if (!conv32_takes(ntiles16, cin, cout, true, split)) return 0;
  }
  return rows;
}

int launch_convt_dgrad_mfma(const void* dy, const void* packed_section, void* dx, int n, int d, int h, int w, int cin,
                            int cout, hipStream_t s, size_t lo_delta) {
  // dx (d,h,w; Cin channels) <- dy (2d,2h,2w; Cout channels)
  return launch_fwd<2>(dy, packed_section, dx, n, d, h, w, 2 * d, 2 * h, 2 * w, cout, cin, nullptr, s, MEDNET_ACT_NONE, nullptr, GnbSpec(),
                       nullptr, lo_delta);
}
// ... + the first pass of the GroupNorm-3 backward of the ExtResNetBlock whose output the ConvTranspose3d upsamples
// (model.py:202-207): one partial row per wave and brick, gn_partial[n][rows][Cin][2]
int convt_dgrad_gn_rows(int n, int d, int h, int w, int cin, int cout, bool split) {
  using G = FwdTile<2>;
  const int tps = ((d + G::TZ - 1) / G::TZ) * ((h + G::TY - 1) / G::TY) * ((w + G::TX - 1) / G::TX);
  // (cin, cout are the ConvTranspose3d's: its data gradient reads cout channels and writes cin)
  if (convt_dgrad32_takes(n * tps, cout, cin, split)) return 2 * 256;  // convt_dgrad32_mfma_kernel: 2 rows per workgroup and sample
  return 4 * tps;
}
int launch_convt_dgrad_gn_mfma(const void* dy, const void* packed_section, void* dx, const void* gn_y, const void* gn_z, int gn_act,
                               float* gn_partial, int n, int d, int h, int w, int cin, int cout, hipStream_t s, size_t lo_delta) {
  GnbSpec g;
  g.y = gn_y;
  g.z = gn_z;
  g.act = gn_act;
  return launch_fwd<2>(dy, packed_section, dx, n, d, h, w, 2 * d, 2 * h, 2 * w, cout, cin, gn_partial, s, MEDNET_ACT_NONE, nullptr, g,
                       nullptr, lo_delta);
}

// ================================================================================================== weight gradient
// ---- weight gradient, second generation (stride 1): 8 waves, one workgroup per CU, 4x8x16 bricks ---------------------
// Measured on the first kernel: its time is the per-brick staging (2.8x halo at 2x8x16, no overlap inside the
// workgroup), not the MFMAs.  Here the brick is twice as deep (halo 2.1x, half the fixed cost per MFMA), the 32 k-steps
// of a brick are split between two groups of 4 waves (each wave still owns 7 taps = 112 accumulator registers; the two
// groups' partial sums are separate slabs for the reduce kernel), and with 512 threads the next brick fits in 13 staging
// registers per lane, fetched with buffer loads (hardware zero padding) while the current brick is on the matrix cores.
struct Wg2Args {
  const elt* A;
  const elt* B;
  float* part;  // [wg][27][32][32]
  int n, d, h, w, ka, kb;
  int tiles_z, tiles_y, tiles_x, ntiles;
  int nab, nbb, splits;
  unsigned rcp_tiles_x, rcp_tiles_y, rcp_tiles_z;  // ceil(2^32 / d), see fastdiv
  unsigned bytesA, bytesB;
};

// TX = 16, or 8 on narrow volumes (see FwdTile<3>): a k-step's 16 voxels are then two x-rows of 8.
template <int TX>
__global__ __launch_bounds__(512, 2) void wgrad_mfma2_kernel(Wg2Args a) {
  constexpr int TZ = 4, TY = 8, HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
  static_assert(TX == 16 || TX == 8, "brick widths");
  constexpr int NA = TZ * TY * TX, NB = HZ * HY * HX;
  constexpr int A_ROUNDS = NA * 4 / 512, B_ROUNDS = (NB * 4 + 511) / 512;
  constexpr int KSTEPS = NA / 16;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  elt* A_lds = reinterpret_cast<elt*>(smem);
  elt* B_lds = reinterpret_cast<elt*>(smem) + NA * 32;

  const int pair = blockIdx.x / a.splits, split = blockIdx.x % a.splits;
  const int ab = pair / a.nbb, bb = pair % a.nbb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
  const int kgrp = wvu >> 2, tw = wvu & 3;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, hk = lane >> 5;
  const int coloff = (16 * (g & 1) + 4 * p) * 2;

  constexpr unsigned OOB = 0xFFFFFF00u;
  u32x4 regA[A_ROUNDS], regB[B_ROUNDS];
  // Staging of one brick = A_ROUNDS + B_ROUNDS (4 + 9) buffer loads per thread.  `Next` holds the wave-uniform part
  // (brick origin, per-sample resources); fetch_one(j) issues load j.  Branch-free: one unsigned compare per axis, and a
  // piece outside the volume / past the channel count / past the brick list gets an out-of-range offset (hardware zero).
  struct Next {
    int tz0, ty0, tx0;
    unsigned kill;
    __amdgpu_buffer_rsrc_t rA, rB;
  };
  auto plan_next = [&](int tile, bool valid) {
    Next nx;
    int tt = valid ? tile : 0;
    int qd = fastdiv(tt, a.tiles_x, a.rcp_tiles_x);
    nx.tx0 = (tt - qd * a.tiles_x) * TX;
    tt = qd;
    qd = fastdiv(tt, a.tiles_y, a.rcp_tiles_y);
    nx.ty0 = (tt - qd * a.tiles_y) * TY;
    tt = qd;
    qd = fastdiv(tt, a.tiles_z, a.rcp_tiles_z);
    nx.tz0 = (tt - qd * a.tiles_z) * TZ;
    const size_t svox = (size_t)qd * a.d * a.h * a.w;  // one resource per sample: 32-bit offsets span a single sample
    nx.rA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + svox * a.ka), 0, a.bytesA, 0x00020000);
    nx.rB = __builtin_amdgcn_make_buffer_rsrc((void*)(a.B + svox * a.kb), 0, a.bytesB, 0x00020000);
    nx.kill = valid ? 0u : OOB;
    return nx;
  };
  auto fetch_one = [&](int j, const Next& nx) {
    if (j < A_ROUNDS) {
      const int c = j * 512 + tid;
      const int v = c >> 2, part = c & 3;
      const int gz = nx.tz0 + v / (TX * TY), gy = nx.ty0 + (v / TX) % TY, gx = nx.tx0 + v % TX;
      const bool in_vol = (gz < a.d) & (gy < a.h) & (gx < a.w) & (ab * 32 + part * 8 < a.ka);
      const unsigned off = ((unsigned)((gz * a.h + gy) * a.w + gx) * (unsigned)a.ka + ab * 32 + part * 8) * 2u;
      regA[j] = __builtin_amdgcn_raw_buffer_load_b128(nx.rA, (in_vol ? off : OOB) | nx.kill, 0, 0);
    } else {
      const int it = j - A_ROUNDS;
      const int c = it * 512 + tid;
      const int v = c >> 2, part = c & 3;
      const int gz = nx.tz0 - 1 + v / (HX * HY), gy = nx.ty0 - 1 + (v / HX) % HY, gx = nx.tx0 - 1 + v % HX;
      const bool in_vol = (c < NB * 4) & ((unsigned)gz < (unsigned)a.d) & ((unsigned)gy < (unsigned)a.h) &
                          ((unsigned)gx < (unsigned)a.w) & (bb * 32 + part * 8 < a.kb);
      const unsigned off = ((unsigned)((gz * a.h + gy) * a.w + gx) * (unsigned)a.kb + bb * 32 + part * 8) * 2u;
      regB[it] = __builtin_amdgcn_raw_buffer_load_b128(nx.rB, (in_vol ? off : OOB) | nx.kill, 0, 0);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < A_ROUNDS; ++it) *reinterpret_cast<u32x4*>(A_lds + (it * 512 + tid) * 8) = regA[it];
#pragma unroll
    for (int it = 0; it < B_ROUNDS; ++it) {
      const int c = it * 512 + tid;
      if (c < NB * 4) *reinterpret_cast<u32x4*>(B_lds + c * 8) = regB[it];
    }
  };

  f32x16 acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const char* Ab = reinterpret_cast<const char*>(A_lds) + coloff;
  const char* Bb = reinterpret_cast<const char*>(B_lds) + coloff;
  int toff[7];  // tap offsets live in SGPRs; the wave with 6 taps recomputes tap 26 in its 7th slot (discarded)
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = tw + 4 * i < 27 ? tw + 4 * i : 26;
    toff[i] = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * 64;
  }

  int tile = split;
  const int t_step = a.splits, t_end = a.ntiles;
  if (tile < t_end) {
    const Next first = plan_next(tile, true);
#pragma unroll
    for (int j = 0; j < A_ROUNDS + B_ROUNDS; ++j) fetch_one(j, first);
  }
  static_assert(A_ROUNDS + B_ROUNDS <= KSTEPS / 2, "one staging load per k-step");
  for (; tile < t_end; tile += t_step) {
    __syncthreads();  // previous brick fully consumed
    commit();
    __syncthreads();
    // the next brick flies while this one is on the matrix cores: its 13 loads are dealt out one per k-step (a burst
    // blocks the wave's instruction issue behind the CU's texture-address path for thousands of cycles)
    const Next nx = plan_next(tile + t_step, tile + t_step < t_end);
#pragma unroll
    for (int k2 = 0; k2 < KSTEPS / 2; ++k2) {
      const int ks = 2 * k2 + kgrp;  // the two wave groups interleave the brick's k-steps
      const eltx8 fa = tr_operand(Ab + (ks * 16 + 8 * hk + q) * 64, 4 * 64);  // (dy brick: voxels in x, y, z order, no halo)
      // x: the k-step's voxels in the halo brick -- one x-row of 16 (lanes hk = 1 hold its second half), or the x-rows 2 ks and
      // 2 ks + 1 of 8 (TY is even: both lie in the same z-plane)
      const int krow = TX == 16 ? ks : 2 * ks;
      const char* brow = Bb + (((krow / TY) * HY + krow % TY) * HX + (TX == 16 ? 8 * hk : hk * HX) + q) * 64;
      eltx8 fb[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) fb[i] = tr_operand(brow + toff[i], 4 * 64);
      if (k2 < A_ROUNDS + B_ROUNDS) fetch_one(k2, nx);
#pragma unroll
      for (int i = 0; i < 7; ++i) acc[i] = MEDNET_MFMA_32x32x16(fa, fb[i], acc[i], 0, 0, 0);
    }
  }
  // The two k-groups merge their accumulators through LDS (k-group 1 parks them, k-group 0 adds in a fixed order), so a
  // workgroup writes ONE partial slab: half the partial traffic of this kernel and of the reduce kernel.  The LDS holds
  // 4 tap slots of the 4 tap-waves at a time (64 KB), so it takes two rounds.
  float* mlds = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i0 = 0; i0 < 7; i0 += 4) {
    __syncthreads();  // the bricks' operands (first round) / the previous round's sums are no longer needed
    if (kgrp == 1) {
#pragma unroll
      for (int i = i0; i < (i0 + 4 < 7 ? i0 + 4 : 7); ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) mlds[((tw * 4 + (i - i0)) * 16 + j) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    if (kgrp == 0) {
#pragma unroll
      for (int i = i0; i < (i0 + 4 < 7 ? i0 + 4 : 7); ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] += mlds[((tw * 4 + (i - i0)) * 16 + j) * 64 + lane];
    }
  }
  if (kgrp == 0) {
    float* out = a.part + (size_t)blockIdx.x * 27 * 1024;
    const int col = lane & 31;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int tap = tw + 4 * i;
      if (tap < 27) {
#pragma unroll
        for (int j = 0; j < 16; ++j) out[((size_t)tap * 32 + (j & 3) + 8 * (j >> 2) + 4 * hk) * 32 + col] = acc[i][j];
      }
    }
  }
}

static int wgrad2_tx(int w) { return narrow_bricks(w) ? 8 : 16; }
static void wgrad2_plan(int n, int d, int h, int w, int ka, int kb, int workgroups, Wg2Args& a) {
  const int tx = wgrad2_tx(w);
  a.tiles_z = (d + 3) / 4;
  a.tiles_y = (h + 7) / 8;
  a.tiles_x = (w + tx - 1) / tx;
  a.ntiles = n * a.tiles_z * a.tiles_y * a.tiles_x;
  auto rcp = [](int d) { return d == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
  a.rcp_tiles_x = rcp(a.tiles_x); a.rcp_tiles_y = rcp(a.tiles_y); a.rcp_tiles_z = rcp(a.tiles_z);
  a.nab = (ka + 31) / 32;  // a 16-channel operand is zero-padded to a 32-wide block by the buffer loads
  a.nbb = (kb + 31) / 32;
  const int pairs = a.nab * a.nbb;
  // workgroups per launch (the caller's `workgroups` argument, 0 = one per CU: the launch has the chip to itself); the trainer
  // asks for HALF of them when the weight gradients run on their own stream beside the main one (train.use_side_stream, DESIGN
  // 12.4): a workgroup takes a CU's whole register file, so 256 of them lock every kernel of the main stream out until they retire
  const int target = workgroups > 0 ? workgroups : 256;
  int splits = (target + pairs - 1) / pairs;
  if (splits > a.ntiles) splits = a.ntiles;
  if (splits < 1) splits = 1;
  a.splits = splits;
}
static size_t wgrad2_ws_bytes(int n, int d, int h, int w, int cin, int cout, int workgroups) {
  Wg2Args a;
  wgrad2_plan(n, d, h, w, cout, cin, workgroups, a);
  return (size_t)a.nab * a.nbb * a.splits * 27 * 1024 * sizeof(float);
}

// dw[(a*KB + b)*27 + tap] = sum_split part[(pair*splits + split)][tap][a%32][b%32]
// (few slabs per output -- deep layers, where the channel-block pairs alone fill the chip.  For one row a of a pair the 32 x 27
//  outputs are CONTIGUOUS in dw: a workgroup takes 8 rows of a pair, sums the slabs with coalesced reads (thread = (row, b), the
//  27 taps in turn), turns the tile to [row][b][tap] in LDS and writes each row's run of 864 floats in order.  The element-per-
//  thread form it replaces wrote 4-byte pieces 108 bytes apart: 136 us for the 113 MB of config 5's 1024 -> 1024 layers, 0.65 ms
//  of its step.)
__global__ __launch_bounds__(256) void wgrad_mfma_reduce_kernel_few(const float* __restrict__ part, float* __restrict__ dw, int ka, int kb, int nbb,
                                               int splits) {
  __shared__ float tile[8][32 * 27 + 1];
  const int pair = (int)blockIdx.x >> 2, a0 = ((int)blockIdx.x & 3) * 8;
  const int ab = pair / nbb, bb = pair % nbb;
  const int tr = threadIdx.x >> 5, tb = threadIdx.x & 31;
  const float* src = part + ((size_t)pair * splits) * 27 * 1024 + (size_t)(a0 + tr) * 32 + tb;
#pragma unroll 3
  for (int tap = 0; tap < 27; ++tap) {
    float s0 = 0.f;
    for (int k = 0; k < splits; ++k) s0 += src[((size_t)k * 27 + tap) * 1024];
    tile[tr][tb * 27 + tap] = s0;
  }
  __syncthreads();
  const int nb = kb - bb * 32 < 32 ? kb - bb * 32 : 32;  // valid columns of this block: the row's run is nb * 27 floats
  for (int r = 0; r < 8; ++r) {
    const int arow = ab * 32 + a0 + r;
    if (arow >= ka) break;
    float* dst = dw + ((size_t)arow * kb + bb * 32) * 27;
    for (int e = threadIdx.x; e < nb * 27; e += 256) dst[e] = tile[r][e];
  }
}
__global__ __launch_bounds__(256) void wgrad_mfma_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int ka, int kb,
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

// ---- weight gradient, fourth generation: ONE wave per SIMD, half a CU's registers, z-ring through LDS-DMA -----------------
// wgrad_mfma2_kernel takes a CU whole (8 waves x 244 registers), so nothing of the main stream -- not even a bandwidth-bound
// GroupNorm pass that would use none of the matrix pipe -- runs on a CU while a weight gradient is resident.
// This kernel computes the same sums with 4 waves (one per SIMD) of < 256 registers: the other half of every SIMD's register
// file, 28 wave slots and 60 KB of LDS stay free for kernels of another stream.
//   * a wave owns 7 of the 27 taps (7 x 16 accumulators) and EVERY k-step of its workgroup (the lean variant of round 3 split
//     the taps over 8 waves, 3-4 each, and became LDS-read-bound: 1.25 transposing reads per MFMA; here 16 reads per 7 MFMAs);
//   * work items are z-COLUMNS: an 8 x 16 (y, x) tile walked plane by plane.  One "tick" = one plane of dy (8 k-steps of 16
//     voxels) against the three planes z-1, z, z+1 of x: the halo in z is never re-fetched, a tick needs ONE new plane of each
//     operand (8 KB + 11.25 KB for 56 MFMAs per wave; the 4x8x16 brick of wgrad_mfma2 fetches 101 KB per 224);
//   * the planes arrive by LDS-DMA (buffer_load ... lds, 1 KB per wave instruction, no staging registers, no commit phase) into
//     rings of RB / RA slots, DEPTH ticks ahead of their use; one s_barrier per tick, one counted s_waitcnt vmcnt in front of it;
//   * the operands of k-step s+1 are read (ds_read_b64_tr_b16) while the MFMAs of k-step s run, across the tick boundary too:
//     the barrier that publishes tick t+1's planes sits in front of the LAST k-step of tick t.
// An item = (sample, z-slab of ZS planes, y tile, x tile) costs ZS + 2 ticks (the slab's ZS + 2 planes of x); the two extra
// ticks issue loads only.  Per-workgroup partial slabs and the fixed-order reduce are those of wgrad_mfma2_kernel.
struct Wg4Args {
  const elt* A;  // dy (ka = Cout channels)
  const elt* B;  // x  (kb = Cin channels)
  float* part;   // [wg][27][32][32]
  int n, d, h, w, ka, kb;
  int tiles_y, tiles_x, zslabs, zs, nitems;
  int nab, nbb, splits, xcd_remap;
  unsigned rcp_tiles_x, rcp_tiles_y, rcp_zslabs;
  unsigned bytesA, bytesB;  // one sample
};

#ifndef MEDNET_WG4_DEPTH
#define MEDNET_WG4_DEPTH 3
#endif
constexpr int WG4_DEPTH = MEDNET_WG4_DEPTH;             // ticks between a plane's LDS-DMA and its first use
constexpr int WG4_RB = WG4_DEPTH + 3, WG4_RA = WG4_DEPTH + 1;
constexpr int WG4_BSLOT = 12 * 1024, WG4_ASLOT = 8 * 1024;  // 10 x 18 voxel rows = 11.25 DMA pieces, padded to 12; 8 x 16 = 8
constexpr size_t WG4_LDS = (size_t)WG4_RB * WG4_BSLOT + (size_t)WG4_RA * WG4_ASLOT;

__global__ __launch_bounds__(256, 2) void wgrad_mfma4_kernel(Wg4Args a) {
  constexpr int TY = 8, TX = 16, HX = 18;
  constexpr int D = WG4_DEPTH, RB = WG4_RB, RA = WG4_RA, BSLOT = WG4_BSLOT, ASLOT = WG4_ASLOT;
  constexpr unsigned OOB = 0xFFFFFF00u;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [RB slots of x planes][RA slots of dy planes]

  const int pair = blockIdx.x / a.splits, split = blockIdx.x % a.splits;
  const int ab = pair / a.nbb, bb = pair % a.nbb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int tw = __builtin_amdgcn_readfirstlane(tid >> 6);  // tap wave (provably wave-uniform)
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, hk = lane >> 5;
  const int lanepart = (8 * hk + q) * 64 + (16 * (g & 1) + 4 * p) * 2;  // see tr_operand / wgrad_mfma2_kernel

  f32x16 acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  int tap_dz[7], tap_off[7];  // wave-uniform (SGPRs): tap = tw + 4 i; the wave with 6 taps recomputes tap 26 (discarded)
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = tw + 4 * i < 27 ? tw + 4 * i : 26;
    tap_dz[i] = tap / 9;
    tap_off[i] = (((tap / 3) % 3) * HX + tap % 3) * 64;
  }

  // ---- this workgroup's items: split, split + splits, ... (with xcd_remap the 32 workgroups of an XCD -- ids 8 apart -- work
  //      on 32 NEIGHBOURING columns at a time, whose x / y halos then meet in that XCD's L2)
  const int first = a.xcd_remap ? (split & 7) * (a.splits >> 3) + (split >> 3) : split;
  const int my_items = first < a.nitems ? (a.nitems - first + a.splits - 1) / a.splits : 0;
  const int TPI = a.zs + 2;  // ticks per item
  const int load_ticks = my_items * TPI, total_ticks = my_items > 0 ? load_ticks + D : 0;

  // load stage state
  int li = 0, lt = 0;                      // item counter, tick inside the item
  int it_z0 = 0, it_y0 = 0;
  const elt* it_A = a.A;
  const elt* it_B = a.B;
  unsigned voffA = OOB, voffB[3] = {OOB, OOB, OOB};
  auto open_item = [&](int k) {  // decode item `first + k * splits`, make the per-lane parts of its plane loads
    int t = first + k * a.splits;
    int qd = fastdiv(t, a.tiles_x, a.rcp_tiles_x);
    const int x0 = (t - qd * a.tiles_x) * TX;
    t = qd;
    qd = fastdiv(t, a.tiles_y, a.rcp_tiles_y);
    it_y0 = (t - qd * a.tiles_y) * TY;
    t = qd;
    qd = fastdiv(t, a.zslabs, a.rcp_zslabs);
    it_z0 = (t - qd * a.zslabs) * a.zs;
    const size_t svox = (size_t)qd * a.d * a.h * a.w;  // one buffer resource per sample: 32-bit offsets span a single sample
    it_A = a.A + svox * a.ka;
    it_B = a.B + svox * a.kb;
    {  // A: row r of the plane is one DMA: lane = (voxel x 4 + 16-byte part); the row's offset is the load's scalar offset
      const int gx = x0 + (lane >> 2), part = lane & 3;
      const bool ok = (gx < a.w) & (ab * 32 + part * 8 < a.ka);
      voffA = ok ? ((unsigned)gx * (unsigned)a.ka + ab * 32 + part * 8) * 2u : OOB;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // B: pieces 64 k .. 64 k + 63 of the plane's 720 (10 rows x 18 voxels x 4), k = tw + 4 j
      const int pc = (tw + 4 * j) * 64 + lane, v = pc >> 2, part = pc & 3;
      const int row = v / HX, hx = v - row * HX;
      const int gy = it_y0 - 1 + row, gx = x0 - 1 + hx;
      const bool ok = (pc < 720) & ((unsigned)gy < (unsigned)a.h) & ((unsigned)gx < (unsigned)a.w) & (bb * 32 + part * 8 < a.kb);
      voffB[j] = ok ? ((unsigned)(gy * a.w + gx) * (unsigned)a.kb + bb * 32 + part * 8) * 2u : OOB;
    }
  };
  // DMA j (0..4) of the load tick: 0..2 pieces tw, tw + 4, tw + 8 of x's plane z0 + lt - 1; 3, 4 rows tw, tw + 4 of dy's plane
  // z0 + lt - 2.  Planes outside the slab / the volume (and every load past the last item) get a resource of 0 bytes: the
  // hardware returns zeros, and every wave issues exactly 5 DMAs per tick whatever happens (the vmcnt arithmetic relies on it).
  int sBw = 0, sAw = 0;  // ring slots the load tick writes
  // One LDS-DMA (1 KB: lane l's 16 bytes land at lds_byte + 16 l).  Hand-written because hipcc (ROCm 7.2) orders EVERY later LDS
  // read of the kernel behind an LDS-DMA it knows about with s_waitcnt vmcnt(0) -- it cannot tell that the ring slot being filled
  // is not the one being read -- which would drain the DEPTH ticks of loads in flight at every k-step.  The kernel keeps its own
  // vmcnt arithmetic instead (one counted wait per tick).  M0 = the LDS byte address, written in the same statement that uses it.
  const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto lds_dma = [&](const elt* base, unsigned num_records, unsigned voff, unsigned soff, unsigned lds_byte) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const unsigned long long ba = (unsigned long long)base;
    u32x4 rs;
    rs[0] = (unsigned)ba;
    rs[1] = (unsigned)(ba >> 32) & 0xFFFFu;
    rs[2] = num_records;
    rs[3] = 0x00020000u;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(smem_base + lds_byte), "v"(voff), "s"(rs), "s"(soff)
                 : "memory");
  };
  auto dma = [&](int j, bool live) {
    if (j < 3) {
      const int z = it_z0 + lt - 1;
      const bool ok = live & ((unsigned)z < (unsigned)a.d);
      const unsigned soff = (unsigned)z * (unsigned)(a.h * a.w) * (unsigned)a.kb * 2u;  // (anything for an invalid plane)
      lds_dma(it_B, a.bytesB & (0u - (unsigned)ok), voffB[j], soff, (unsigned)(sBw * BSLOT + (tw + 4 * j) * 1024));
    } else {
      const int z = it_z0 + lt - 2, row = tw + 4 * (j - 3);
      const bool ok = live & (lt >= 2) & (z < a.d) & (it_y0 + row < a.h);  // (lt - 2 < zs by construction: lt < zs + 2)
      const unsigned soff = (unsigned)((z * a.h + it_y0 + row) * a.w) * (unsigned)a.ka * 2u;
      lds_dma(it_A, a.bytesA & (0u - (unsigned)ok), voffA, soff, (unsigned)(RB * BSLOT + sAw * ASLOT + row * 1024));
    }
  };

  // compute stage: at tick T the planes of load tick T - D are complete: dy plane in A slot (T - D) % RA, x planes z-1, z, z+1
  // in B slots (T - D - 2 .. T - D) % RB.  Operand addresses of a tick: one register for dy, one per tap for x.
  int vA = 0, vB[7];
  auto bases = [&](int cA, int cB0) {  // cB0: slot of plane z - 1
    vA = lanepart + RB * BSLOT + cA * ASLOT;
    int sb[3];
    sb[0] = cB0;
    sb[1] = cB0 + 1 >= RB ? cB0 + 1 - RB : cB0 + 1;
    sb[2] = cB0 + 2 >= RB ? cB0 + 2 - RB : cB0 + 2;
#pragma unroll
    for (int i = 0; i < 7; ++i) vB[i] = lanepart + (tap_dz[i] == 0 ? sb[0] : (tap_dz[i] == 1 ? sb[1] : sb[2])) * BSLOT + tap_off[i];
  };
  eltx4 opA[2][2], opB[2][7][2];  // two operand sets x (dy, 7 taps of x) x two transposing reads
  auto rd = [&](int set, int idx, int ks) {  // read number idx (0..15) of the operands of k-step ks (row ks of the plane)
    const int o = idx >> 1, second = (idx & 1) * 256;
    if (o == 0) opA[set][idx & 1] = MEDNET_DS_READ_TR16(smem + vA + ks * 1024 + second);
    else opB[set][o - 1][idx & 1] = MEDNET_DS_READ_TR16(smem + vB[o - 1] + ks * (HX * 64) + second);
  };

  int T = 0, ct = -D;         // global tick; tick inside the item of the compute stage (negative: pipeline filling)
  int cA = 0, cB0 = RB - 2;   // compute-stage slots at tick T = D: A slot 0, B slots (RB - 2, RB - 1, 0)
  // (slots advance only from tick D on, see the loop's tail)
  bases(cA, cB0);
  auto tick = [&](auto compute_tag) {
    constexpr bool COMPUTE = decltype(compute_tag)::value;
    const bool live = T < load_ticks;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks == 7) {
        // the planes of load tick T + 1 - D are needed from here on (the reads below fetch the first operands of tick T + 1):
        // of this wave's DMAs only those of the D - 1 youngest ticks may still be in flight; then the barrier makes that true
        // for every wave's share, and tells everybody that the slots tick T + 1 overwrites are no longer read
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 5) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int nA = cA + 1 >= RA ? 0 : cA + 1, nB = cB0 + 1 >= RB ? 0 : cB0 + 1;
        if (T + 1 >= D) bases(T + 1 == D ? cA : nA, T + 1 == D ? cB0 : nB);
      }
      const int nks = ks == 7 ? 0 : ks + 1;
      // issue order written out: one MFMA, then up to three transposing reads of the NEXT k-step's operands (front-loaded: the
      // last reads must be back before the next k-step's first MFMA), the tick's five DMAs one per k-step
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (COMPUTE) {
          const eltx8 fa = __builtin_shufflevector(opA[cur][0], opA[cur][1], 0, 1, 2, 3, 4, 5, 6, 7);
          const eltx8 fb = __builtin_shufflevector(opB[cur][i][0], opB[cur][i][1], 0, 1, 2, 3, 4, 5, 6, 7);
          acc[i] = MEDNET_MFMA_32x32x16(fa, fb, acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int first_rd[8] = {0, 3, 6, 9, 12, 14, 16, 16};
        for (int idx = first_rd[i]; idx < first_rd[i + 1]; ++idx) rd(nxt, idx, nks);
        if (i == 6 && ks < 5) dma(ks, live);  // (the gap without reads: 1 % faster than beside two of them, profiles/r05_ab.md 6)
      }
    }
  };
  if (my_items > 0) open_item(0);
  for (; T < total_ticks; ++T) {
    if (ct >= 2) tick(std::true_type{});
    else tick(std::false_type{});
    // advance the load stage ...
    sBw = sBw + 1 >= RB ? 0 : sBw + 1;
    sAw = sAw + 1 >= RA ? 0 : sAw + 1;
    if (++lt == TPI) {
      lt = 0;
      ++li;
      if (li < my_items) open_item(li);
    }
    // ... and the compute stage (D ticks behind; the slots were advanced for the prefetch inside tick())
    if (T + 1 > D) {
      cA = cA + 1 >= RA ? 0 : cA + 1;
      cB0 = cB0 + 1 >= RB ? 0 : cB0 + 1;
    }
    ct = ct + 1 == TPI ? 0 : ct + 1;
  }
  // ---- partial[wg][tap][a][b]: row a = (j&3) + 8*(j>>2) + 4*hk, col b = lane & 31 (the layout wgrad_mfma_reduce_kernel reads)
  float* out = a.part + (size_t)blockIdx.x * 27 * 1024;
  const int col = lane & 31;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = tw + 4 * i;
    if (tap < 27) {
#pragma unroll
      for (int j = 0; j < 16; ++j) out[((size_t)tap * 32 + (j & 3) + 8 * (j >> 2) + 4 * hk) * 32 + col] = acc[i][j];
    }
  }
}

// (volumes below one 8 x 16 tile or a few planes stay with wgrad_mfma2_kernel: a column's ring needs 2 + DEPTH ticks to fill)
static bool wgrad4_applies(int d, int h, int w) { return tuning_option("wgrad_v4", 1) != 0 && d >= 8 && h >= 8 && w >= 16; }
bool wgrad_mfma_coresident(int d, int h, int w) { return wgrad4_applies(d, h, w); }
static void wgrad4_plan(int n, int d, int h, int w, int ka, int kb, int workgroups, Wg4Args& a) {
  const int zmax = tuning_option("wgrad4_zs", 32);
  const int slabs = (d + zmax - 1) / zmax;  // z-slabs of at most 32 planes, equal depth
  a.zslabs = slabs;
  a.zs = (d + slabs - 1) / slabs;
  a.tiles_y = (h + 7) / 8;
  a.tiles_x = (w + 15) / 16;
  a.nitems = n * slabs * a.tiles_y * a.tiles_x;
  auto rcp = [](int d) { return d == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
  a.rcp_tiles_x = rcp(a.tiles_x); a.rcp_tiles_y = rcp(a.tiles_y); a.rcp_zslabs = rcp(a.zslabs);
  a.nab = (ka + 31) / 32;
  a.nbb = (kb + 31) / 32;
  const int pairs = a.nab * a.nbb;
  const int target = workgroups > 0 ? workgroups : 256;
  int splits = (target + pairs - 1) / pairs;
  if (splits > a.nitems) splits = a.nitems;
  if (splits < 1) splits = 1;
  a.splits = splits;
  a.xcd_remap = splits % 8 == 0 ? 1 : 0;
}
static size_t wgrad4_ws_bytes(int n, int d, int h, int w, int cin, int cout, int workgroups) {
  Wg4Args a;
  wgrad4_plan(n, d, h, w, cout, cin, workgroups, a);
  return (size_t)a.nab * a.nbb * a.splits * 27 * 1024 * sizeof(float);
}

// Launch plan of the 3x3x3 weight gradient for audits without a device (mednet_conv3d_wgrad_plan, tests/test_plan_audit.py):
// out[10] = {kind (4: wgrad_mfma4_kernel, z-columns; 2: wgrad_mfma2_kernel, bricks), workgroups, pairs, splits, items per pair,
//            tiles_x, tiles_y, tiles_z or z-slabs, planes per slab (kind 4) or brick width (kind 2), xcd_remap}
int wgrad_mfma_plan(int n, int d, int h, int w, int cin, int cout, int workgroups, int* out) {
  if (wgrad4_applies(d, h, w)) {
    Wg4Args a;
    wgrad4_plan(n, d, h, w, cout, cin, workgroups, a);
    const int v[10] = {4, a.nab * a.nbb * a.splits, a.nab * a.nbb, a.splits, a.nitems, a.tiles_x, a.tiles_y, a.zslabs, a.zs, a.xcd_remap};
    for (int i = 0; i < 10; ++i) out[i] = v[i];
  } else {
    Wg2Args a;
    wgrad2_plan(n, d, h, w, cout, cin, workgroups, a);
    const int v[10] = {2, a.nab * a.nbb * a.splits, a.nab * a.nbb, a.splits, a.ntiles, a.tiles_x, a.tiles_y, a.tiles_z, wgrad2_tx(w), 0};
    for (int i = 0; i < 10; ++i) out[i] = v[i];
  }
  return MEDNET_OK;
}

bool wgrad_mfma_supported(int cin, int cout, int ksize, int x_dtype, int dy_dtype, int x_layout, int dy_layout) {
  return ksize == 3 && cin % 16 == 0 && cout % 16 == 0 && x_dtype == ELT_DTYPE && dy_dtype == ELT_DTYPE &&
         x_layout == MEDNET_NDHWC && dy_layout == MEDNET_NDHWC;
}
// the kernel addresses its operands through buffer resources with 32-bit byte offsets
bool wgrad_mfma_fits(int n, int d, int h, int w, int cmax, int scale) {
  (void)n;
  return (double)d * h * w * scale * cmax * 2.0 < 4294960000.0;
}

size_t wgrad_mfma_ws_bytes(int n, int d, int h, int w, int cin, int cout, int ksize, int workgroups) {
  if (ksize != 3 || cin % 16 || cout % 16) return 0;
  const size_t v2 = wgrad2_ws_bytes(n, d, h, w, cin, cout, workgroups), v4 = wgrad4_ws_bytes(n, d, h, w, cin, cout, workgroups);
  return v2 > v4 ? v2 : v4;
}

int launch_wgrad_mfma(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cin, int cout, int dtype,
                      void* ws, size_t ws_bytes, hipStream_t s, int workgroups) {
  (void)dtype;
  // conv: A = dy (Cout rows), B = x (Cin cols) shifted by tap - 1
  if (wgrad4_applies(d, h, w)) {
    Wg4Args a;
    a.A = (const elt*)dy;
    a.B = (const elt*)x;
    a.part = (float*)ws;
    a.n = n; a.d = d; a.h = h; a.w = w; a.ka = cout; a.kb = cin;
    wgrad4_plan(n, d, h, w, cout, cin, workgroups, a);
    a.bytesA = (unsigned)((size_t)d * h * w * cout * 2);  // per sample
    a.bytesB = (unsigned)((size_t)d * h * w * cin * 2);
    const size_t need = (size_t)a.nab * a.nbb * a.splits * 27 * 1024 * sizeof(float);
    MEDNET_REQUIRE(ws_bytes >= need, MEDNET_E_WORKSPACE, "wgrad_mfma4: workspace %zu < %zu", ws_bytes, need);
    static bool attr4 = false;
    if (!attr4) {
      if (hipFuncSetAttribute((const void*)wgrad_mfma4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG4_LDS) != hipSuccess)
        return fail(MEDNET_E_HIP, "wgrad_mfma4: cannot raise dynamic LDS to %zu", WG4_LDS);
      attr4 = true;
    }
    hipLaunchKernelGGL(wgrad_mfma4_kernel, dim3(a.nab * a.nbb * a.splits), dim3(256), WG4_LDS, s, a);
    int rc4 = check_launch("wgrad_mfma4");
    if (rc4) return rc4;
    const size_t total4 = (size_t)a.nab * a.nbb * 1024 * 27;
    if (a.splits < 4) hipLaunchKernelGGL(wgrad_mfma_reduce_kernel_few, dim3((unsigned)(a.nab * a.nbb * 4)), dim3(256), 0, s, a.part, dw, cout, cin,
                       a.nbb, a.splits);
    else hipLaunchKernelGGL(wgrad_mfma_reduce_kernel, dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, s, a.part, dw, cout, cin,
                       a.nbb, a.splits);
    return check_launch("wgrad_mfma_reduce");
  }
  const int tx = wgrad2_tx(w);
  size_t lds = ((size_t)4 * 8 * tx + 6 * 10 * (tx + 2)) * 64;
  if (lds < 65536) lds = 65536;  // (the k-groups' merge area at the end: 4 tap waves x 4 slots x 16 x 64 floats)
  Wg2Args a;
  a.A = (const elt*)dy;
  a.B = (const elt*)x;
  a.part = (float*)ws;
  a.n = n; a.d = d; a.h = h; a.w = w; a.ka = cout; a.kb = cin;
  wgrad2_plan(n, d, h, w, cout, cin, workgroups, a);
  a.bytesA = (unsigned)((size_t)d * h * w * cout * 2);  // per sample
  a.bytesB = (unsigned)((size_t)d * h * w * cin * 2);
  const size_t need = (size_t)a.nab * a.nbb * a.splits * 27 * 1024 * sizeof(float);
  MEDNET_REQUIRE(ws_bytes >= need, MEDNET_E_WORKSPACE, "wgrad_mfma2: workspace %zu < %zu", ws_bytes, need);
  static bool attr_set[2] = {false, false};
  if (!attr_set[tx == 8]) {
    const hipError_t e = tx == 8 ? hipFuncSetAttribute((const void*)wgrad_mfma2_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                 : hipFuncSetAttribute((const void*)wgrad_mfma2_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(MEDNET_E_HIP, "wgrad_mfma2: cannot raise dynamic LDS to %zu", lds);
    attr_set[tx == 8] = true;
  }
  if (tx == 8) hipLaunchKernelGGL(wgrad_mfma2_kernel<8>, dim3(a.nab * a.nbb * a.splits), dim3(512), lds, s, a);
  else hipLaunchKernelGGL(wgrad_mfma2_kernel<16>, dim3(a.nab * a.nbb * a.splits), dim3(512), lds, s, a);
  int rc = check_launch("wgrad_mfma2");
  if (rc) return rc;
  const size_t total = (size_t)a.nab * a.nbb * 1024 * 27;
  if (a.splits < 4) hipLaunchKernelGGL(wgrad_mfma_reduce_kernel_few, dim3((unsigned)(a.nab * a.nbb * 4)), dim3(256), 0, s, a.part, dw, cout, cin,
                     a.nbb, a.splits);
  else hipLaunchKernelGGL(wgrad_mfma_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, a.part, dw, cout, cin,
                     a.nbb, a.splits);
  return check_launch("wgrad_mfma_reduce");
}

// ---- first-layer weight gradient (Cin = 1) on the matrix cores -----------------------------------------------------
//   dW[co][tap] = sum_v x[v + tap - 1] * dy[v][co]:  D[tap (27 of 32 rows)][co] += A[tap][k = voxel] * B[k = voxel][co].
// B comes from the dy brick in LDS through the transposing read (as in wgrad_mfma2); A is gathered from an fp32 halo brick
// of x (lane = tap row: 8 consecutive x-values of its shifted row) and split into elt hi + lo parts (two MFMAs), so the
// network input keeps fp32-level precision.  The kernel reads dy once and is bound by that (537 MB at config 2); the VALU
// kernel it replaces (27 FMAs per voxel and channel) took 0.65 ms.
struct Wc1Args {
  const float* x;  // N x D x H x W
  const elt* dy;  // N x D x H x W x cout
  float* part;     // [workgroup][cout][27]
  int n, d, h, w, cout;
  int tiles_z, tiles_y, tiles_x, ntiles;
  unsigned rcp_tiles_x, rcp_tiles_y, rcp_tiles_z;
  unsigned bytes_x, bytes_dy;  // per sample
  int x16;                     // 1: x holds elt values, else fp32
  // GN form: `dy` holds dz (the gradient of the layer's activated, normalised output) and the kernel applies GroupNorm's
  // backward while it stages:  dy = k1 * dz * act'(ca * y + cb) + k2 * y + k3, rounded to elt -- the value
  // norm_act.hip's gn_bwd_apply_kernel would have stored, expression for expression -- so dy is never written or re-read
  const elt* y;        // N x D x H x W x cout: the convolution's output
  const float* coef;   // [n][cout][2] = {ca, cb}
  const float* bcoef;  // [n][cout][3] = {k1, k2, k3}
  int act;
};

template <int NB, bool GN>  // 32-channel blocks of dy
__global__ __launch_bounds__(256, 2) void wgrad_c1_mfma_kernel(Wc1Args a) {
  constexpr int TZ = 4, TY = 8, TX = 16, HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
  constexpr int NJ = TZ * TY * TX, NH = HZ * HY * HX;
  constexpr int ROWB = 64 * NB;                         // bytes of one voxel row of dy in LDS
  constexpr int XH_BYTES = (NH * 4 + 255) / 256 * 256;  // fp32 halo brick of x
  constexpr int DY_ROUNDS = NJ * 4 * NB / 256, X_ROUNDS = (NH + 255) / 256;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
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

  constexpr unsigned OOB = 0xFFFFFF00u;
  u32x4 rdy[DY_ROUNDS];
  u32x4 ryy[GN ? DY_ROUNDS : 1];
  float rx[X_ROUNDS];
  // GN form: this thread's 8 channels are the same in every round (256 % (4 * NB) == 0)
  float gca[GN ? 8 : 1], gcb[GN ? 8 : 1], gk1[GN ? 8 : 1], gk2[GN ? 8 : 1], gk3[GN ? 8 : 1];
  unsigned in_mask = 0;  // bit `it`: round `it` of the fetched brick lies inside the volume
  int coef_n = -1;       // sample the coefficients in registers belong to
  auto fetch = [&](int tile) {
    int tt = tile;
    int qd = fastdiv(tt, a.tiles_x, a.rcp_tiles_x);
    const int tx0 = (tt - qd * a.tiles_x) * TX;
    tt = qd;
    qd = fastdiv(tt, a.tiles_y, a.rcp_tiles_y);
    const int ty0 = (tt - qd * a.tiles_y) * TY;
    tt = qd;
    qd = fastdiv(tt, a.tiles_z, a.rcp_tiles_z);
    const int tz0 = (tt - qd * a.tiles_z) * TZ;
    const size_t svox = (size_t)qd * a.d * a.h * a.w;
    const auto rD = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + svox * a.cout), 0, a.bytes_dy, 0x00020000);
    const auto rY = __builtin_amdgcn_make_buffer_rsrc((void*)((GN ? a.y : a.dy) + svox * a.cout), 0, a.bytes_dy, 0x00020000);
    if constexpr (GN) {
      in_mask = 0;
      if (qd != coef_n) {  // (workgroup-uniform) first brick of a sample
        coef_n = qd;
        const int ch0 = (tid % (4 * NB)) * 8;
        if (ch0 < a.cout) {
          const float* pc = a.coef + ((size_t)qd * a.cout + ch0) * 2;
          const float* pb = a.bcoef + ((size_t)qd * a.cout + ch0) * 3;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            gca[k] = pc[2 * k];
            gcb[k] = pc[2 * k + 1];
            gk1[k] = pb[3 * k];
            gk2[k] = pb[3 * k + 1];
            gk3[k] = pb[3 * k + 2];
          }
        }
      }
    }
    const auto rX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + svox), 0, a.bytes_x, 0x00020000);
    const auto rX16 = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const elt*>(a.x) + svox), 0, a.bytes_x, 0x00020000);
#pragma unroll
    for (int it = 0; it < DY_ROUNDS; ++it) {
      const int c = it * 256 + tid;
      const int part = c % (4 * NB), v = c / (4 * NB);
      const int gz = tz0 + v / (TX * TY), gy = ty0 + (v / TX) % TY, gx = tx0 + v % TX;
      const bool in_vol = (gz < a.d) & (gy < a.h) & (gx < a.w) & (part * 8 < a.cout);  // (16 channels: the row's second half is zeros)
      const unsigned off = ((unsigned)((gz * a.h + gy) * a.w + gx) * (unsigned)a.cout + part * 8) * 2u;
      rdy[it] = __builtin_amdgcn_raw_buffer_load_b128(rD, in_vol ? off : OOB, 0, 0);
      if constexpr (GN) {
        ryy[it] = __builtin_amdgcn_raw_buffer_load_b128(rY, in_vol ? off : OOB, 0, 0);
        in_mask |= in_vol ? 1u << it : 0u;
      }
    }
#pragma unroll
    for (int it = 0; it < X_ROUNDS; ++it) {
      const int v = it * 256 + tid;
      const int gz = tz0 - 1 + v / (HX * HY), gy = ty0 - 1 + (v / HX) % HY, gx = tx0 - 1 + v % HX;
      const bool in_vol = (v < NH) & ((unsigned)gz < (unsigned)a.d) & ((unsigned)gy < (unsigned)a.h) & ((unsigned)gx < (unsigned)a.w);
      const unsigned vidx = (unsigned)((gz * a.h + gy) * a.w + gx);
      if (a.x16) {  // (wave-uniform) 2-byte elements: a 16-bit buffer load of the element, widened
        const unsigned short raw = __builtin_amdgcn_raw_buffer_load_b16(rX16, in_vol ? vidx * 2u : OOB, 0, 0);
        rx[it] = (float)__builtin_bit_cast(elt, raw);
      } else {
        rx[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, in_vol ? vidx * 4u : OOB, 0, 0));
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < DY_ROUNDS; ++it) {
      if constexpr (GN) {
        const eltx8 gz8 = __builtin_bit_cast(eltx8, rdy[it]), yv8 = __builtin_bit_cast(eltx8, ryy[it]);
        float g[8], u[8], yy[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          yy[k] = (float)yv8[k];
          g[k] = (float)gz8[k];
          u[k] = fmaf(gca[k], yy[k], gcb[k]);
        }
        act_grad_pre_n<8>(g, u, a.act);
        eltx8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (elt)fmaf(gk1[k], g[k], fmaf(gk2[k], yy[k], gk3[k]));
        const u32x4 zero = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(dyl + (it * 256 + tid) * 16) = (in_mask >> it) & 1u ? __builtin_bit_cast(u32x4, o) : zero;
      } else {
        *reinterpret_cast<u32x4*>(dyl + (it * 256 + tid) * 16) = rdy[it];
      }
    }
#pragma unroll
    for (int it = 0; it < X_ROUNDS; ++it) {
      const int v = it * 256 + tid;
      if (v < NH) xh[v] = rx[it];
    }
  };

  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[b][j] = 0.f;

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
      eltx8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = px[j];
        hi[j] = (elt)xv;
        lo[j] = (elt)(xv - (float)hi[j]);
      }
      const char* brow = dyl + (row * TX + 8 * hk + q) * ROWB + coloff;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const eltx8 fb = tr_operand(brow + b * 64, 4 * ROWB);
        acc[b] = MEDNET_MFMA_32x32x16(hi, fb, acc[b], 0, 0, 0);
        acc[b] = MEDNET_MFMA_32x32x16(lo, fb, acc[b], 0, 0, 0);
      }
    }
  }
  // ---- sum the 4 waves in LDS (fixed order), write the workgroup's partial in dW layout [co][27]
  float* red = reinterpret_cast<float*>(smem);  // [4 waves][NB][16][64 lanes]
  __syncthreads();
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 16; ++j) red[((wv * NB + b) * 16 + j) * 64 + lane] = acc[b][j];
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float s = (red[((0 * NB + b) * 16 + j) * 64 + lane] + red[((1 * NB + b) * 16 + j) * 64 + lane]) +
                        (red[((2 * NB + b) * 16 + j) * 64 + lane] + red[((3 * NB + b) * 16 + j) * 64 + lane]);
        const int tap = (j & 3) + 8 * (j >> 2) + 4 * hk, co = b * 32 + (lane & 31);
        if (tap < 27 && co < a.cout) a.part[((size_t)blockIdx.x * a.cout + co) * 27 + tap] = s;
      }
  }
}

bool wgrad_c1_mfma_supported(int cout, int x_dtype, int dy_dtype) {
  return (cout == 16 || cout == 32 || cout == 64) && (x_dtype == MEDNET_F32 || x_dtype == ELT_DTYPE) && dy_dtype == ELT_DTYPE;
}
int wgrad_c1_mfma_blocks(int n, int d, int h, int w) {
  const int nt = n * ((d + 3) / 4) * ((h + 7) / 8) * ((w + 15) / 16);
  return nt < 1024 ? nt : 1024;
}
int launch_wgrad_c1_mfma(const void* x, const void* dy, float* part, int n, int d, int h, int w, int cout, hipStream_t s,
                         int x_dtype, const void* gn_y, const float* gn_coef, const float* gn_bcoef, int gn_act) {
  Wc1Args a;
  const bool gn = gn_y != nullptr;  // dy is dz: GroupNorm's backward applied while staging (Wc1Args)
  MEDNET_REQUIRE(!gn || (gn_coef && gn_bcoef), MEDNET_E_SHAPE, "wgrad_c1_mfma: the GroupNorm form needs both coefficient tables");
  a.y = (const elt*)gn_y; a.coef = gn_coef; a.bcoef = gn_bcoef; a.act = gn_act;
  a.x = (const float*)x;
  a.x16 = x_dtype != MEDNET_F32;
  a.dy = (const elt*)dy;
  a.part = part;
  a.n = n; a.d = d; a.h = h; a.w = w; a.cout = cout;
  a.tiles_z = (d + 3) / 4; a.tiles_y = (h + 7) / 8; a.tiles_x = (w + 15) / 16;
  a.ntiles = n * a.tiles_z * a.tiles_y * a.tiles_x;
  auto rcp = [](int d) { return d == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
  a.rcp_tiles_x = rcp(a.tiles_x); a.rcp_tiles_y = rcp(a.tiles_y); a.rcp_tiles_z = rcp(a.tiles_z);
  MEDNET_REQUIRE((double)d * h * w * cout * 2.0 < 4294960000.0, MEDNET_E_UNSUPPORTED, "wgrad_c1_mfma: one sample must stay below 4 GB");
  a.bytes_x = (unsigned)((size_t)d * h * w * (a.x16 ? 2 : 4));
  a.bytes_dy = (unsigned)((size_t)d * h * w * cout * 2);
  const int blocks = wgrad_c1_mfma_blocks(n, d, h, w);
  const int nb = (cout + 31) / 32;
  const size_t stage = 4352 + (size_t)512 * 64 * nb, red = (size_t)4 * nb * 16 * 64 * 4;
  const size_t lds = stage > red ? stage : red;
  if (nb == 1) {
    if (gn) hipLaunchKernelGGL((wgrad_c1_mfma_kernel<1, true>), dim3(blocks), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((wgrad_c1_mfma_kernel<1, false>), dim3(blocks), dim3(256), lds, s, a);
  } else {
    static bool attr_set[2] = {false, false};
    if (!attr_set[gn]) {
      const void* fn = gn ? (const void*)wgrad_c1_mfma_kernel<2, true> : (const void*)wgrad_c1_mfma_kernel<2, false>;
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return fail(MEDNET_E_HIP, "wgrad_c1_mfma: cannot raise dynamic LDS to %zu", lds);
      attr_set[gn] = true;
    }
    if (gn) hipLaunchKernelGGL((wgrad_c1_mfma_kernel<2, true>), dim3(blocks), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((wgrad_c1_mfma_kernel<2, false>), dim3(blocks), dim3(256), lds, s, a);
  }
  return check_launch("wgrad_c1_mfma");
}

// ---- ConvTranspose3d weight gradient, second generation: output-parity classes ------------------------------------
//   dW[k][ci][co] = sum_i x[i][ci] * dy[2i - 1 + k][co]          (per dimension: k=1 -> dy[2i], k=2 -> dy[2i+1], k=0 -> dy[2i-1])
// With E[j] = dy[2j], O[j] = dy[2j+1] per dimension:  k=1: x[j] E[j],  k=2: x[j] O[j],  k=0: x[j+1] O[j].  So a workgroup
// that owns the low-resolution brick j in [j0, j0+T) needs exactly the 2T x 2T x 2T block of dy that starts at 2 j0 -- NO
// halo on the big tensor, every dy voxel is staged once by one workgroup -- plus the (T+1)^3-ish halo brick of x.  The
// first kernel staged a (2T+1)^3 halo of dy per brick (2.8x at 1x4x16) and was bound by that.  In LDS dy is stored by
// parity class [8][128 voxels][32 ch]; tap (kz,ky,kx) reads class (k != 1) and x shifted by (k == 0).
// 512 threads: 4 tap-waves x 2 k-groups as in wgrad_mfma2.  Taps are dealt so that the taps of a wave share few x
// shifts (slots 0-3 one shift, 4-5 one, 6 one => 3 A operands + 7 B operands per k-step):
//   wave 0: 13 14 16 17 | 22 23 | 25     (shift 000 throughout)
//   wave 1: 12 15 21 24 |  1  2 |  0     (shifts 001, 110, 111)
//   wave 2: 10 11 19 20 |  3  6 | 26     (shifts 010, 101, 000)
//   wave 3:  4  5  7  8 |  9 18 | (18)   (shifts 100, 011; the 7th slot is a discarded duplicate)
struct Ct2Args {
  const elt* A;  // x  (n, d, h, w, ka)
  const elt* B;  // dy (n, 2d, 2h, 2w, kb)
  float* part;    // [wg][27][32][32]
  int n, d, h, w, ka, kb;
  int tiles_z, tiles_y, tiles_x, ntiles;
  int nab, nbb, splits;
  unsigned rcp_tiles_x, rcp_tiles_y, rcp_tiles_z;
  unsigned bytesA, bytesB;  // per sample
};

__global__ __launch_bounds__(512, 2) void convt_wgrad_mfma2_kernel(Ct2Args a) {
  constexpr int TZ = 2, TY = 4, TX = 16, HZ = TZ + 1, HY = TY + 1, HX = TX + 1;
  constexpr int NJ = TZ * TY * TX, NAH = HZ * HY * HX;      // 128 brick voxels, 255 halo voxels of x
  constexpr int A_ROUNDS = (NAH * 4 + 511) / 512, B_ROUNDS = 8 * NJ * 4 / 512;  // 2 + 8 loads per thread
  constexpr int KSTEPS = NJ / 16;
  constexpr int A_BYTES = 16384;  // 255 x 64 B rounded up
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* A_lds = smem;             // [HZ][HY][HX][32 ch]
  char* B_lds = smem + A_BYTES;   // [8 classes][TZ][TY][TX][32 ch]

  const int pair = blockIdx.x / a.splits, split = blockIdx.x % a.splits;
  const int ab = pair / a.nbb, bb = pair % a.nbb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kgrp = wvu >> 2, tw = wvu & 3;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, hk = lane >> 5;
  const int coloff = (16 * (g & 1) + 4 * p) * 2;

  // ---- this wave's 7 taps: byte offsets of their dy class image and of their x shift (wave-uniform => SGPRs)
  int tap_of[7], boff[7], aoff[3];
  {
    const int t0 = tw == 0 ? 13 : tw == 1 ? 12 : tw == 2 ? 10 : 4;
    const int t1 = tw == 0 ? 14 : tw == 1 ? 15 : tw == 2 ? 11 : 5;
    const int t2 = tw == 0 ? 16 : tw == 1 ? 21 : tw == 2 ? 19 : 7;
    const int t3 = tw == 0 ? 17 : tw == 1 ? 24 : tw == 2 ? 20 : 8;
    const int t4 = tw == 0 ? 22 : tw == 1 ? 1 : tw == 2 ? 3 : 9;
    const int t5 = tw == 0 ? 23 : tw == 1 ? 2 : tw == 2 ? 6 : 18;
    const int t6 = tw == 0 ? 25 : tw == 1 ? 0 : tw == 2 ? 26 : 18;
    const int tt[7] = {t0, t1, t2, t3, t4, t5, t6};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int kz = tt[i] / 9, ky = (tt[i] / 3) % 3, kx = tt[i] % 3;
      tap_of[i] = tt[i];
      boff[i] = (((kz != 1) * 4 + (ky != 1) * 2 + (kx != 1)) * NJ) * 64;
      if (i == 0 || i == 4 || i == 6) aoff[i == 0 ? 0 : i == 4 ? 1 : 2] = (((kz == 0) * HY + (ky == 0)) * HX + (kx == 0)) * 64;
    }
  }

  constexpr unsigned OOB = 0xFFFFFF00u;
  u32x4 regA[A_ROUNDS], regB[B_ROUNDS];
  struct Next {
    int tz0, ty0, tx0;
    unsigned kill;
    __amdgpu_buffer_rsrc_t rA, rB;
  };
  auto plan_next = [&](int tile, bool valid) {
    Next nx;
    int tt = valid ? tile : 0;
    int qd = fastdiv(tt, a.tiles_x, a.rcp_tiles_x);
    nx.tx0 = (tt - qd * a.tiles_x) * TX;
    tt = qd;
    qd = fastdiv(tt, a.tiles_y, a.rcp_tiles_y);
    nx.ty0 = (tt - qd * a.tiles_y) * TY;
    tt = qd;
    qd = fastdiv(tt, a.tiles_z, a.rcp_tiles_z);
    nx.tz0 = (tt - qd * a.tiles_z) * TZ;
    const size_t svox = (size_t)qd * a.d * a.h * a.w;  // one resource per sample
    nx.rA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + svox * a.ka), 0, a.bytesA, 0x00020000);
    nx.rB = __builtin_amdgcn_make_buffer_rsrc((void*)(a.B + svox * 8 * a.kb), 0, a.bytesB, 0x00020000);
    nx.kill = valid ? 0u : OOB;
    return nx;
  };
  auto fetch_one = [&](int j, const Next& nx) {
    if (j < B_ROUNDS) {  // dy: 16 consecutive high-resolution voxels (1 KB contiguous) per wave instruction
      const int c = j * 512 + tid;
      const int part = c & 3, vx = (c >> 2) & 31, vy = (c >> 7) & 7, vz = c >> 10;
      const int gz = 2 * nx.tz0 + vz, gy = 2 * nx.ty0 + vy, gx = 2 * nx.tx0 + vx;
      const bool in_vol = (gz < 2 * a.d) & (gy < 2 * a.h) & (gx < 2 * a.w) & (bb * 32 + part * 8 < a.kb);
      const unsigned off = ((unsigned)((gz * 2 * a.h + gy) * 2 * a.w + gx) * (unsigned)a.kb + bb * 32 + part * 8) * 2u;
      regB[j] = __builtin_amdgcn_raw_buffer_load_b128(nx.rB, (in_vol ? off : OOB) | nx.kill, 0, 0);
    } else {  // x halo brick (the +1 planes are beyond the volume at its far faces: hardware zeros)
      const int it = j - B_ROUNDS;
      const int c = it * 512 + tid;
      const int part = c & 3, v = c >> 2;
      const int gz = nx.tz0 + v / (HX * HY), gy = nx.ty0 + (v / HX) % HY, gx = nx.tx0 + v % HX;
      const bool in_vol = (v < NAH) & (gz < a.d) & (gy < a.h) & (gx < a.w) & (ab * 32 + part * 8 < a.ka);
      const unsigned off = ((unsigned)((gz * a.h + gy) * a.w + gx) * (unsigned)a.ka + ab * 32 + part * 8) * 2u;
      regA[it] = __builtin_amdgcn_raw_buffer_load_b128(nx.rA, (in_vol ? off : OOB) | nx.kill, 0, 0);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < B_ROUNDS; ++it) {
      const int c = it * 512 + tid;
      const int part = c & 3, vx = (c >> 2) & 31, vy = (c >> 7) & 7, vz = c >> 10;
      const int cls = (vz & 1) * 4 + (vy & 1) * 2 + (vx & 1);
      const int j = ((vz >> 1) * TY + (vy >> 1)) * TX + (vx >> 1);
      *reinterpret_cast<u32x4*>(B_lds + ((cls * NJ + j) * 4 + part) * 16) = regB[it];
    }
#pragma unroll
    for (int it = 0; it < A_ROUNDS; ++it) {
      const int c = it * 512 + tid;
      if (c < NAH * 4) *reinterpret_cast<u32x4*>(A_lds + c * 16) = regA[it];
    }
  };

  f32x16 acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const char* Ab = A_lds + coloff + (8 * hk + q) * 64;
  const char* Bb = B_lds + coloff + (8 * hk + q) * 64;

  int tile = split;
  const int t_step = a.splits, t_end = a.ntiles;
  if (tile < t_end) {
    const Next first = plan_next(tile, true);
#pragma unroll
    for (int j = 0; j < A_ROUNDS + B_ROUNDS; ++j) fetch_one(j, first);
  }
  constexpr int LOADS_PER_STEP = (A_ROUNDS + B_ROUNDS + KSTEPS / 2 - 1) / (KSTEPS / 2);
  for (; tile < t_end; tile += t_step) {
    __syncthreads();  // previous brick fully consumed
    commit();
    __syncthreads();
    const Next nx = plan_next(tile + t_step, tile + t_step < t_end);  // flies while this brick is on the matrix cores
#pragma unroll
    for (int k2 = 0; k2 < KSTEPS / 2; ++k2) {
      const int ks = 2 * k2 + kgrp;  // the two wave groups interleave the brick's k-steps (rows of 16 x-voxels)
      const char* arow = Ab + (((ks / TY) * HY + ks % TY) * HX) * 64;
      const char* brow = Bb + (ks * TX) * 64;
      eltx8 fa[3], fb[7];
#pragma unroll
      for (int i = 0; i < 3; ++i) fa[i] = tr_operand(arow + aoff[i], 4 * 64);
#pragma unroll
      for (int i = 0; i < 7; ++i) fb[i] = tr_operand(brow + boff[i], 4 * 64);
#pragma unroll
      for (int l = 0; l < LOADS_PER_STEP; ++l)
        if (k2 * LOADS_PER_STEP + l < A_ROUNDS + B_ROUNDS) fetch_one(k2 * LOADS_PER_STEP + l, nx);
#pragma unroll
      for (int i = 0; i < 7; ++i) acc[i] = MEDNET_MFMA_32x32x16(fa[i < 4 ? 0 : i < 6 ? 1 : 2], fb[i], acc[i], 0, 0, 0);
    }
  }
  // k-group merge through LDS and write-out: as in wgrad_mfma2
  float* mlds = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i0 = 0; i0 < 7; i0 += 4) {
    __syncthreads();
    if (kgrp == 1) {
#pragma unroll
      for (int i = i0; i < (i0 + 4 < 7 ? i0 + 4 : 7); ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) mlds[((tw * 4 + (i - i0)) * 16 + j) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    if (kgrp == 0) {
#pragma unroll
      for (int i = i0; i < (i0 + 4 < 7 ? i0 + 4 : 7); ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] += mlds[((tw * 4 + (i - i0)) * 16 + j) * 64 + lane];
    }
  }
  if (kgrp == 0) {
    float* out = a.part + (size_t)blockIdx.x * 27 * 1024;
    const int col = lane & 31;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      if (tw == 3 && i == 6) continue;  // the duplicate slot
#pragma unroll
      for (int j = 0; j < 16; ++j) out[((size_t)tap_of[i] * 32 + (j & 3) + 8 * (j >> 2) + 4 * hk) * 32 + col] = acc[i][j];
    }
  }
}

static void ct2_plan(int n, int d, int h, int w, int ka, int kb, int workgroups, Ct2Args& a) {
  a.tiles_z = (d + 1) / 2;
  a.tiles_y = (h + 3) / 4;
  a.tiles_x = (w + 15) / 16;
  a.ntiles = n * a.tiles_z * a.tiles_y * a.tiles_x;
  auto rcp = [](int d) { return d == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
  a.rcp_tiles_x = rcp(a.tiles_x); a.rcp_tiles_y = rcp(a.tiles_y); a.rcp_tiles_z = rcp(a.tiles_z);
  a.nab = (ka + 31) / 32;
  a.nbb = (kb + 31) / 32;
  const int pairs = a.nab * a.nbb;
  int splits = ((workgroups > 0 ? workgroups : 256) + pairs - 1) / pairs;  // one workgroup per CU (or what the caller asks for: wgrad2_plan)
  if (splits > a.ntiles) splits = a.ntiles;
  if (splits < 1) splits = 1;
  a.splits = splits;
}

size_t convt_wgrad_mfma_ws_bytes(int n, int d, int h, int w, int cin, int cout, int workgroups) {
  if (cin % 32 || cout % 32) return 0;
  Ct2Args b;
  ct2_plan(n, d, h, w, cin, cout, workgroups, b);
  return (size_t)b.nab * b.nbb * b.splits * 27 * 1024 * sizeof(float);
}

int launch_convt_wgrad_mfma(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cin, int cout,
                            void* ws, size_t ws_bytes, hipStream_t s, int workgroups) {
  // convT: A = x (Cin rows) on the (d,h,w) grid, B = dy (Cout cols) on the (2d,2h,2w) grid at 2v - 1 + tap
  constexpr size_t lds = 16384 + 8 * 128 * 64;
  Ct2Args a;
  a.A = (const elt*)x;
  a.B = (const elt*)dy;
  a.part = (float*)ws;
  a.n = n; a.d = d; a.h = h; a.w = w; a.ka = cin; a.kb = cout;
  ct2_plan(n, d, h, w, cin, cout, workgroups, a);
  a.bytesA = (unsigned)((size_t)d * h * w * cin * 2);
  a.bytesB = (unsigned)((size_t)8 * d * h * w * cout * 2);
  const size_t need = (size_t)a.nab * a.nbb * a.splits * 27 * 1024 * sizeof(float);
  MEDNET_REQUIRE(ws_bytes >= need, MEDNET_E_WORKSPACE, "convt_wgrad_mfma2: workspace %zu < %zu", ws_bytes, need);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)convt_wgrad_mfma2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(MEDNET_E_HIP, "convt_wgrad_mfma2: cannot raise dynamic LDS to %zu", lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(convt_wgrad_mfma2_kernel, dim3(a.nab * a.nbb * a.splits), dim3(512), lds, s, a);
  int rc = check_launch("convt_wgrad_mfma2");
  if (rc) return rc;
  const size_t total = (size_t)a.nab * a.nbb * 1024 * 27;
  if (a.splits < 4) hipLaunchKernelGGL(wgrad_mfma_reduce_kernel_few, dim3((unsigned)(a.nab * a.nbb * 4)), dim3(256), 0, s, a.part, dw, cin, cout,
                     a.nbb, a.splits);
  else hipLaunchKernelGGL(wgrad_mfma_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, a.part, dw, cin, cout,
                     a.nbb, a.splits);
  return check_launch("wgrad_mfma_reduce");
}

}  // namespace mednet
