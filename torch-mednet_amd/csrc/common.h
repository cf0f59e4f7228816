// Shared device/host helpers for libmednet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mednet_hip.h"

// ---- error plumbing + tuning knobs (defined once in api.hip, global names: conv_mfma.hip is compiled a second time under
//      a renamed namespace for the fp16 element type and must still reach them) -------------------------------------
int mednet_internal_fail(int code, const char* fmt, ...);
int mednet_internal_check_launch(const char* what);
int mednet_internal_tuning_option(const char* name, int default_value);
int mednet_internal_cu_count(void);

namespace mednet {

typedef __bf16 bf16;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <typename... A>
static inline int fail(int code, const char* fmt, A... a) { return ::mednet_internal_fail(code, fmt, a...); }
static inline int check_launch(const char* what) { return ::mednet_internal_check_launch(what); }
// tuning knobs (mednet_set_option): experiments A/B kernel variants inside ONE process
static inline int tuning_option(const char* name, int default_value) { return ::mednet_internal_tuning_option(name, default_value); }

#define MEDNET_REQUIRE(cond, code, ...)        \
  do {                                         \
    if (!(cond)) return fail(code, __VA_ARGS__); \
  } while (0)

inline bool dtype_ok(int dt) { return dt == MEDNET_F32 || dt == MEDNET_BF16 || dt == MEDNET_F16; }
inline size_t dtype_size(int dt) { return dt == MEDNET_F32 ? 4 : 2; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- scalar load/store with conversion ---------------------------------------------------------------------
__device__ __forceinline__ float ld(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float ld(const bf16* p, size_t i) { return (float)p[i]; }
__device__ __forceinline__ float ld(const f16* p, size_t i) { return (float)p[i]; }
__device__ __forceinline__ float ld(const uint8_t* p, size_t i) { return (float)p[i]; }
__device__ __forceinline__ void st(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void st(bf16* p, size_t i, float v) { p[i] = (bf16)v; }
__device__ __forceinline__ void st(f16* p, size_t i, float v) { p[i] = (f16)v; }

// ---- 8-wide vector load/store (caller guarantees 8-element alignment) ----------------------------------------
struct F8 {
  float v[8];
};
__device__ __forceinline__ F8 ld8(const float* p, size_t i) {
  F8 r;
  const f32x4 a = *reinterpret_cast<const f32x4*>(p + i);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p + i + 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r.v[k] = a[k];
    r.v[k + 4] = b[k];
  }
  return r;
}
__device__ __forceinline__ F8 ld8(const bf16* p, size_t i) {
  F8 r;
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p + i);
#pragma unroll
  for (int k = 0; k < 8; ++k) r.v[k] = (float)a[k];
  return r;
}
__device__ __forceinline__ F8 ld8(const f16* p, size_t i) {
  F8 r;
  const f16x8 a = *reinterpret_cast<const f16x8*>(p + i);
#pragma unroll
  for (int k = 0; k < 8; ++k) r.v[k] = (float)a[k];
  return r;
}
// last-use streaming loads (nt): the line is not kept in L2 once delivered
__device__ __forceinline__ F8 ld8_nt(const float* p, size_t i) {
  F8 r;
  const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i));
  const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i + 4));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r.v[k] = a[k];
    r.v[k + 4] = b[k];
  }
  return r;
}
__device__ __forceinline__ F8 ld8_nt(const bf16* p, size_t i) {
  F8 r;
  const bf16x8 a = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p + i));
#pragma unroll
  for (int k = 0; k < 8; ++k) r.v[k] = (float)a[k];
  return r;
}
__device__ __forceinline__ F8 ld8_nt(const f16* p, size_t i) {
  F8 r;
  const f16x8 a = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(p + i));
#pragma unroll
  for (int k = 0; k < 8; ++k) r.v[k] = (float)a[k];
  return r;
}
__device__ __forceinline__ void st8(f16* p, size_t i, const F8& r) {
  f16x8 a;
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = (f16)r.v[k];
  *reinterpret_cast<f16x8*>(p + i) = a;
}
__device__ __forceinline__ void st8(float* p, size_t i, const F8& r) {
  f32x4 a, b;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = r.v[k];
    b[k] = r.v[k + 4];
  }
  *reinterpret_cast<f32x4*>(p + i) = a;
  *reinterpret_cast<f32x4*>(p + i + 4) = b;
}
__device__ __forceinline__ void st8(bf16* p, size_t i, const F8& r) {
  bf16x8 a;
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = (bf16)r.v[k];
  *reinterpret_cast<bf16x8*>(p + i) = a;
}

// ---- activations: components.py:36-40 (ReLU, LeakyReLU(0.1), ELU(alpha=1)), all "in-place" in the reference,
//      so the backward is written in terms of the OUTPUT z (ATen elu_backward(is_result=true)) ---------------
__device__ __forceinline__ float act_apply(float u, int act) {
  switch (act) {
    case MEDNET_ACT_RELU: return u > 0.f ? u : 0.f;
    case MEDNET_ACT_LEAKY: return u > 0.f ? u : 0.1f * u;
    case MEDNET_ACT_ELU: return u > 0.f ? u : __expf(u) - 1.f;  // ATen computes exp(x) - 1 too (not expm1)
    default: return u;
  }
}
__device__ __forceinline__ float act_grad_from_out(float z, int act) {
  switch (act) {
    case MEDNET_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case MEDNET_ACT_LEAKY: return z > 0.f ? 1.f : 0.1f;
    case MEDNET_ACT_ELU: return z > 0.f ? 1.f : z + 1.f;
    default: return 1.f;
  }
}

// MFMA accumulator form: ONE (wave-uniform) branch per 16 values; nothing at all for MEDNET_ACT_NONE (a switch per
// element costs a conv epilogue thousands of cycles in scalar branches)
typedef __attribute__((ext_vector_type(16))) float act_f32x16;
__device__ __forceinline__ void act_apply_v16(act_f32x16& u, int act) {
  if (act == MEDNET_ACT_NONE) return;
  if (act == MEDNET_ACT_RELU) {
#pragma unroll
    for (int k = 0; k < 16; ++k) u[k] = u[k] > 0.f ? u[k] : 0.f;
  } else if (act == MEDNET_ACT_LEAKY) {
#pragma unroll
    for (int k = 0; k < 16; ++k) u[k] = u[k] > 0.f ? u[k] : 0.1f * u[k];
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) u[k] = u[k] > 0.f ? u[k] : __expf(u[k]) - 1.f;
  }
}
// N-wide forms: ONE (wave-uniform) switch per vector instead of one per element
template <int N>
__device__ __forceinline__ void act_apply_n(float* u, int act) {
  switch (act) {
    case MEDNET_ACT_RELU:
#pragma unroll
      for (int k = 0; k < N; ++k) u[k] = u[k] > 0.f ? u[k] : 0.f;
      break;
    case MEDNET_ACT_LEAKY:
#pragma unroll
      for (int k = 0; k < N; ++k) u[k] = u[k] > 0.f ? u[k] : 0.1f * u[k];
      break;
    case MEDNET_ACT_ELU:
#pragma unroll
      for (int k = 0; k < N; ++k) u[k] = u[k] > 0.f ? u[k] : __expf(u[k]) - 1.f;
      break;
    default: break;
  }
}
// g[k] *= act'(.) expressed through the activation OUTPUT z
template <int N>
__device__ __forceinline__ void act_grad_n(float* g, const float* z, int act) {
  switch (act) {
    case MEDNET_ACT_RELU:
#pragma unroll
      for (int k = 0; k < N; ++k) g[k] = z[k] > 0.f ? g[k] : 0.f;
      break;
    case MEDNET_ACT_LEAKY:
#pragma unroll
      for (int k = 0; k < N; ++k) g[k] = z[k] > 0.f ? g[k] : 0.1f * g[k];
      break;
    case MEDNET_ACT_ELU:
#pragma unroll
      for (int k = 0; k < N; ++k) g[k] = z[k] > 0.f ? g[k] : g[k] * (z[k] + 1.f);
      break;
    default: break;
  }
}

// g[k] *= act'(u) from the pre-activation u (used when z is not read back: u is recomputed from the conv output)
template <int N>
__device__ __forceinline__ void act_grad_pre_n(float* g, const float* u, int act) {
  switch (act) {
    case MEDNET_ACT_RELU:
#pragma unroll
      for (int k = 0; k < N; ++k) g[k] = u[k] > 0.f ? g[k] : 0.f;
      break;
    case MEDNET_ACT_LEAKY:
#pragma unroll
      for (int k = 0; k < N; ++k) g[k] = u[k] > 0.f ? g[k] : 0.1f * g[k];
      break;
    case MEDNET_ACT_ELU:
#pragma unroll
      for (int k = 0; k < N; ++k) g[k] = u[k] > 0.f ? g[k] : g[k] * __expf(u[k]);
      break;
    default: break;
  }
}

// ---- reductions.  Cross-lane steps are DPP / v_permlane*_swap (plain VALU), NOT __shfl_xor: hipcc lowers that to
//      ds_bpermute_b32, an LDS-queue instruction, and a young wave's LDS requests starve while an older LDS-saturating
//      kernel shares the CU.  Measured (tools/probes/coresidency_probe.py): next to the persistent weight-gradient kernel on
//      the side stream, a 32-workgroup kernel with ds_bpermute shuffles took 300-430 us (it ended when the weight gradient
//      did) against 8 us with none; every reduce / finalize kernel of the GroupNorm backward sat on the critical path.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {  // CTRL: quad_perm 0x00-0xFF, row_ror:n 0x120+n (within 16-lane rows)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// v(l) + v(l ^ 16) / v(l) + v(l ^ 32) in every lane.  v_permlane16_swap / v_permlane32_swap exchange, IN PLACE between their
// two registers, the odd rows / upper half of the first with the even rows / lower half of the second; fed the value and
// a copy of it, the first register ends up holding the even rows' (lower half's) values in both places and the second the
// odd rows' (upper half's).  Inline asm, not the builtin: given two copies of one value, hipcc (ROCm 7.2) folded the
// builtin's two results into one register and returned 2 * v (tools/probes/wave_sum_probe.hip).  The two v_nop are the
// wait states a VALU write of either operand needs before the swap reads it (LLVM gfx950 hazard rule; nothing is padded
// inside an asm string).
#define MEDNET_SWAP_PAIR(INSN, lo_, hi_) asm volatile("v_nop\n\tv_nop\n\t" INSN " %0, %1" : "+v"(lo_), "+v"(hi_))
__device__ __forceinline__ float xor16_sum(float v) {
  float a = v, b = v;
  MEDNET_SWAP_PAIR("v_permlane16_swap_b32", a, b);
  return a + b;
}
__device__ __forceinline__ float xor32_sum(float v) {
  float a = v, b = v;
  MEDNET_SWAP_PAIR("v_permlane32_swap_b32", a, b);
  return a + b;
}
__device__ __forceinline__ double xor16_sum(double v) {
  int alo = __double2loint(v), ahi = __double2hiint(v), blo = alo, bhi = ahi;
  MEDNET_SWAP_PAIR("v_permlane16_swap_b32", alo, blo);
  MEDNET_SWAP_PAIR("v_permlane16_swap_b32", ahi, bhi);
  return __hiloint2double(ahi, alo) + __hiloint2double(bhi, blo);
}
__device__ __forceinline__ double xor32_sum(double v) {
  int alo = __double2loint(v), ahi = __double2hiint(v), blo = alo, bhi = ahi;
  MEDNET_SWAP_PAIR("v_permlane32_swap_b32", alo, blo);
  MEDNET_SWAP_PAIR("v_permlane32_swap_b32", ahi, bhi);
  return __hiloint2double(ahi, alo) + __hiloint2double(bhi, blo);
}
// Sum over the lanes that agree in (lane % STRIDE), STRIDE a power of two <= 64; result in every lane of the class.
// (STRIDE = 1: the whole wave.)  Must be called with all 64 lanes active.
template <int STRIDE>
__device__ __forceinline__ float lane_class_sum(float v) {
  if (STRIDE <= 1) v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  if (STRIDE <= 2) v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  if (STRIDE <= 4) v += dpp_f32<0x124>(v);  // row_ror:4  } together: the four quads of a 16-lane row
  if (STRIDE <= 8) v += dpp_f32<0x128>(v);  // row_ror:8  }
  if (STRIDE <= 16) v = xor16_sum(v);
  if (STRIDE <= 32) v = xor32_sum(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return lane_class_sum<1>(v); }
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x124>(v);
  v += dpp_f64<0x128>(v);
  v = xor16_sum(v);
  return xor32_sum(v);
}
// Sum over a workgroup of NW waves; result valid in every thread. `scratch` holds >= NW floats.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += scratch[i];
  return t;
}

// Lanes of ONE wave hand data to each other through LDS without a barrier (the hardware executes a wave's LDS operations in
// order).  To the compiler that is a data race between threads: its alias analysis is per thread and it may move a lane's read
// above the same lane's write when it can prove THOSE two addresses differ (it did, in the ConvTranspose epilogue: the first
// read round came back stale).  This fence is no instruction; it pins the order of memory operations around it.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

}  // namespace mednet
