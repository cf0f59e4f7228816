// Which bf16 MFMA shape does the chip clock higher under load?  Two bare loops with the operand traffic of the 32 -> 32
// convolution (A = weights in registers, B = one ds_read_b128 per 32 KFLOP... per 1 KB of B), one wave per SIMD, random data:
//   shape 0: v_mfma_f32_32x32x16_bf16, 4 MFMAs + 4 ds_read_b128 per step
//   shape 1: v_mfma_f32_16x16x32_bf16, 16 MFMAs + 8 ds_read_b128 per step (each B fragment feeds two MFMAs)
// Shape 1 does twice the FLOPs per step (16 x 16384 against 4 x 32768) and runs half as many steps: the same LDS bytes per
// FLOP and the same total FLOPs; prints wall time and TFLOP/s of both.  Measured on two boxes of the round-2 pool (profiles/
// r02_mfma_shape_probe.log): 1.17-1.32 PFLOP/s, the two shapes within 2 % of each other -- the clock the chip holds under this
// load, not the issue stream, bounds a bf16 MFMA loop fed from LDS at about half of the 2.5 PFLOP/s nominal peak, and the
// shape does not move it here.
// build: hipcc -O3 --offload-arch=gfx950 mfma_shape_probe.hip -o mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void probe(const u32x4* src, float* out, int steps) {
  extern __shared__ u32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) lds[i] = src[i + (blockIdx.x & 7) * 8192];  // 128 KB of random bf16
  __syncthreads();
  bf16x8 w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = __builtin_bit_cast(bf16x8, src[tid + 256 * i]);
  const u32x4* p = lds + (tid >> 6) * 1024 + lane;
  if constexpr (SHAPE == 0) {
    f32x16 acc[4] = {};
    bf16x8 b[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) b[0][t] = __builtin_bit_cast(bf16x8, p[t * 64]);
    for (int s = 0; s < steps; s += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          b[u ^ 1][t] = __builtin_bit_cast(bf16x8, p[((s + u + 1) & 3) * 256 + t * 64]);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(u * 4 + t) & 7], b[u][t], acc[t], 0, 0, 0);
        }
      }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * 256 + tid] = r;
  } else if constexpr (SHAPE == 2) {
    // 32x32x16 with NO LDS traffic in the loop: the four B fragments are read once and reused (what the matrix pipe alone sustains)
    f32x16 acc[4] = {};
    bf16x8 b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = __builtin_bit_cast(bf16x8, p[t * 64]);
    for (int s = 0; s < steps; s += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(u * 4 + t) & 7], b[t], acc[t], 0, 0, 0);
      }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * 256 + tid] = r;
  } else if constexpr (SHAPE == 5) {
    // the general convolution kernel's tiling: per tap step 4 N-tiles x 1 channel block = 4 B reads + 1 A read per 4 MFMAs
    f32x16 acc[4] = {};
    bf16x8 b[2][4], a[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) b[0][t] = __builtin_bit_cast(bf16x8, p[t * 64]);
    a[0] = __builtin_bit_cast(bf16x8, p[4096]);
    for (int s = 0; s < steps; s += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        a[u ^ 1] = __builtin_bit_cast(bf16x8, p[4096 + ((s + u + 1) & 7) * 64]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          b[u ^ 1][t] = __builtin_bit_cast(bf16x8, p[((s + u + 1) & 3) * 256 + t * 64]);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[u][t], acc[t], 0, 0, 0);
        }
      }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * 256 + tid] = r;
  } else if constexpr (SHAPE == 6) {
    // two channel blocks per wave: 4 N-tiles x 2 blocks = 4 B reads + 2 A reads per 8 MFMAs (128 accumulator registers)
    f32x16 acc[8] = {};
    bf16x8 b[2][4], a[2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) b[0][t] = __builtin_bit_cast(bf16x8, p[t * 64]);
    a[0][0] = __builtin_bit_cast(bf16x8, p[4096]);
    a[0][1] = __builtin_bit_cast(bf16x8, p[4160]);
    for (int s = 0; s < steps; s += 4) {  // a step of this shape = 8 MFMAs = two steps of the others
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        a[u ^ 1][0] = __builtin_bit_cast(bf16x8, p[4096 + ((s + u + 1) & 7) * 128]);
        a[u ^ 1][1] = __builtin_bit_cast(bf16x8, p[4096 + ((s + u + 1) & 7) * 128 + 64]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          b[u ^ 1][t] = __builtin_bit_cast(bf16x8, p[((s / 2 + u + 1) & 3) * 256 + t * 64]);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u][0], b[u][t], acc[t], 0, 0, 0);
          acc[4 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u][1], b[u][t], acc[4 + t], 0, 0, 0);
        }
      }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * 256 + tid] = r;
  } else if constexpr (SHAPE == 7) {
    // round 6: y-reuse by REGISTER RENAMING.  An N-tile = the two x-rows (t, t + 4) of the wave's z-plane (not (2t, 2t + 1)): the tap
    // shifted by ky then needs rows (t + ky, t + ky + 4) -- the fragment of tile t + ky.  Six fragments (row pairs (a, a + 4), a = 0..5;
    // the last two reach into the halo) serve the three ky taps of all four tiles: 6 ds_read_b128 per 12 MFMAs, no shuffle at all.
    f32x16 acc[4] = {};
    bf16x8 f[2][6];
#pragma unroll
    for (int a = 0; a < 6; ++a) f[0][a] = __builtin_bit_cast(bf16x8, p[a * 64]);
    for (int s = 0; s < steps; s += 6) {  // 6 steps of 4 MFMAs = 2 groups of 12 MFMAs, each with 6 reads for the next group
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int m = ky * 4 + t;
            if (m % 2 == 0) f[u ^ 1][m / 2] = __builtin_bit_cast(bf16x8, p[((s / 3 + u + 1) & 3) * 384 + (m / 2) * 64]);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(ky * 4 + t + u) & 7], f[u][t + ky], acc[t], 0, 0, 0);
          }
      }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * 256 + tid] = r;
  } else if constexpr (SHAPE == 8) {
    // ... + x-reuse: the kx = 1 fragments made from the kx = 0 and kx = 2 ones (row_shl:1 of the first; the row's last lanes from
    // row_shr:1 of the second, bank 3 only): 12 reads + 48 DPP moves per 36 MFMAs (0.33 reads per MFMA)
    f32x16 acc[4] = {};
    bf16x8 f0[2][6], f2[2][6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      f0[0][a] = __builtin_bit_cast(bf16x8, p[a * 64]);
      f2[0][a] = __builtin_bit_cast(bf16x8, p[384 + a * 64]);
    }
    for (int s = 0; s < steps; s += 18) {  // 18 steps of 4 MFMAs = 2 groups of 36 MFMAs
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bf16x8 f1[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const u32x4 q0 = __builtin_bit_cast(u32x4, f0[u][a]), q2 = __builtin_bit_cast(u32x4, f2[u][a]);
          u32x4 r1;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int sh = __builtin_amdgcn_update_dpp(0, (int)q0[i], 0x101, 0xF, 0xF, false);       // row_shl:1: lane x <- x + 1
            r1[i] = (unsigned)__builtin_amdgcn_update_dpp(sh, (int)q2[i], 0x111, 0xF, 0x8, false);  // row_shr:1, lanes 12..15 only
          }
          f1[a] = __builtin_bit_cast(bf16x8, r1);
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int m = (kx * 3 + ky) * 4 + t;  // 0..35: one read every third MFMA
              if (m % 3 == 0) {
                const int k = m / 3;  // 0..11
                if (k < 6) f0[u ^ 1][k] = __builtin_bit_cast(bf16x8, p[((s / 9 + u + 1) & 3) * 768 + k * 64]);
                else f2[u ^ 1][k - 6] = __builtin_bit_cast(bf16x8, p[((s / 9 + u + 1) & 3) * 768 + k * 64]);
              }
              const bf16x8 b = kx == 0 ? f0[u][t + ky] : (kx == 1 ? f1[t + ky] : f2[u][t + ky]);
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(m + u) & 7], b, acc[t], 0, 0, 0);
            }
      }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * 256 + tid] = r;
  } else if constexpr (SHAPE == 3 || SHAPE == 4) {
    // 32x32x16, ONE ds_read_b128 per THREE MFMAs: the x-shifted taps of a 3x3x3 convolution read the same voxel rows moved by one
    // lane, so two of three B operands can be made from the first with 4 DPP row shifts each instead of a 1 KB LDS read
    // (the boundary lane's value is not patched here: timing only)
    f32x16 acc[4] = {};
    bf16x8 b[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) b[0][t] = __builtin_bit_cast(bf16x8, p[t * 64]);
    // lanes without a source inside their 16-lane row keep `edge` (bound_ctrl off): the halo voxel needs no merge instruction
    auto shift = [](bf16x8 v, bf16x8 edge, bool right) {
      u32x4 q = __builtin_bit_cast(u32x4, v), e = __builtin_bit_cast(u32x4, edge), r;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        r[i] = right ? (unsigned)__builtin_amdgcn_update_dpp((int)e[i], (int)q[i], 0x111, 0xF, 0xF, false)   // row_shr:1
                     : (unsigned)__builtin_amdgcn_update_dpp((int)e[i], (int)q[i], 0x101, 0xF, 0xF, false);  // row_shl:1
      return __builtin_bit_cast(bf16x8, r);
    };
    for (int s = 0; s < steps; s += 6) {  // 6 steps of 4 MFMAs = 2 groups of 3 steps sharing one set of reads
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          b[u ^ 1][t] = __builtin_bit_cast(bf16x8, p[((s + u + 1) & 3) * 256 + t * 64]);
          bf16x8 hl = b[u][t], hr = b[u][t];
          if constexpr (SHAPE == 4) {  // the voxel beyond either end of a 16-voxel row: lanes 0 / 15 of every row read it (8 of 64 lanes)
            if ((lane & 15) == 0) hl = __builtin_bit_cast(bf16x8, p[((s + u + 2) & 3) * 256 + t * 64 + 1]);
            if ((lane & 15) == 15) hr = __builtin_bit_cast(bf16x8, p[((s + u + 3) & 3) * 256 + t * 64 - 1]);
          }
          const bf16x8 bl = shift(b[u][t], hr, false), br = shift(b[u][t], hl, true);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(u * 4 + t) & 7], bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(u * 4 + t + 1) & 7], b[u][t], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(u * 4 + t + 2) & 7], br, acc[t], 0, 0, 0);
        }
      }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * 256 + tid] = r;
  } else {
    f32x4 acc[16] = {};
    bf16x8 b[2][8];
#pragma unroll
    for (int t = 0; t < 8; ++t) b[0][t] = __builtin_bit_cast(bf16x8, p[t * 64]);
    for (int s = 0; s < steps; s += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          b[u ^ 1][t] = __builtin_bit_cast(bf16x8, p[((s + u + 1) & 1) * 512 + t * 64]);
          acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[(2 * t) & 7], b[u][t], acc[2 * t], 0, 0, 0);
          acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[(2 * t + 1) & 7], b[u][t], acc[2 * t + 1], 0, 0, 0);
        }
      }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) r += acc[t][i];
    out[blockIdx.x * 256 + tid] = r;
  }
}

int main() {
  const size_t n = 8 * 8192;
  std::vector<unsigned> h(n * 4);
  srand(1);
  for (auto& v : h) {  // random bf16 pairs in [-2, 2): random mantissas and signs, exponents near 0
    unsigned a = (rand() & 0x807F) | (((rand() % 3) + 126) << 7), b = (rand() & 0x807F) | (((rand() % 3) + 126) << 7);
    v = a | (b << 16);
  }
  u32x4* d;
  float* o;
  hipMalloc(&d, n * 16);
  hipMalloc(&o, 256 * 256 * 4);
  hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)probe<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)probe<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)probe<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int steps0 = 39996;  // shape 0 steps (a multiple of 6 for shapes 3 and 7 and of 18 for shape 8); shape 1 runs half as many of twice the FLOPs
  const char* names[9] = {"32x32x16, 1 ds_read_b128 per MFMA", "16x16x32, 1 ds_read_b128 per 2 MFMAs", "32x32x16, no LDS reads in the loop",
                          "32x32x16, 1 ds_read_b128 + 8 DPP shifts per 3 MFMAs",
                          "... + the two halo columns by 4-lane ds_read_b128",
                          "4 N-tiles x 1 block: 4 B + 1 A reads per 4 MFMAs", "4 N-tiles x 2 blocks: 4 B + 2 A reads per 8 MFMAs",
                          "row pairs (t, t+4): 6 reads per 12 MFMAs, no shuffles", "... + kx = 1 by DPP: 12 reads + 48 DPP per 36 MFMAs"};
  for (int rep = 0; rep < 3; ++rep)
    for (int shape = 0; shape < 9; ++shape) {
      for (int warm = 0; warm < 2; ++warm) {
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 131072, 0, d, o, steps0);
        else if (shape == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 131072, 0, d, o, steps0 / 2);
        else if (shape == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 131072, 0, d, o, steps0);
        else if (shape == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 131072, 0, d, o, steps0);
        else if (shape == 4) hipLaunchKernelGGL(probe<4>, dim3(256), dim3(256), 131072, 0, d, o, steps0);
        else if (shape == 5) hipLaunchKernelGGL(probe<5>, dim3(256), dim3(256), 131072, 0, d, o, steps0);
        else if (shape == 6) hipLaunchKernelGGL(probe<6>, dim3(256), dim3(256), 131072, 0, d, o, steps0);
        else if (shape == 7) hipLaunchKernelGGL(probe<7>, dim3(256), dim3(256), 131072, 0, d, o, steps0);
        else hipLaunchKernelGGL(probe<8>, dim3(256), dim3(256), 131072, 0, d, o, steps0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = 256.0 * 4 * steps0 * 4 * 2.0 * 32 * 32 * 16;  // the same for every shape
      printf("%-52s %8.3f ms  %7.1f TFLOP/s  (%.1f cycles per 32768 FLOP per SIMD at 2.4 GHz; 32 = nominal peak)\n",
             names[shape], ms, flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (steps0 * 4.0));
    }
  return 0;
}
