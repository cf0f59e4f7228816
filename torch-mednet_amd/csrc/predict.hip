// Inference path: the data movement around the forward kernels at prediction time, on the device.
//   grid_gather      midasmednet/dataset.py:349-390 (grid_patch_generator): np.pad + overlapping patch extraction
//   predict_assemble examples/predict.py:88-95 (argmax of the class logits, heat maps clipped to uint8) fused with
//                    GridPatchSampler.add_processed_batch (dataset.py:446-474): crop the overlap, clip at the volume
//                    edge, write into the result volume -- one pass over the logits, nothing intermediate in HBM.
#include "common.h"

namespace mednet {

// original index of padded index q (pad of `ov` in front), or -1 for a constant-mode zero
__device__ __forceinline__ int pad_src(int q, int ov, int n, int mode) {
  int i = q - ov;
  if (mode == MEDNET_PAD_CONSTANT) return (i >= 0 && i < n) ? i : -1;
  // numpy 'symmetric': ... 1 0 | 0 1 2 ... n-1 | n-1 n-2 ...  with period 2n
  const int period = 2 * n;
  i %= period;
  if (i < 0) i += period;
  return i < n ? i : period - 1 - i;
}

__global__ __launch_bounds__(256) void grid_gather_kernel(const float* __restrict__ vol, const int* __restrict__ pos,
                                                          float* __restrict__ out, int c, int d, int h, int w, int pd,
                                                          int ph, int pw, int ov0, int ov1, int ov2, int mode) {
  const size_t per = (size_t)c * pd * ph * pw;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per) return;
  const int b = blockIdx.y;
  const int x = (int)(i % pw);
  size_t r = i / pw;
  const int y = (int)(r % ph);
  r /= ph;
  const int z = (int)(r % pd);
  const int cc = (int)(r / pd);
  const int sz = pad_src(pos[b * 3] + z, ov0, d, mode), sy = pad_src(pos[b * 3 + 1] + y, ov1, h, mode),
            sx = pad_src(pos[b * 3 + 2] + x, ov2, w, mode);
  float v = 0.f;
  if (sz >= 0 && sy >= 0 && sx >= 0) v = vol[(((size_t)cc * d + sz) * h + sy) * w + sx];
  out[(size_t)b * per + i] = v;
}

// one thread per voxel of the cropped window of one patch: all output channels of that voxel
__global__ __launch_bounds__(256) void predict_assemble_kernel(const float* __restrict__ logits, const int* __restrict__ pos,
                                                               uint8_t* __restrict__ result, int nh, int ncls, int d, int h,
                                                               int w, int pd, int ph, int pw, int cs0, int cs1, int cs2,
                                                               int cz, int cy, int cx) {
  const size_t win = (size_t)cz * cy * cx;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= win) return;
  const int b = blockIdx.y;
  const int x = (int)(i % cx);
  size_t r = i / cx;
  const int y = (int)(r % cy);
  const int z = (int)(r / cy);
  const int gz = pos[b * 3] + z, gy = pos[b * 3 + 1] + y, gx = pos[b * 3 + 2] + x;
  if (gz >= d || gy >= h || gx >= w) return;  // the part of the last patches that hangs over the volume
  const size_t pvox = (size_t)pd * ph * pw;
  const float* src = logits + (size_t)b * (nh + ncls) * pvox + ((size_t)(z + cs0) * ph + (y + cs1)) * pw + (x + cs2);
  const size_t vox = (size_t)d * h * w, dst = ((size_t)gz * h + gy) * w + gx;
  for (int k = 0; k < nh; ++k) {
    const float v = fminf(fmaxf(src[(size_t)k * pvox], 0.f), 255.f);  // np.clip(.., 0, 255).astype(uint8): truncation
    result[(size_t)k * vox + dst] = (uint8_t)(int)v;
  }
  int best = 0;
  float bv = src[(size_t)nh * pvox];
  for (int k = 1; k < ncls; ++k) {  // argmax(softmax(x)) == argmax(x); first maximum on ties like torch.argmax
    const float v = src[(size_t)(nh + k) * pvox];
    if (v > bv) {
      bv = v;
      best = k;
    }
  }
  result[(size_t)nh * vox + dst] = (uint8_t)best;
}

// ---- training-patch crop: MedDataset.__getitem__'s `vol[:, i0:i1, j0:j1, k0:k1].astype(..)`
//      (dataset.py:313-331) from a device-resident volume straight into a slot / channel range of the batch tensors
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void crop_patches_kernel(const TS* __restrict__ src, const int* __restrict__ pos,
                                                           const int* __restrict__ slot, TD* __restrict__ out, int c, int d,
                                                           int h, int w, int c_total, int c_off, int pd, int ph, int pw) {
  const size_t per = (size_t)c * pd * ph * pw;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per) return;
  const int b = blockIdx.y;
  const int x = (int)(i % pw);
  size_t r = i / pw;
  const int y = (int)(r % ph);
  r /= ph;
  const int z = (int)(r % pd);
  const int cc = (int)(r / pd);
  const TS v = src[(((size_t)cc * d + pos[b * 3] + z) * h + pos[b * 3 + 1] + y) * w + pos[b * 3 + 2] + x];
  out[(((size_t)slot[b] * c_total + c_off + cc) * pd + z) * ph * pw + (size_t)y * pw + x] = (TD)v;
}

}  // namespace mednet

using namespace mednet;

extern "C" int mednet_crop_patches(const void* src, int src_dtype, const int* pos, const int* slot, int count, void* out,
                                   int dst_dtype, int c, int d, int h, int w, int c_total, int c_off, int pd, int ph, int pw,
                                   mednet_stream stream) {
  MEDNET_REQUIRE(count > 0 && c > 0 && pd > 0 && ph > 0 && pw > 0 && pd <= d && ph <= h && pw <= w && c_off >= 0 &&
                     c_off + c <= c_total,
                 MEDNET_E_SHAPE, "crop_patches: bad shape (patch %dx%dx%d in %dx%dx%d, channels %d+%d of %d)", pd, ph, pw, d, h, w,
                 c_off, c, c_total);
  const size_t per = (size_t)c * pd * ph * pw;
  const dim3 grid((unsigned)((per + 255) / 256), count);
  hipStream_t s = (hipStream_t)stream;
#define CP_GO(TS_, TD_) hipLaunchKernelGGL((crop_patches_kernel<TS_, TD_>), grid, dim3(256), 0, s, (const TS_*)src, pos, slot, (TD_*)out, c, d, h, w, c_total, c_off, pd, ph, pw)
  if (src_dtype == MEDNET_F16 && dst_dtype == MEDNET_F32) CP_GO(_Float16, float);
  else if (src_dtype == MEDNET_F32 && dst_dtype == MEDNET_F32) CP_GO(float, float);
  else if (src_dtype == MEDNET_U8 && dst_dtype == MEDNET_U8) CP_GO(uint8_t, uint8_t);
  else return fail(MEDNET_E_DTYPE, "crop_patches: %d -> %d (supported: f16->f32, f32->f32, u8->u8)", src_dtype, dst_dtype);
#undef CP_GO
  return check_launch("crop_patches");
}

extern "C" int mednet_grid_gather(const float* volume, const int* pos, float* out, int batch, int c, int d, int h, int w,
                                  int pd, int ph, int pw, int ov0, int ov1, int ov2, int pad_mode, mednet_stream stream) {
  MEDNET_REQUIRE(batch > 0 && c > 0 && d > 0 && h > 0 && w > 0 && pd > 0 && ph > 0 && pw > 0, MEDNET_E_SHAPE, "grid_gather: bad shape");
  MEDNET_REQUIRE(pad_mode == MEDNET_PAD_CONSTANT || pad_mode == MEDNET_PAD_SYMMETRIC, MEDNET_E_UNSUPPORTED,
                 "grid_gather: pad mode %d (supported: constant, symmetric)", pad_mode);
  const size_t per = (size_t)c * pd * ph * pw;
  hipLaunchKernelGGL(grid_gather_kernel, dim3((unsigned)((per + 255) / 256), batch), dim3(256), 0, (hipStream_t)stream, volume,
                     pos, out, c, d, h, w, pd, ph, pw, ov0, ov1, ov2, pad_mode);
  return check_launch("grid_gather");
}

extern "C" int mednet_predict_assemble(const float* logits, const int* pos, uint8_t* result, int batch, int num_heatmaps,
                                       int num_classes, int d, int h, int w, int pd, int ph, int pw, int crop_start0,
                                       int crop_start1, int crop_start2, int crop_d, int crop_h, int crop_w,
                                       mednet_stream stream) {
  MEDNET_REQUIRE(batch > 0 && num_heatmaps >= 0 && num_classes >= 1, MEDNET_E_SHAPE, "predict_assemble: bad channel counts");
  MEDNET_REQUIRE(crop_start0 >= 0 && crop_start1 >= 0 && crop_start2 >= 0 && crop_start0 + crop_d <= pd &&
                     crop_start1 + crop_h <= ph && crop_start2 + crop_w <= pw,
                 MEDNET_E_SHAPE, "predict_assemble: crop window outside the patch");
  const size_t win = (size_t)crop_d * crop_h * crop_w;
  if (win == 0) return MEDNET_OK;  // (the reference's slicing yields an empty window when an overlap is 0)
  hipLaunchKernelGGL(predict_assemble_kernel, dim3((unsigned)((win + 255) / 256), batch), dim3(256), 0, (hipStream_t)stream,
                     logits, pos, result, num_heatmaps, num_classes, d, h, w, pd, ph, pw, crop_start0, crop_start1,
                     crop_start2, crop_d, crop_h, crop_w);
  return check_launch("predict_assemble");
}
