// bf16 MFMA convolution kernels (placeholder translation unit: filled in by the MFMA milestone).
#include "conv.h"

namespace mednet {

PackLayout pack_layout(int cin, int cout, int ksize) {
  PackLayout L;
  L.taps = ksize * ksize * ksize;
  const size_t f32 = align_up((size_t)L.taps * cin * cout * sizeof(float), 256);
  L.f32_fwd = 0;
  L.f32_bwd = f32;
  L.mfma_bytes = 0;
  L.mfma_fwd = 2 * f32;
  L.mfma_bwd = 2 * f32;
  L.total = 2 * f32 + 2 * L.mfma_bytes + 256;
  return L;
}

bool conv_mfma_supported(int, int, int, int, int, int, int, bool) { return false; }
int launch_conv_mfma(const void*, const void*, void*, int, int, int, int, int, int, int, int, hipStream_t) {
  return fail(MEDNET_E_UNSUPPORTED, "conv_mfma: not built");
}
int launch_pack_mfma(const float*, void*, void*, int, int, int, int, hipStream_t) { return MEDNET_OK; }
bool wgrad_mfma_supported(int, int, int, int, int, int, int) { return false; }
size_t wgrad_mfma_ws_bytes(int, int, int, int, int, int, int) { return 0; }
int launch_wgrad_mfma(const void*, const void*, float*, int, int, int, int, int, int, int, void*, size_t, hipStream_t) {
  return fail(MEDNET_E_UNSUPPORTED, "wgrad_mfma: not built");
}

}  // namespace mednet
