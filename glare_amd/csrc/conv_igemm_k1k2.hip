// Instantiations of conv_igemm_kernel (conv_igemm_kernel.h) for the 1x1 convs and the sub-pixel upsample form (KS = 2), LDS-staged 16-bit
// epilogue and hi / lo form; the 1x1 convs' general epilogue lives in conv_igemm_general.hip.
#include "conv_igemm_kernel.h"

int glare_conv_launch_k1(const ConvParams& p, int tn, bool hilo, hipStream_t stream) {
  if (hilo) {
    if (tn == 128) return launch<1, 1, 4, 2, 2, 2, 2, true>(p, stream);
    return tn == 64 ? launch<1, 1, 4, 1, 2, 2, 2, true>(p, stream) : GLARE_ERR_UNSUPPORTED;
  }
  if (conv_pick_epilogue(p, false) != EPI_FAST) return glare_conv_launch_k1_general(p, tn, stream);
  if (tn == 128) return launch<1, 1, 4, 2, 2, 2, 2>(p, stream);
  if (tn == 64) return launch<1, 1, 4, 1, 2, 2, 2>(p, stream);
  return launch<1, 1, 2, 1, 4, 1, 2>(p, stream);
}

int glare_conv_launch_k2(const ConvParams& p, int tn, hipStream_t stream) {   // the interleaved scatter exists in the slab epilogue only
  if (conv_pick_epilogue(p, false) != EPI_FAST) return GLARE_ERR_UNSUPPORTED;
  if (tn == 128) return launch<2, 1, 4, 2, 2, 2, 1>(p, stream);
  if (tn == 64) return launch<2, 1, 4, 1, 2, 2, 1>(p, stream);
  return launch<2, 1, 2, 1, 4, 1, 1>(p, stream);
}
