// Instantiations of conv_igemm_kernel (conv_igemm_kernel.h) for the 3x3 stride-2 convs (Downsample, encoder_decoder.py:68-75): the
// LDS-staged 16-bit epilogue and the hi / lo form; the general epilogue lives in conv_igemm_general.hip.
#include "conv_igemm_kernel.h"

int glare_conv_launch_k3s2(const ConvParams& p, int tn, bool hilo, hipStream_t stream) {
  if (hilo) return tn == 128 ? launch<3, 2, 4, 2, 2, 2, 1, true>(p, stream) : GLARE_ERR_UNSUPPORTED;
  if (conv_pick_epilogue(p, false) != EPI_FAST) return glare_conv_launch_k3s2_general(p, tn, stream);
  if (tn == 128) return launch<3, 2, 4, 2, 2, 2, 1>(p, stream);
  if (tn == 64) return launch<3, 2, 4, 1, 2, 2, 1>(p, stream);
  return launch<3, 2, 2, 1, 4, 1, 1>(p, stream);
}
