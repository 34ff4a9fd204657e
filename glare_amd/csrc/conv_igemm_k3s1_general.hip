// conv_igemm_kernel, 3x3 stride 1, general epilogue (EPI_GENERAL: fp32 NHWC, 16-bit planes, odd pitches, sigmoid / swish on the
// accumulators): its own translation unit (see conv_igemm_k3s1.hip).
#include "conv_igemm_kernel.h"

int glare_conv_launch_k3s1_general(const ConvParams& p, int tn, hipStream_t stream) {
  if (tn == 128) return launch<3, 1, 4, 2, 2, 2, 1, false, false, EPI_GENERAL>(p, stream);
  if (tn == 64) return launch<3, 1, 4, 1, 2, 2, 1, false, false, EPI_GENERAL>(p, stream);
  return launch<3, 1, 2, 1, 4, 1, 1, false, false, EPI_GENERAL>(p, stream);
}
