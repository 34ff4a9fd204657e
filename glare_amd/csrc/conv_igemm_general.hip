// conv_igemm_kernel with the general epilogue (EPI_GENERAL) for the 3x3 stride-2 and the 1x1 convs: its own translation unit (see
// conv_igemm_k3s1.hip).
#include "conv_igemm_kernel.h"

int glare_conv_launch_k3s2_general(const ConvParams& p, int tn, hipStream_t stream) {
  if (tn == 128) return launch<3, 2, 4, 2, 2, 2, 1, false, false, EPI_GENERAL>(p, stream);
  if (tn == 64) return launch<3, 2, 4, 1, 2, 2, 1, false, false, EPI_GENERAL>(p, stream);
  return launch<3, 2, 2, 1, 4, 1, 1, false, false, EPI_GENERAL>(p, stream);
}

int glare_conv_launch_k1_general(const ConvParams& p, int tn, hipStream_t stream) {
  if (tn == 128) return launch<1, 1, 4, 2, 2, 2, 2, false, false, EPI_GENERAL>(p, stream);
  if (tn == 64) return launch<1, 1, 4, 1, 2, 2, 2, false, false, EPI_GENERAL>(p, stream);
  return launch<1, 1, 2, 1, 4, 1, 2, false, false, EPI_GENERAL>(p, stream);
}
