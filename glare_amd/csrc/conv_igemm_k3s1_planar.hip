// conv_igemm_kernel, 3x3 stride 1, fp32-planar epilogue through an LDS slab (EPI_PLANAR: the DCN's offset / mask-logit planes,
// deform_conv.py:357-364; the AFT decoder's final 128 -> 3 conv): its own translation unit (see conv_igemm_k3s1.hip).
#include "conv_igemm_kernel.h"

int glare_conv_launch_k3s1_planar(const ConvParams& p, int tn, hipStream_t stream) {
  if (tn == 128) return launch<3, 1, 4, 2, 2, 2, 1, false, false, EPI_PLANAR>(p, stream);
  if (tn == 64) return launch<3, 1, 4, 1, 2, 2, 1, false, false, EPI_PLANAR>(p, stream);
  return launch<3, 1, 2, 1, 4, 1, 1, false, false, EPI_PLANAR>(p, stream);
}
