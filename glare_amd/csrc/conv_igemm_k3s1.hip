// Instantiations of conv_igemm_kernel (conv_igemm_kernel.h) for the 3x3 stride-1 convs: the LDS-staged 16-bit epilogue (EPI_FAST), the
// hi / lo epilogue form and the GroupNorm-prologue form; the planar and general epilogues live in conv_igemm_k3s1_planar.hip /
// _general.hip -- one translation unit per epilogue kind, so the build compiles them in parallel.
// (Measured and not kept as dispatch options: a 16 x 32 px tile on 8 waves; a 12 x 32 px tile on 6 waves, 20-25 % slower -- 6 waves map
// 2,2,1,1 onto the 4 SIMDs and the doubly-loaded SIMDs set the barrier pace; 8 x 32 px on 8 waves of 64 accumulators: +-2 %.)
#include "conv_igemm_kernel.h"

int glare_conv_launch_k3s1(const ConvParams& p, int tn, bool hilo, hipStream_t stream) {
  if (hilo) return tn == 128 ? launch<3, 1, 4, 2, 2, 2, 1, true>(p, stream) : GLARE_ERR_UNSUPPORTED;
  const int epi = conv_pick_epilogue(p, true);
  if (p.gn_coef) return (tn == 128 && epi == EPI_FAST) ? launch<3, 1, 4, 2, 2, 2, 1, false, true>(p, stream) : GLARE_ERR_UNSUPPORTED;
  if (epi == EPI_PLANAR) return glare_conv_launch_k3s1_planar(p, tn, stream);
  if (epi == EPI_GENERAL) return glare_conv_launch_k3s1_general(p, tn, stream);
  if (tn == 128) return launch<3, 1, 4, 2, 2, 2, 1>(p, stream);
  if (tn == 64) return launch<3, 1, 4, 1, 2, 2, 1>(p, stream);
  return launch<3, 1, 2, 1, 4, 1, 1>(p, stream);
}
