// Instantiations of conv_igemm_kernel (conv_igemm_kernel.h) for the 3x3 stride-1 convs (and their hi / lo epilogue form): its own translation unit so the build compiles the
// kernel families in parallel.
#include "conv_igemm_kernel.h"

int glare_conv_launch_k3s1(const ConvParams& p, int tn, bool hilo, hipStream_t stream) {
  if (hilo) return (tn == 128 && !CONV_TILE16) ? launch<3, 1, 4, 2, 2, 2, 1, true>(p, stream) : GLARE_ERR_UNSUPPORTED;
  if (p.gn_coef) return (tn == 128 && !CONV_TILE16) ? launch<3, 1, 4, 2, 2, 2, 1, false, true>(p, stream) : GLARE_ERR_UNSUPPORTED;
  if (tn == 128) {
    if (3 == 3 && 1 == 1 && CONV_TILE16) return launch<3, 1, 4, 2, 4, 2, 1>(p, stream);   /* 16 x 32 px, 8 waves */
    /* a 12 x 32 px tile on 6 waves (fewer weight DMAs per MFMA) measured 20-25 % SLOWER: 6 waves map 2,2,1,1 onto the 4 SIMDs and */
    /* the doubly-loaded SIMDs set the barrier pace; keep wave counts multiples of 4 */
#ifdef CONV_W8
    return launch<3, 1, 2, 2, 4, 2, 1>(p, stream);   /* experiment: 8 x 32 px on 8 waves (64 accumulators each), 2 workgroups / CU */
#endif
    return launch<3, 1, 4, 2, 2, 2, 1>(p, stream);
  }
  if (tn == 64) return launch<3, 1, 4, 1, 2, 2, 1>(p, stream);
  return launch<3, 1, 2, 1, 4, 1, 1>(p, stream);
}
