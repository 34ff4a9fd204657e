// Instantiations of conv_igemm_kernel (conv_igemm_kernel.h) for the 1x1 convs and the sub-pixel upsample form (KS = 2): its own translation unit so the build compiles the
// kernel families in parallel.
#include "conv_igemm_kernel.h"

int glare_conv_launch_k1(const ConvParams& p, int tn, bool hilo, hipStream_t stream) {
  if (hilo) {
    if (tn == 128) return launch<1, 1, 4, 2, 2, 2, 2, true>(p, stream);
    return tn == 64 ? launch<1, 1, 4, 1, 2, 2, 2, true>(p, stream) : GLARE_ERR_UNSUPPORTED;
  }
  if (tn == 128) {
    if (1 == 3 && 1 == 1 && CONV_TILE16) return launch<1, 1, 4, 2, 4, 2, 2>(p, stream);   /* 16 x 32 px, 8 waves */
    /* a 12 x 32 px tile on 6 waves (fewer weight DMAs per MFMA) measured 20-25 % SLOWER: 6 waves map 2,2,1,1 onto the 4 SIMDs and */
    /* the doubly-loaded SIMDs set the barrier pace; keep wave counts multiples of 4 */
    return launch<1, 1, 4, 2, 2, 2, 2>(p, stream);
  }
  if (tn == 64) return launch<1, 1, 4, 1, 2, 2, 2>(p, stream);
  return launch<1, 1, 2, 1, 4, 1, 2>(p, stream);
}

int glare_conv_launch_k2(const ConvParams& p, int tn, hipStream_t stream) {
  if (tn == 128) {
    if (2 == 3 && 1 == 1 && CONV_TILE16) return launch<2, 1, 4, 2, 4, 2, 1>(p, stream);   /* 16 x 32 px, 8 waves */
    /* a 12 x 32 px tile on 6 waves (fewer weight DMAs per MFMA) measured 20-25 % SLOWER: 6 waves map 2,2,1,1 onto the 4 SIMDs and */
    /* the doubly-loaded SIMDs set the barrier pace; keep wave counts multiples of 4 */
    return launch<2, 1, 4, 2, 2, 2, 1>(p, stream);
  }
  if (tn == 64) return launch<2, 1, 4, 1, 2, 2, 1>(p, stream);
  return launch<2, 1, 2, 1, 4, 1, 1>(p, stream);
}
