/*
 * TEST INFRASTRUCTURE -- plain-C restatement of the codebook retrieval of
 * VectorQuantizer2.forward (/root/reference/code/models/modules/quantize.py:276-285):
 *     d = sum(z^2, dim=1, keepdim) + sum(e^2, dim=1) - 2 * einsum('bd,dn->bn', z, e^T)
 *     idx = argmin(d, dim=1)        (first minimum)
 *     z_q = embedding(idx)
 * with the exact fp32 operation order that reproduces torch-CPU's `d` bit for bit on the build
 * host (hard part 2; re-verified by tests/golden/make_golden.py, which
 * stores torch's own `d` rows and indices as the golden vector):
 *     zz  = (z0*z0 + z1*z1) + z2*z2                      no fused multiply-add
 *     ee  = (e0*e0 + e1*e1) + e2*e2                      no fused multiply-add
 *     dot = fma(z2, e2, fma(z1, e1, z0*e0))
 *     d   = (zz + ee) - 2*dot
 * Compile with -ffp-contract=off (oracle/Makefile); fmaf() is the only fused operation.
 * Never linked into the product.
 */
#include <math.h>
#include <stdint.h>

int vq_ref_nearest(const float* z, const float* cb, long long n_tokens, int n_codes, int64_t* idx, float* zq,
                   float* d_out /* optional [n_tokens][n_codes] */) {
  for (long long t = 0; t < n_tokens; ++t) {
    const float z0 = z[t * 3], z1 = z[t * 3 + 1], z2 = z[t * 3 + 2];
    const float zz = (z0 * z0 + z1 * z1) + z2 * z2;
    float best = INFINITY;
    int bi = 0;
    for (int c = 0; c < n_codes; ++c) {
      const float e0 = cb[c * 3], e1 = cb[c * 3 + 1], e2 = cb[c * 3 + 2];
      const float ee = (e0 * e0 + e1 * e1) + e2 * e2;
      const float dot = fmaf(z2, e2, fmaf(z1, e1, z0 * e0));
      const float d = (zz + ee) - 2.0f * dot;
      if (d_out) d_out[t * (long long)n_codes + c] = d;
      if (d < best) {
        best = d;
        bi = c;
      }
    }
    idx[t] = bi;
    if (zq) {
      zq[t * 3] = cb[bi * 3];
      zq[t * 3 + 1] = cb[bi * 3 + 1];
      zq[t * 3 + 2] = cb[bi * 3 + 2];
    }
  }
  return 0;
}
