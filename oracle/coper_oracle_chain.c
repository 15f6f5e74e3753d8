/* TEST INFRASTRUCTURE ONLY -- bit-exact restatement of the score arithmetic of libcoper_hip.so.
 *
 * The reference computes predictions_all = matmul(h, ent_emb^T) + pred_bias in fp32
 * (CoPER_ConvE/qa_cpg/models.py:434-437) with whatever summation order TF's matmul picks; any fp32
 * order is "the reference result" to within rounding.  The HIP kernels pick ONE order -- the k-ordered
 * fma chain of v_mfma_f32_32x32x2_f32, k-pairs (k, k+4), started from pred_bias -- and this file
 * restates exactly that chain with C fmaf (single rounding), so tests can demand bit equality of the
 * logits and therefore of the integer ranks (coper_amd/csrc/kernels_score.hip header).
 *
 * Also: a plain-C filtered rank count (metrics.py:44-50 closed form) fast enough for full-size checks.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__) && defined(__GNUC__)
#define CLONES __attribute__((target_clones("fma", "default")))
#else
#define CLONES
#endif

CLONES
void oracle_score_chain(const float* h, const float* E, const float* bias, int64_t B, int64_t N, int d,
                        float* out /* [B, N] */) {
  int KS = (d + 7) / 8;
  for (int64_t b = 0; b < B; ++b) {
    const float* hr = h + b * d;
    for (int64_t e = 0; e < N; ++e) {
      const float* er = E + e * d;
      float s = bias[e];
      for (int ks = 0; ks < KS; ++ks)
        for (int t = 0; t < 4; ++t) {
          int k0 = 8 * ks + t, k1 = k0 + 4;
          if (k0 < d) s = fmaf(er[k0], hr[k0], s);
          if (k1 < d) s = fmaf(er[k1], hr[k1], s);
        }
      out[b * N + e] = s;
    }
  }
}

/* n_greater / n_equal over unfiltered j != e2 (metrics.py:44-50 closed form); idx sorted or not. */
void oracle_rank_counts(const float* pred, const int64_t* e2, const int64_t* indptr, const int64_t* idx, int64_t B,
                        int64_t N, int64_t* ng, int64_t* ne) {
  unsigned char* mask = (unsigned char*)malloc((size_t)N);
  for (int64_t b = 0; b < B; ++b) {
    memset(mask, 0, (size_t)N);
    for (int64_t i = indptr[b]; i < indptr[b + 1]; ++i)
      if (idx[i] >= 0 && idx[i] < N) mask[idx[i]] = 1;
    mask[e2[b]] = 1;
    const float* row = pred + b * N;
    float t = row[e2[b]];
    int64_t g = 0, q = 0;
    for (int64_t j = 0; j < N; ++j) {
      if (mask[j]) continue;
      g += row[j] > t;
      q += row[j] == t;
    }
    ng[b] = g;
    ne[b] = q;
  }
  free(mask);
}
