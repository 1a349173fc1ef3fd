// tests/hostcheck/hostcheck.cpp — host build of neurec_amd/csrc/nr_core.h.
//
// TEST HARNESS ONLY: compiles the per-thread arithmetic that the HIP kernels
// inline (heap emulation, metric formulas, sampler RNG/permutation, Adam) with
// g++ so that CPU tests can exercise the very same source against oracle/
// without a GPU.  The product never loads this library.
#include <cstdint>
#include <vector>
#include "nr_core.h"

extern "C" {

void hc_partial_sort_copy(const float* score, int n, int sort_len, int* out_idx) {
  int m = sort_len < n ? sort_len : n;
  std::vector<float> hv(m);
  nr::partial_sort_copy_emul(score, n, sort_len, hv.data(), out_idx);
}

void hc_metric(int metric_id, const unsigned char* hits, int K, int T, float* out) {
  std::vector<double> tbl(K);
  for (int i = 0; i < K; ++i) tbl[i] = 1.0 / log2((double)(unsigned)(i + 2));
  nr::metric_eval(metric_id, [hits](int i) { return hits[i] != 0; }, K, T, tbl.data(), out);
}

uint64_t hc_pack_key(float s, uint32_t idx) { return nr::pack_key(s, idx); }
float hc_key_score(uint64_t k) { return nr::unorder_f32(nr::key_order(k)); }
uint32_t hc_key_index(uint64_t k) { return nr::key_index(k); }

uint64_t hc_permute_index(uint64_t i, uint64_t n, uint64_t key) { return nr::permute_index(i, n, key); }

int32_t hc_draw_negative(uint64_t seed, uint64_t stream, uint64_t counter, int32_t high,
                         const int32_t* excl, int n_excl) {
  nr::XorShift64s g;
  g.seed(seed, stream, counter);
  return nr::draw_negative(g, high, excl, n_excl);
}
int32_t hc_nth_allowed(int32_t r, const int32_t* excl, int n_excl) { return nr::nth_allowed(r, excl, n_excl); }

float hc_softplus(float z) { return nr::tf_softplus(z); }
float hc_bpr_loss(float x) { return nr::bpr_loss(x); }
float hc_bpr_dloss(float x) { return nr::bpr_dloss(x); }

void hc_adam_dense(int n, const float* g, float* var, float* m, float* v, float alpha, float b1,
                   float b2, float eps) {
  for (int i = 0; i < n; ++i) nr::adam_dense_tf(g[i], var[i], m[i], v[i], alpha, 1.0f - b1, 1.0f - b2, eps);
}
void hc_adam_sparse(int n, const float* g, float* var, float* m, float* v, float alpha, float b1,
                    float b2, float eps) {
  for (int i = 0; i < n; ++i)
    nr::adam_sparse_tf(g[i], var[i], m[i], v[i], alpha, b1, b2, 1.0f - b1, 1.0f - b2, eps);
}

}  // extern "C"
