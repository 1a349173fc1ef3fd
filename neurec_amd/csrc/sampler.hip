// sampler.hip — BPR negative sampling + epoch shuffle, entirely on the device.
//
// Stands in for PairwiseSampler.__iter__ (data/sampler.py:198-206):
//   _sampling_negative_items   data/sampler.py:71-90  (per-user rejection
//                              sampling through batch_randint_choice,
//                              util/cython/random_choice.pyx:20-89)
//   DataIterator(shuffle=True) util/data_iterator.py:58-60,145-152
//                              (one permutation of the E positives per epoch)
//
// One thread per output slot p of the epoch stream: t = perm(p) is the
// positive it carries (position in the user-major, item-ascending train CSR,
// i.e. the reference's users_list/pos_items_list order), the negative comes
// from a counter-based xorshift64* stream keyed by (seed, epoch, t, n) and the
// exclusion test is a binary search in the user's ascending CSR row instead
// of an unordered_set.  No host round trip, no materialised permutation.
//
// The reference's own random streams (glibc rand() for negatives, numpy
// MT19937 for the permutation) are inputs that cannot be reproduced on a
// GPU; what is preserved is the distribution and the stream structure
// (alignment of negatives with positives, one pass over every positive per
// epoch, last short batch kept).  See DESIGN.md §sampler.
#include "nr_common.h"

namespace {

__global__ __launch_bounds__(256) void sample_epoch_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const int32_t* __restrict__ row_of, int64_t n_inter, int n_items, int neg_num, uint64_t seed,
    uint64_t epoch, int shuffle, int64_t out_begin, int64_t out_count,
    int32_t* __restrict__ users_out, int32_t* __restrict__ pos_out,
    int32_t* __restrict__ neg_out) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= out_count) return;
  const int64_t p = out_begin + q;
  const uint64_t perm_key = nr::splitmix64(seed ^ nr::splitmix64(epoch + 0x51ed27ull));
  const int64_t t = shuffle ? (int64_t)nr::permute_index((uint64_t)p, (uint64_t)n_inter, perm_key)
                            : p;
  const int32_t u = row_of[t];
  const int64_t b = indptr[u];
  const int n_excl = (int)(indptr[u + 1] - b);
  users_out[q] = u;
  pos_out[q] = indices[t];
  for (int n = 0; n < neg_num; ++n) {
    nr::XorShift64s g;
    g.seed(seed, epoch, (uint64_t)t * (uint64_t)neg_num + (uint64_t)n);
    neg_out[q * neg_num + n] = nr::draw_negative(g, n_items, indices + b, n_excl);
  }
}

// batch_randint_choice: one thread per draw when replace=True; one thread per
// request (sequential, checks its own earlier draws) when replace=False.
__global__ __launch_bounds__(256) void randint_choice_kernel(
    int high, int n_req, const int64_t* __restrict__ out_off,
    const int64_t* __restrict__ excl_ptr, const int32_t* __restrict__ excl, int replace,
    uint64_t seed, uint64_t counter, int32_t* __restrict__ out) {
  if (replace) {
    const int64_t total = out_off[n_req];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // request owning output slot i: last q with out_off[q] <= i
    int lo = 0, hi = n_req;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (out_off[mid] <= i) lo = mid; else hi = mid;
    }
    const int64_t eb = excl_ptr ? excl_ptr[lo] : 0;
    const int ne = excl_ptr ? (int)(excl_ptr[lo + 1] - eb) : 0;
    nr::XorShift64s g;
    g.seed(seed, counter, (uint64_t)i);
    out[i] = nr::draw_negative(g, high, excl + eb, ne);
  } else {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_req) return;
    const int64_t ob = out_off[q];
    const int cnt = (int)(out_off[q + 1] - ob);
    const int64_t eb = excl_ptr ? excl_ptr[q] : 0;
    const int ne = excl_ptr ? (int)(excl_ptr[q + 1] - eb) : 0;
    nr::XorShift64s g;
    g.seed(seed, counter, (uint64_t)ob);
    for (int k = 0; k < cnt;) {
      const int32_t a = nr::draw_negative(g, high, excl + eb, ne);
      bool dup = false;
      for (int m = 0; m < k; ++m) dup |= (out[ob + m] == a);
      if (!dup) { out[ob + k] = a; ++k; }
    }
  }
}

// Training INSTANCES beyond the plain BPR triplet (data/sampler.py:93-155 PointwiseSampler,
// :216-289 TimeOrderPointwiseSampler, :292-354 TimeOrderPairwiseSampler), one thread per slot of
// the epoch stream, everything the reference builds with Python lists + DataIterator formed here:
//
//   rows      r = 0..R-1, in the iteration order of the reference's user_pos_dict; row_user[r] is
//             the user id;  seq[seq_ptr[r] ..) the row's items in the order the dict holds them
//             (time order for the TimeOrder samplers);  excl[excl_ptr[r] ..) the same items
//             ascending, for the exclusion test (random_choice.pyx:50-54 keeps an unordered_set);
//   instance  t = inst_ptr[r] + k, k < len_r - high_order: recent items seq[k .. k+high_order),
//             positive item seq[k + high_order]  (_generative_time_order_positive_items,
//             sampler.py:42-68; high_order = 0 is _generate_positive_items, :24-39);
//   pairwise  slot s <-> instance s; neg_out[s][0..neg_num) drawn for instance s;
//   pointwise slot s <-> (c = s / n_inst, t = s % n_inst): c = 0 is the positive (label 1), c >= 1
//             the (c-1)-th negative of instance t (label 0) — the layout the reference obtains by
//             concatenating pos_items_list with the TRANSPOSED negative array (sampler.py:141-143)
//             and repeating users_list (neg_num + 1) times (:131);
//   stream    output position p carries slot perm(p) (one permutation per epoch, data_iterator.py:58-60).
// The negative of (instance t, n) comes from the same counter-based stream as the BPR kernel's:
// key (seed, epoch, t * neg_num + n).
struct InstanceStream {
  const int64_t* seq_ptr; const int32_t* seq;
  const int64_t* excl_ptr; const int32_t* excl;
  const int64_t* inst_ptr; const int32_t* inst_row; const int32_t* row_user;
  int64_t n_inst, n_slots;
  int high_order, n_items, neg_num, pointwise, shuffle;
  uint64_t seed, epoch;
  int64_t out_begin, out_count;
  int32_t* users_out; int32_t* recent_out; int32_t* items_out; int32_t* neg_out; float* labels_out;
};

__global__ __launch_bounds__(256) void sample_instances_kernel(const InstanceStream a) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.out_count) return;
  const int64_t p = a.out_begin + q;
  const uint64_t perm_key = nr::splitmix64(a.seed ^ nr::splitmix64(a.epoch + 0x51ed27ull));
  const int64_t s = a.shuffle ? (int64_t)nr::permute_index((uint64_t)p, (uint64_t)a.n_slots, perm_key) : p;
  const int64_t t = a.pointwise ? s % a.n_inst : s;
  const int c = a.pointwise ? (int)(s / a.n_inst) : 0;
  const int32_t r = a.inst_row[t];
  const int64_t at = a.seq_ptr[r] + (t - a.inst_ptr[r]);       // first recent item of the window
  a.users_out[q] = a.row_user[r];
  for (int h = 0; h < a.high_order; ++h) a.recent_out[q * a.high_order + h] = a.seq[at + h];
  const int64_t eb = a.excl_ptr[r];
  const int n_excl = (int)(a.excl_ptr[r + 1] - eb);
  if (a.pointwise) {
    if (c == 0) {
      a.items_out[q] = a.seq[at + a.high_order];
      a.labels_out[q] = 1.0f;
    } else {
      nr::XorShift64s g;
      g.seed(a.seed, a.epoch, (uint64_t)t * (uint64_t)a.neg_num + (uint64_t)(c - 1));
      a.items_out[q] = nr::draw_negative(g, a.n_items, a.excl + eb, n_excl);
      a.labels_out[q] = 0.0f;
    }
  } else {
    a.items_out[q] = a.seq[at + a.high_order];
    for (int n = 0; n < a.neg_num; ++n) {
      nr::XorShift64s g;
      g.seed(a.seed, a.epoch, (uint64_t)t * (uint64_t)a.neg_num + (uint64_t)n);
      a.neg_out[q * a.neg_num + n] = nr::draw_negative(g, a.n_items, a.excl + eb, n_excl);
    }
  }
}

}  // namespace

extern "C" {

int nrhip_sample_bpr_epoch(const int64_t* d_tr_indptr, const int32_t* d_tr_indices,
                           const int32_t* d_row_of, int64_t n_inter, int n_items, int neg_num,
                           uint64_t seed, uint64_t epoch, int shuffle, int64_t out_begin,
                           int64_t out_count, int32_t* d_users_out, int32_t* d_pos_out,
                           int32_t* d_neg_out, void* stream) {
  NR_REQUIRE(d_tr_indptr && d_tr_indices && d_row_of && d_users_out && d_pos_out && d_neg_out,
             NR_ERR_ARG, "sample_bpr_epoch: null pointer argument");
  NR_REQUIRE(neg_num >= 1, NR_ERR_ARG, "'neg_num' must be a positive integer.");
  NR_REQUIRE(n_items >= 1 && n_inter >= 0 && out_begin >= 0 && out_count >= 0 &&
                 out_begin + out_count <= n_inter,
             NR_ERR_ARG, "sample_bpr_epoch: bad range [%lld,+%lld) of %lld", (long long)out_begin,
             (long long)out_count, (long long)n_inter);
  if (out_count == 0) return NR_OK;
  const int64_t blocks = (out_count + 255) / 256;
  hipLaunchKernelGGL(sample_epoch_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, d_tr_indptr, d_tr_indices, d_row_of, n_inter, n_items,
                     neg_num, seed, epoch, shuffle, out_begin, out_count, d_users_out, d_pos_out,
                     d_neg_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_randint_choice_batch(int high, int n_req, int64_t total, const int64_t* d_out_offsets,
                               const int64_t* d_excl_indptr, const int32_t* d_excl, int replace,
                               uint64_t seed, uint64_t call_counter, int32_t* d_out,
                               void* stream) {
  NR_REQUIRE(high >= 1 && n_req >= 0 && total >= 0 && d_out_offsets && d_out, NR_ERR_ARG,
             "randint_choice_batch: bad arguments");
  if (n_req == 0 || total == 0) return NR_OK;
  const int64_t work = replace ? total : (int64_t)n_req;
  hipLaunchKernelGGL(randint_choice_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, high, n_req, d_out_offsets, d_excl_indptr, d_excl,
                     replace, seed, call_counter, d_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_sample_instances_epoch(const int64_t* d_seq_ptr, const int32_t* d_seq,
                                 const int64_t* d_excl_ptr, const int32_t* d_excl,
                                 const int64_t* d_inst_ptr, const int32_t* d_inst_row,
                                 const int32_t* d_row_user, int64_t n_inst, int high_order,
                                 int n_items, int neg_num, int pointwise, uint64_t seed,
                                 uint64_t epoch, int shuffle, int64_t out_begin, int64_t out_count,
                                 int32_t* d_users_out, int32_t* d_recent_out, int32_t* d_items_out,
                                 int32_t* d_neg_out, float* d_labels_out, void* stream) {
  NR_REQUIRE(d_seq_ptr && d_seq && d_excl_ptr && d_excl && d_inst_ptr && d_inst_row && d_row_user &&
                 d_users_out && d_items_out,
             NR_ERR_ARG, "sample_instances_epoch: null pointer argument");
  NR_REQUIRE(neg_num >= 1, NR_ERR_ARG, "'neg_num' must be a positive integer.");
  NR_REQUIRE(high_order >= 0, NR_ERR_ARG, "'high_order' must be a positive integer.");
  NR_REQUIRE(high_order == 0 || d_recent_out, NR_ERR_ARG, "sample_instances_epoch: no buffer for the recent items");
  NR_REQUIRE(pointwise ? d_labels_out != nullptr : d_neg_out != nullptr, NR_ERR_ARG,
             "sample_instances_epoch: %s", pointwise ? "no label buffer" : "no negative buffer");
  const int64_t n_slots = pointwise ? n_inst * ((int64_t)neg_num + 1) : n_inst;
  NR_REQUIRE(n_items >= 1 && n_inst >= 0 && out_begin >= 0 && out_count >= 0 && out_begin + out_count <= n_slots,
             NR_ERR_ARG, "sample_instances_epoch: bad range [%lld,+%lld) of %lld", (long long)out_begin,
             (long long)out_count, (long long)n_slots);
  if (out_count == 0) return NR_OK;
  InstanceStream a{d_seq_ptr, d_seq, d_excl_ptr, d_excl, d_inst_ptr, d_inst_row, d_row_user, n_inst, n_slots,
                   high_order, n_items, neg_num, pointwise ? 1 : 0, shuffle ? 1 : 0, seed, epoch, out_begin,
                   out_count, d_users_out, d_recent_out, d_items_out, d_neg_out, d_labels_out};
  hipLaunchKernelGGL(sample_instances_kernel, dim3((unsigned)((out_count + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, a);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"
