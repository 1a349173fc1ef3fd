// Routing of a batch's row lookups for row-sharded tables (neurec_amd/sharded.py): which rank owns
// each of the 3·B looked-up rows (tf.nn.embedding_lookup of MF.py:57-58 / LightGCN.py:99-104 when the table is cut
// into per-rank blocks), the requests in owner order, and — on the owner's side — the keys that put the returning
// gradient rows into the order of the GLOBAL batch (TF's unsorted_segment_sum over the concatenated batch).
// Integer bookkeeping only; replaces a dozen torch glue ops per step (argsort / bincount / cat / index_put ...).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "neurec_hip.h"
#include "nr_common.h"

namespace {

// BipartitePartition (neurec_amd/parallel.py): rank r holds users [r·bu, (r+1)·bu) then items [r·bi, (r+1)·bi)
__device__ __forceinline__ void owner_local(int node, int n_users, int bu, int bi, int& owner, int& local) {
  if (node < n_users) {
    owner = node / bu;
    local = node - owner * bu;
  } else {
    const int it = node - n_users;
    owner = it / bi;
    local = bu + (it - owner * bi);
  }
}

__device__ __forceinline__ int node_of(const int32_t* users, const int32_t* pos, const int32_t* neg, int batch,
                                       int n_users, int p) {
  const int cls = p / batch, b = p - cls * batch;
  return cls == 0 ? users[b] : n_users + (cls == 1 ? pos : neg)[b];
}

__global__ __launch_bounds__(256) void route_keys_kernel(const int32_t* __restrict__ users,
                                                         const int32_t* __restrict__ pos,
                                                         const int32_t* __restrict__ neg, int batch, int n_users,
                                                         int bu, int bi, uint64_t* __restrict__ keys,
                                                         int32_t* __restrict__ counts) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= 3 * batch) return;
  int owner, local;
  owner_local(node_of(users, pos, neg, batch, n_users, p), n_users, bu, bi, owner, local);
  keys[p] = ((uint64_t)(uint32_t)owner << 32) | (uint32_t)p;      // sorted: by owner, then request order (stable)
  if (counts) atomicAdd(&counts[owner], 1);
}

__global__ __launch_bounds__(256) void route_pack_kernel(const uint64_t* __restrict__ keys,
                                                         const int32_t* __restrict__ users,
                                                         const int32_t* __restrict__ pos,
                                                         const int32_t* __restrict__ neg, int batch, int n_users,
                                                         int bu, int bi, int code_base, int32_t* __restrict__ packed,
                                                         int32_t* __restrict__ order, int32_t* __restrict__ inv) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * batch) return;
  const int p = (int)(uint32_t)keys[i];
  int owner, local;
  owner_local(node_of(users, pos, neg, batch, n_users, p), n_users, bu, bi, owner, local);
  const int cls = p / batch;
  packed[2 * i] = local;
  packed[2 * i + 1] = cls * code_base + (p - cls * batch);        // occurrence code: class, position in MY batch
  order[i] = p;
  inv[p] = i;
}

// owner side: key = (local row << 32) | global position of the occurrence in the concatenated global batch
// (class-major, then source rank, then position in that rank's batch)
__global__ __launch_bounds__(256) void route_owner_keys_kernel(const int32_t* __restrict__ rows,
                                                               const int32_t* __restrict__ codes, int n,
                                                               const int32_t* __restrict__ recv_prefix,
                                                               const int32_t* __restrict__ size_off, int world,
                                                               int G, int code_base, uint64_t* __restrict__ keys,
                                                               int32_t* __restrict__ index_of_pos) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int src = 0;                                     // rows arrive ordered by source rank: recv_prefix[src] <= i
  while (src + 1 < world && recv_prefix[src + 1] <= i) ++src;
  const int code = codes[i], cls = code / code_base;
  const int gpos = cls * G + size_off[src] + (code - cls * code_base);
  keys[i] = ((uint64_t)(uint32_t)rows[i] << 32) | (uint32_t)gpos;
  index_of_pos[gpos] = i;
}

// ---- the routing tables of a whole epoch in three launches (one slot of 3 * batch requests per batch; the last batch
// may be shorter) — per batch exactly what route_keys / route_pack / route_owner_keys compute
__device__ __forceinline__ bool epoch_slot(int64_t q, int64_t n, int batch, int& k, int& p, int& nbk) {
  k = (int)(q / (3 * (int64_t)batch));
  p = (int)(q - (int64_t)k * 3 * batch);
  const int64_t left = n - (int64_t)k * batch;
  nbk = (int)(left < batch ? left : batch);
  return p < 3 * nbk;
}

__global__ __launch_bounds__(256) void route_epoch_keys_kernel(const int32_t* __restrict__ users,
                                                               const int32_t* __restrict__ pos,
                                                               const int32_t* __restrict__ neg, int64_t n, int batch,
                                                               int n_users, int bu, int bi, int world,
                                                               uint64_t* __restrict__ keys,
                                                               int32_t* __restrict__ counts) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int k, p, nbk;
  if (q >= 3 * ((n + batch - 1) / batch) * batch) return;
  if (!epoch_slot(q, n, batch, k, p, nbk)) { keys[q] = ~0ull; return; }
  const int64_t b0 = (int64_t)k * batch;
  int owner, local;
  owner_local(node_of(users + b0, pos + b0, neg + b0, nbk, n_users, p), n_users, bu, bi, owner, local);
  keys[q] = ((uint64_t)(uint32_t)owner << 32) | (uint32_t)p;
  if (counts) atomicAdd(&counts[(int64_t)k * world + owner], 1);      // optional: 3 * batch atomics per counter
}

__global__ __launch_bounds__(256) void route_epoch_pack_kernel(const uint64_t* __restrict__ keys,
                                                               const int32_t* __restrict__ users,
                                                               const int32_t* __restrict__ pos,
                                                               const int32_t* __restrict__ neg, int64_t n, int batch,
                                                               int n_users, int bu, int bi, int code_base,
                                                               int32_t* __restrict__ packed,
                                                               int32_t* __restrict__ order, int32_t* __restrict__ inv) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int k, i, nbk;
  if (q >= 3 * ((n + batch - 1) / batch) * batch) return;
  if (!epoch_slot(q, n, batch, k, i, nbk)) return;
  const int64_t b0 = (int64_t)k * batch, s0 = (int64_t)k * 3 * batch;
  const int p = (int)(uint32_t)keys[q];
  int owner, local;
  owner_local(node_of(users + b0, pos + b0, neg + b0, nbk, n_users, p), n_users, bu, bi, owner, local);
  const int cls = p / nbk;
  packed[2 * q] = local;
  packed[2 * q + 1] = cls * code_base + (p - cls * nbk);
  order[q] = p;
  inv[s0 + p] = i;
}

// element j of the received stream (laid out [batch][source rank]) belongs to batch batch_of[j]
__global__ __launch_bounds__(256) void route_epoch_owner_keys_kernel(
    const int32_t* __restrict__ rows, const int32_t* __restrict__ codes, const int32_t* __restrict__ batch_of,
    int64_t n, const int64_t* __restrict__ asked_off, const int32_t* __restrict__ recv_prefix,
    const int32_t* __restrict__ size_off, const int32_t* __restrict__ global_len, int world, int code_base,
    int64_t iop_stride, uint64_t* __restrict__ keys, int32_t* __restrict__ index_of_pos) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int k = batch_of[j];
  const int i = (int)(j - asked_off[k]);
  const int32_t* rp = recv_prefix + (int64_t)k * (world + 1);
  int src = 0;
  while (src + 1 < world && rp[src + 1] <= i) ++src;
  const int code = codes[j], cls = code / code_base;
  const int gpos = cls * global_len[k] + size_off[(int64_t)k * world + src] + (code - cls * code_base);
  keys[j] = ((uint64_t)(uint32_t)rows[j] << 32) | (uint32_t)gpos;
  index_of_pos[(int64_t)k * iop_stride + gpos] = i;
}

}  // namespace

extern "C" {

int nrhip_route_batch(const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg, int batch, int n_users,
                      int bu, int bi, int code_base, uint64_t* d_keys, int32_t* d_packed, int32_t* d_order,
                      int32_t* d_inv, int32_t* d_counts, int world, void* stream) {
  NR_REQUIRE(d_users && d_pos && d_neg && d_keys && d_packed && d_order && d_inv && batch >= 0 && n_users >= 0 &&
                 bu >= 1 && bi >= 1 && code_base > batch && world >= 1, NR_ERR_ARG, "route_batch: bad arguments");
  if (batch == 0) return NR_OK;
  hipStream_t st = (hipStream_t)stream;
  if (d_counts) NR_CHECK_HIP(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * world, st));
  const dim3 grid((3 * batch + 255) / 256), block(256);
  hipLaunchKernelGGL(route_keys_kernel, grid, block, 0, st, d_users, d_pos, d_neg, batch, n_users, bu, bi, d_keys,
                     d_counts);
  NR_LAUNCH_CHECK();
  NR_TRY(nrhip_sort_u64(d_keys, 3 * batch, stream));
  hipLaunchKernelGGL(route_pack_kernel, grid, block, 0, st, d_keys, d_users, d_pos, d_neg, batch, n_users, bu, bi,
                     code_base, d_packed, d_order, d_inv);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_route_owner_keys(const int32_t* d_rows, const int32_t* d_codes, int n, const int32_t* d_recv_prefix,
                           const int32_t* d_size_off, int world, int global_batch, int code_base,
                           uint64_t* d_keys_out, int32_t* d_index_of_pos, void* stream) {
  NR_REQUIRE(d_recv_prefix && d_size_off && d_keys_out && d_index_of_pos && n >= 0 && world >= 1 &&
                 global_batch >= 0 && code_base > 0 && (n == 0 || (d_rows && d_codes)),
             NR_ERR_ARG, "route_owner_keys: bad arguments");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL(route_owner_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_rows,
                     d_codes, n, d_recv_prefix, d_size_off, world, global_batch, code_base, d_keys_out,
                     d_index_of_pos);
  NR_LAUNCH_CHECK();
  return nrhip_sort_u64(d_keys_out, n, stream);
}

/* The routing of EVERY batch of an epoch stream (d_users / d_pos / d_neg: n triplets cut into batches of `batch`; batch
 * k's requests live in slot [3*batch*k, 3*batch*(k+1))): nrhip_route_batch's outputs for all batches in three launches.
 * d_counts [n_batches][world] (optional; zeroed here) receives the per-destination request counts. */
int nrhip_route_epoch(const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg, int64_t n, int batch,
                      int n_users, int bu, int bi, int code_base, int world, uint64_t* d_keys, int32_t* d_packed,
                      int32_t* d_order, int32_t* d_inv, int32_t* d_counts, const int64_t* d_seg_off,
                      const int32_t* d_seg_len, void* stream) {
  NR_REQUIRE(d_users && d_pos && d_neg && d_keys && d_packed && d_order && d_inv && d_seg_off &&
                 d_seg_len && n >= 0 && batch >= 1 && n_users >= 0 && bu >= 1 && bi >= 1 && code_base > batch &&
                 world >= 1, NR_ERR_ARG, "route_epoch: bad arguments");
  if (n == 0) return NR_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nb = (n + batch - 1) / batch, slots = 3 * nb * batch;
  NR_REQUIRE(nb < (1ll << 31) && slots / 256 < (1ll << 31), NR_ERR_UNSUPPORTED, "route_epoch: %lld batches", (long long)nb);
  if (d_counts) NR_CHECK_HIP(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * nb * world, st));
  const dim3 grid((unsigned)((slots + 255) / 256)), block(256);
  hipLaunchKernelGGL(route_epoch_keys_kernel, grid, block, 0, st, d_users, d_pos, d_neg, n, batch, n_users, bu, bi,
                     world, d_keys, d_counts);
  NR_LAUNCH_CHECK();
  NR_TRY(nrhip_sort_u64_segments(d_keys, d_seg_off, d_seg_len, (int)nb, 3 * batch, stream));
  hipLaunchKernelGGL(route_epoch_pack_kernel, grid, block, 0, st, d_keys, d_users, d_pos, d_neg, n, batch, n_users, bu,
                     bi, code_base, d_packed, d_order, d_inv);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* Owner side of the same: the received (row, code) stream of the whole epoch laid out [batch][source rank]
 * (d_asked_off [n_batches + 1], d_batch_of[j]), per batch the source prefixes d_recv_prefix [n_batches][world + 1], the
 * offsets of the source ranks' batch lengths d_size_off [n_batches][world] and the global batch lengths d_global_len:
 * sorted keys (row << 32 | global position) per batch and d_index_of_pos [n_batches][iop_stride]. */
int nrhip_route_epoch_owner_keys(const int32_t* d_rows, const int32_t* d_codes, const int32_t* d_batch_of, int64_t n,
                                 const int64_t* d_asked_off, const int32_t* d_asked_len, int n_batches, int max_asked,
                                 const int32_t* d_recv_prefix, const int32_t* d_size_off,
                                 const int32_t* d_global_len, int world, int code_base, int64_t iop_stride,
                                 uint64_t* d_keys_out, int32_t* d_index_of_pos, void* stream) {
  NR_REQUIRE(n >= 0 && n_batches >= 0 && world >= 1 && code_base > 0 && iop_stride >= 0 &&
                 (n == 0 || (d_rows && d_codes && d_batch_of && d_asked_off && d_asked_len && d_recv_prefix &&
                             d_size_off && d_global_len && d_keys_out && d_index_of_pos)),
             NR_ERR_ARG, "route_epoch_owner_keys: bad arguments");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL(route_epoch_owner_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, d_rows, d_codes, d_batch_of, n, d_asked_off, d_recv_prefix, d_size_off,
                     d_global_len, world, code_base, iop_stride, d_keys_out, d_index_of_pos);
  NR_LAUNCH_CHECK();
  return nrhip_sort_u64_segments(d_keys_out, d_asked_off, d_asked_len, n_batches, max_asked, stream);
}

}  // extern "C"
