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

}  // extern "C"
