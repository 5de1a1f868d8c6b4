// Row mover for the device-resident trajectory store.
//
// The reference keeps per-environment trajectories in tf.Variables and moves them with
// scatter_nd_update / sparse_read / gather_nd (/root/reference/common/utils.py:155-257, 461-543), then
// transposes whole batches to time-major on the host (utils.py:735-761).  Here every field of the store
// lives TIME-MAJOR in HBM ([full_length, num_envs, row]); appending a step, resetting, carrying the overlap
// and emitting completed unrolls straight into a time-major training batch are all the same primitive:
//     dst[dst_rows[i]] = src[src_rows[i]]     (rows of `row_bytes` bytes; NULL index = identity; NULL src = 0)
// HBM-bound byte work: 16 B per lane when the row size and pointers allow, 2*row_bytes per row moved.
#include "common.h"
#include "../../include/seedhip.h"

namespace {

template <typename V>
__global__ void __launch_bounds__(256)
rows_move_kernel(V* __restrict__ dst, const long long* __restrict__ dst_rows, const V* __restrict__ src,
                 const long long* __restrict__ src_rows, long long n, long long row_elems) {
  const long long total = n * row_elems;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / row_elems, e = i - r * row_elems;
    const long long dr = dst_rows ? dst_rows[r] : r;
    V v{};
    if (src) v = src[(src_rows ? src_rows[r] : r) * row_elems + e];
    dst[dr * row_elems + e] = v;
  }
}

}  // namespace

extern "C" int seedhip_rows_move(void* dst, const long long* dst_rows, const void* src, const long long* src_rows,
                                 long long n, long long row_bytes, void* stream) {
  SEEDHIP_REQUIRE(n >= 0 && row_bytes >= 1, "rows_move: bad n / row_bytes");
  if (n == 0) return SEEDHIP_OK;
  SEEDHIP_REQUIRE(dst, "rows_move: null dst");
  hipStream_t s = (hipStream_t)stream;
  const uintptr_t al = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)row_bytes;
  const long long total_bytes = n * row_bytes;
  auto grid = [](long long elems) { long long g = (elems + 255) / 256; return (int)(g > 8192 ? 8192 : g); };
  if ((al & 15) == 0) {
    hipLaunchKernelGGL(rows_move_kernel<uint4>, dim3(grid(total_bytes / 16)), dim3(256), 0, s, (uint4*)dst, dst_rows,
                       (const uint4*)src, src_rows, n, row_bytes / 16);
  } else if ((al & 3) == 0) {
    hipLaunchKernelGGL(rows_move_kernel<uint32_t>, dim3(grid(total_bytes / 4)), dim3(256), 0, s, (uint32_t*)dst,
                       dst_rows, (const uint32_t*)src, src_rows, n, row_bytes / 4);
  } else {
    hipLaunchKernelGGL(rows_move_kernel<uint8_t>, dim3(grid(total_bytes)), dim3(256), 0, s, (uint8_t*)dst, dst_rows,
                       (const uint8_t*)src, src_rows, n, row_bytes);
  }
  return seedhip::check_launch("rows_move_kernel");
}
