// Dev tool: a read-only stream (what a matrix-vector product is to HBM) of one weight-matrix-sized buffer per launch, in a few
// shapes (membench_read.py) — the practical roof of the decode step's tv_gemv_bf16_fwd on the box at hand.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// U independent 16-byte loads per lane in flight; a lane xors what it read and stores only on an (impossible) value
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u4v* __restrict__ x, unsigned* __restrict__ sink, long n) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  u4v acc = {0, 0, 0, 0};
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  for (; i < n; i += stride) acc ^= x[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) sink[0] = 1;
}

extern "C" int mr_launch(const void* x, void* sink, long nvec, int grid, int unroll, int nt, void* stream) {
  const u4v* xp = (const u4v*)x;
  hipStream_t st = (hipStream_t)stream;
#define MR(U) do { if (nt) read_kernel<U, true><<<grid, 256, 0, st>>>(xp, (unsigned*)sink, nvec); else read_kernel<U, false><<<grid, 256, 0, st>>>(xp, (unsigned*)sink, nvec); } while (0)
  switch (unroll) {
    case 1: MR(1); break;
    case 2: MR(2); break;
    case 4: MR(4); break;
    default: MR(8); break;
  }
  return (int)hipGetLastError();
}
