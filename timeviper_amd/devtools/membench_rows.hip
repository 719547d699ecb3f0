#include <hip/hip_runtime.h>
#include <stdint.h>
// Each work-group (4 waves) owns 640 bytes of every row (its 4 "heads" of 160 bytes): wave w copies the 160 bytes of head w of 6.4 rows per
// instruction pattern simplified: lane l of a wave handles piece (l % 10) of row (rowblock * 6 + l / 10), 60 lanes active.
// layout 0: rows of `row_bytes` (20 480) bytes, work-group g at column offset 640 g;  layout 1: work-group-major: g owns rows back to back.
__global__ __launch_bounds__(256) void mb_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int rows_per_wg_seg, int nseg, int layout,
                                                 long row_bytes, int rw) {
  const int g = blockIdx.x % 32, seg = blockIdx.x / 32;     // 32 work-groups across a row, nseg segments of rows
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane >= 60) return;
  const int piece = lane % 10, rl = lane / 10;
  const long total_rows = (long)rows_per_wg_seg * nseg;
  for (int r0 = 0; r0 < rows_per_wg_seg; r0 += 6 * 8) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long row = (long)seg * rows_per_wg_seg + r0 + 6 * u + rl;
      long off;
      if (layout == 0) off = row * row_bytes + 640 * g + 160 * wave + 16 * piece;
      else off = ((long)g * total_rows + row) * 640 + 160 * wave + 16 * piece;
      v[u] = (r0 + 6 * u + rl < rows_per_wg_seg) ? x[off / 16] : uint4{0, 0, 0, 0};
    }
    if (rw) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long row = (long)seg * rows_per_wg_seg + r0 + 6 * u + rl;
        long off;
        if (layout == 0) off = row * row_bytes + 640 * g + 160 * wave + 16 * piece;
        else off = ((long)g * total_rows + row) * 640 + 160 * wave + 16 * piece;
        if (r0 + 6 * u + rl < rows_per_wg_seg) { uint4 o = v[u]; o.x += 1; y[off / 16] = o; }
      }
    } else {
      unsigned s = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u].x;
      if (s == 0x12345678u) y[0] = v[0];
    }
  }
}
extern "C" void mb_launch(const void* x, void* y, int rows_per_wg_seg, int nseg, int layout, long row_bytes, int rw) {
  mb_kernel<<<dim3(32 * nseg), 256, 0, 0>>>((const uint4*)x, (uint4*)y, rows_per_wg_seg, nseg, layout, row_bytes, rw);
}
