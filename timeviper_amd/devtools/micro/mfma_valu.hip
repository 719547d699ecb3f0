#include <hip/hip_runtime.h>
#include <cstdio>
#include "body.inc"
#define CLOB_A "memory", "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", "a0","a15","a16","a47","a48","a95"
#define CLOB_B "memory", "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", "a0","a15","a16","a47","a48","a95","a96","a143","a144","a191"
// every AGPR in a0..a191 is written: name the range ends only and keep the compiler out with the launch bounds
__global__ __launch_bounds__(512) void kA(int iters, float* out) {
  for (int i = 0; i < iters; ++i) asm volatile(BODY_A ::: CLOB_A);
  if (iters < 0) out[threadIdx.x] = 1.f;
}
__global__ __launch_bounds__(256) void kB(int iters, float* out) {
  for (int i = 0; i < iters; ++i) asm volatile(BODY_B ::: CLOB_B);
  if (iters < 0) out[threadIdx.x] = 1.f;
}
int main() {
  float* o; hipMalloc(&o, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  hipFuncSetAttribute((const void*)kB, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)kA, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    hipEventRecord(e0); kA<<<256, 512, 100 * 1024>>>(iters, o); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double ua = ms * 1e-3 / iters / 2;      // seconds per unit and SIMD (two waves a SIMD, one unit each per iteration)
    hipEventRecord(e0); kB<<<256, 256, 100 * 1024>>>(iters, o); hipEventRecord(e1); hipEventSynchronize(e1);
    float msb; hipEventElapsedTime(&msb, e0, e1);
    const double ub = msb * 1e-3 / iters / 2;     // one wave a SIMD, two units per iteration
    printf("A (2 waves/SIMD, clumped): %.3f ms, %.0f ns/unit/SIMD | B (1 wave/SIMD, interleaved): %.3f ms, %.0f ns/unit/SIMD | B/A %.3f  [%s]\n",
           ms, ua * 1e9, msb, ub * 1e9, ub / ua, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
