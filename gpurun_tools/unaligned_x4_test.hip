// unaligned_x4_test.hip -- are 16-byte global loads at 4-byte-aligned addresses (rows of 135 floats) legal and fast on gfx950?
//   hipcc --offload-arch=gfx950 -O3 unaligned_x4_test.hip -o unaligned_x4_test && ./unaligned_x4_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
// every lane sums `cols` floats of its row (row stride ld floats) starting at column c0, with x4 loads (mode 1) or dword loads (0)
__global__ void k(const float* x, float* out, long rows, int ld, int c0, int mode) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* p = x + r * ld + c0;
  float s = 0.f;
  if (mode) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f4 v;
      asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p + 4 * j) : "memory");
      s += v[0] + v[1] + v[2] + v[3];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 32; ++j) s += p[j];
  }
  out[r] = s;
}
int main() {
  const long rows = 1 << 20; const int ld = 135;
  float *x, *o0, *o1; CK(hipMalloc(&x, rows * ld * 4 + 256)); CK(hipMalloc(&o0, rows * 4)); CK(hipMalloc(&o1, rows * 4));
  float* hx = (float*)malloc(rows * ld * 4);
  for (long i = 0; i < rows * ld; ++i) hx[i] = (float)((i * 7) % 13);
  CK(hipMemcpy(x, hx, rows * ld * 4, hipMemcpyHostToDevice));
  float *h0 = (float*)malloc(rows * 4), *h1 = (float*)malloc(rows * 4);
  for (int c0 = 0; c0 < 4; ++c0) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms[2];
    for (int mode = 0; mode < 2; ++mode) {
      hipLaunchKernelGGL(k, dim3(rows / 256), dim3(256), 0, 0, x, mode ? o1 : o0, rows, ld, c0, mode);
      CK(hipEventRecord(e0));
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(rows / 256), dim3(256), 0, 0, x, mode ? o1 : o0, rows, ld, c0, mode);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[mode], e0, e1));
    }
    CK(hipMemcpy(h0, o0, rows * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1, o1, rows * 4, hipMemcpyDeviceToHost));
    long bad = 0; for (long i = 0; i < rows; ++i) bad += h0[i] != h1[i];
    printf("c0 %d: dword loads %.1f us, x4 loads at 4-byte alignment %.1f us (with a wait per load), mismatches %ld\n", c0, ms[0] / 5 * 1e3, ms[1] / 5 * 1e3, bad);
  }
  return 0;
}
