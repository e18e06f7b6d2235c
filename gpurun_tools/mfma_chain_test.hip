// Diagnostic (round 3): is a v_mfma_f32_16x16x4_f32 accumulation chain bitwise an ordered fmaf chain, and in WHICH order of the
// four k values of one instruction?  16 x 16 x 128 products, random data, several candidate orders.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(const float* A, const float* B, float* D) {     // A[16][128] rows = codes, B[16][128] rows = batch rows
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < 8; ++s) {
    const float4 a = *(const float4*)(A + i * 128 + 16 * s + 4 * q), b = *(const float4*)(B + i * 128 + 16 * s + 4 * q);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = acc[r];         // D[code][row]
}
__global__ void k_fma(const float* A, const float* B, float* D, int order) {
  const int code = threadIdx.x >> 4, row = threadIdx.x & 15;
  float acc = 0.f;
  for (int s = 0; s < 8; ++s)
    for (int c = 0; c < 4; ++c) {
      if (order == 0) for (int qq = 0; qq < 4; ++qq) { const int k = 16 * s + 4 * qq + c; acc = fmaf(A[code * 128 + k], B[row * 128 + k], acc); }
      else if (order == 1) for (int qq = 3; qq >= 0; --qq) { const int k = 16 * s + 4 * qq + c; acc = fmaf(A[code * 128 + k], B[row * 128 + k], acc); }
      else if (order == 2) {   // pairwise inside the instruction: (p0 + p1) + (p2 + p3) + acc, unfused
        float p[4]; for (int qq = 0; qq < 4; ++qq) { const int k = 16 * s + 4 * qq + c; p[qq] = A[code * 128 + k] * B[row * 128 + k]; }
        acc = ((p[0] + p[1]) + (p[2] + p[3])) + acc;
      } else {                 // products exact (double), one rounding per instruction
        double t = acc; for (int qq = 0; qq < 4; ++qq) { const int k = 16 * s + 4 * qq + c; t += (double)A[code * 128 + k] * (double)B[row * 128 + k]; }
        acc = (float)t;
      }
    }
  D[code * 16 + row] = acc;
}
int main() {
  std::vector<float> hA(16 * 128), hB(16 * 128);
  srand(7);
  float *A, *B, *D0, *D1;
  hipMalloc(&A, 8192); hipMalloc(&B, 8192); hipMalloc(&D0, 1024); hipMalloc(&D1, 1024);
  int mism[4] = {0, 0, 0, 0};
  for (int trial = 0; trial < 200; ++trial) {
    for (auto& v : hA) v = (rand() / (float)RAND_MAX * 2 - 1) * (trial % 3 == 0 ? 1e3f : 1.f);
    for (auto& v : hB) v = (rand() / (float)RAND_MAX * 2 - 1) * (trial % 5 == 0 ? 1e-3f : 1.f);
    hipMemcpy(A, hA.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, A, B, D0);
    std::vector<float> r0(256), r1(256);
    hipMemcpy(r0.data(), D0, 1024, hipMemcpyDeviceToHost);
    for (int o = 0; o < 4; ++o) {
      hipLaunchKernelGGL(k_fma, dim3(1), dim3(256), 0, 0, A, B, D1, o);
      hipMemcpy(r1.data(), D1, 1024, hipMemcpyDeviceToHost);
      for (int e = 0; e < 256; ++e) if (memcmp(&r0[e], &r1[e], 4)) ++mism[o];
    }
  }
  printf("mismatching elements of 51200: fmaf q ascending %d, fmaf q descending %d, pairwise unfused %d, exact products one rounding %d\n",
         mism[0], mism[1], mism[2], mism[3]);
  return 0;
}
