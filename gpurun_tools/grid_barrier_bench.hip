// Diagnostic micro-benchmark (not part of the library): cost of one in-kernel grid-wide exchange round on MI355X:
// every workgroup adds 128 fixed-point partial sums into a global slot (device-scope atomics), signals a counter, spins
// until all workgroups have signalled, then reads the 128 totals.  Prints microseconds per round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void rounds_kernel(unsigned long long* sums, unsigned int* counter, float* out, int nround,
                                                     int nblk, int mode) {
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int r = 0; r < nround; ++r) {
    unsigned long long* s = sums + (size_t)r * 128;
    if (mode == 0) {
      if (tid < 128) __hip_atomic_fetch_add(&s[tid], (unsigned long long)(tid + blockIdx.x + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(&counter[r], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&counter[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nblk) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (mode == 0 && tid < 128) acc += (float)__hip_atomic_load(&s[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < 128) out[blockIdx.x * 128 + tid] = acc;
}
int main() {
  const int nblk = 256, nround = 200;
  unsigned long long* sums; unsigned int* counter; float* out;
  hipMalloc(&sums, sizeof(unsigned long long) * 128 * nround);
  hipMalloc(&counter, sizeof(unsigned int) * nround);
  hipMalloc(&out, sizeof(float) * nblk * 128);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(sums, 0, sizeof(unsigned long long) * 128 * nround);
      hipMemset(counter, 0, sizeof(unsigned int) * nround);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(rounds_kernel, dim3(nblk), dim3(256), 0, 0, sums, counter, out, nround, nblk, mode);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (%s) rep %d: %.3f us per round\n", mode, mode == 0 ? "128 atomics + barrier + read" : "barrier only", rep, ms * 1e3f / nround);
    }
  }
  float h[128]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); printf("check %f\n", h[5]);
  return 0;
}
