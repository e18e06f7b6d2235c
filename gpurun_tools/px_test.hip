// px_test.hip -- the product's exchange (csrc/dec_persist.hpp) in isolation: 256 (or argv[1]) workgroups, T steps, every
// (argv[5]: 0 write-through records, 1 hop-1 records through the XCD's L2 unconditionally, 2 the announce / decide protocol)
// received sum checked against the closed form, uneven load, timing.   hipcc --offload-arch=gfx950 -O3 -I.. px_test.hip
__device__ unsigned long long px_stamps[4 * 16];     // [block slot][stamp]
__device__ int px_stamp_base = 0;
#define PX_STAMP(k)                                                                                                       \
  do {                                                                                                                    \
    const int sb_ = blockIdx.x == 0 ? 0 : (blockIdx.x == 5 ? 1 : (blockIdx.x == 128 ? 2 : (blockIdx.x == 255 ? 3 : -1)));  \
    if (threadIdx.x == 0 && sb_ >= 0 && px_stamp_on) px_stamps[sb_ * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
__device__ int px_stamp_on = 0;
#define PX_AFFINE_GRID 1      // rows of the 16 x 16 grid = workgroups of one residue class mod 8 (one XCD under round-robin placement)
#define G2V_PERSIST_DEVICE_CODE      // this translation unit owns the fault latch (dec_persist.hpp)
#include "../gesture2vec_amd/csrc/dec_persist.hpp"
#include <stdlib.h>
namespace g2v { void set_error(const char*, ...) {} }
using namespace g2v;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ float rec_val(int b, int e, int s) { return (float)((b * 131 + e * 7 + s * 13) % 1000); }
__global__ __launch_bounds__(256) void k(PersistX x, unsigned* errs, float* sink, int nblk, int T, int nmfma, int skew, int xl) {
  __shared__ float red[16 * 128];
  __shared__ float tot[128];
  __shared__ int xls;          // xl == 2: the product's protocol (announce the XCC id, decide behind the first hop 1)
  if (xl == 2 && threadIdx.x == 0) px_announce(x, blockIdx.x, &xls);
  if (xl != 2 && threadIdx.x == 0) xls = xl;
  __syncthreads();
  const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float aa = 1.0f + tid * 1e-3f;
  unsigned nerr = 0;
  for (int s = 1; s <= T; ++s) {
    for (int m = 0; m < nmfma; ++m) acc = mfma16(aa, 0.5f, acc);
    if (skew && ((b * 7 + s) & 15) == 0) __builtin_amdgcn_s_sleep(100);
    const int f0 = 16 * wave + 4 * q;
    if (i == 0) {
      __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(x.rec1 + ((size_t)(s & 1) * PX_MAX_NBLK + b) * PX_COLS, 0, PX_COLS * 8, 0x00020000);
      px_publish2(rr, f0, rec_val(b, f0, s), rec_val(b, f0 + 1, s), s, xls > 0);
      px_publish2(rr, f0 + 2, rec_val(b, f0 + 2, s), rec_val(b, f0 + 3, s), s, xls > 0);
      px_publish2(rr, 64 + f0, rec_val(b, 64 + f0, s), rec_val(b, 64 + f0 + 1, s), s, xls > 0);
      px_publish2(rr, 64 + f0 + 2, rec_val(b, 64 + f0 + 2, s), rec_val(b, 64 + f0 + 3, s), s, xls > 0);
    }
    if (tid == 0) { px_stamp_on = (s == 20); }
    __syncthreads();
    { const int sb_ = b == 0 ? 0 : (b == 5 ? 1 : (b == 128 ? 2 : (b == 255 ? 3 : -1)));
      if (tid == 0 && sb_ >= 0 && s == 20) px_stamps[sb_ * 16 + 15] = __builtin_amdgcn_s_memrealtime(); }
    px_exchange(x, s & 1, (unsigned)s, nblk, b, red, tot, tid, &xls);
    { const int sb_ = b == 0 ? 0 : (b == 5 ? 1 : (b == 128 ? 2 : (b == 255 ? 3 : -1)));
      if (tid == 0 && sb_ >= 0 && s == 20) px_stamps[sb_ * 16 + 14] = __builtin_amdgcn_s_memrealtime(); }
    if (tid < 128) {
      float expect = 0.f;
      for (int g = 0; g * 16 < nblk; ++g) { float t = 0.f; for (int m = g * 16; m < min(nblk, g * 16 + 16); ++m) t += rec_val(m, tid, s); expect += t; }
      if (tot[tid] != expect) ++nerr;
      aa += tot[tid] * 1e-12f;
    }
    __syncthreads();
  }
  if (nerr) atomicAdd(errs, nerr);
  if (tid == 0 && xls > 0) atomicAdd(errs + 1, 1u);      // workgroups whose row went through L2
  sink[(size_t)b * 256 + tid] = acc[0] + acc[1];
}
int main(int argc, char** argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 33, nmfma = argc > 3 ? atoi(argv[3]) : 0, skew = argc > 4 ? atoi(argv[4]) : 0, xl = argc > 5 ? atoi(argv[5]) : 0;
  void* xb; unsigned* errs; float* sink;
  CK(hipMalloc(&xb, PX_BYTES)); CK(hipMalloc(&errs, 8)); CK(hipMalloc(&sink, 256 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float sum = 0; unsigned herr = 0, nl2 = 0; const int reps = 20;
  for (int r = 0; r < reps + 3; ++r) {
    CK(hipMemsetAsync(xb, 0, PX_BYTES, 0)); CK(hipMemsetAsync(errs, 0, 8, 0));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(nblk), dim3(256), 0, 0, persist_x_at(xb), errs, sink, nblk, T, nmfma, skew, xl);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned e[2]; CK(hipMemcpy(e, errs, 8, hipMemcpyDeviceToHost)); herr += e[0]; nl2 = e[1];
    if (r >= 3) sum += ms;
  }
  printf("px_exchange nblk %d T %d nmfma %d skew %d xcd-local stores %d : %.2f us/step, errors %u, workgroups publishing through L2 %u\n", nblk, T, nmfma, skew, xl, sum / reps * 1e3f / T, herr, nl2);
  unsigned long long st[64];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(px_stamps), sizeof(st)));
  for (int b = 0; b < 4; ++b) {
    printf("  block slot %d (100 MHz ticks from exchange start):", b);
    for (int k = 0; k < 6; ++k) printf(" s%d=%lld", k, (long long)(st[b * 16 + k] - st[b * 16 + 15]));
    printf(" end=%lld\n", (long long)(st[b * 16 + 14] - st[b * 16 + 15]));
  }
  return herr ? 2 : 0;
}
