// Diagnostic (round 3): what rate does ONE workgroup per CU reach when every CU streams the same L2-resident image (the codebook /
// W_pre fragment images of the fused VQ kernel: ~200 KB per CU), and does the order in which the CUs walk it matter?
//   hipcc --offload-arch=gfx950 -O3 gpurun_tools/l2_stream_bench.hip -o gpurun_tools/l2_stream_bench && ./gpurun_tools/l2_stream_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int NT, int MODE>   // MODE 0: same order everywhere; 1: start rotated by workgroup; 2: rotated + wave-interleaved blocks
__global__ __launch_bounds__(NT) void stream_kernel(const float4* __restrict__ img, int nblk /* 1 KB blocks */, float* out) {
  constexpr int NW = NT / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = nblk / NW;                    // blocks per wave
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int rot = MODE == 0 ? 0 : (int)((blockIdx.x * 2654435761u) >> 8) % per;
  constexpr int U = 16;
  for (int j0 = 0; j0 < per; j0 += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int j = (j0 + u + rot) % per;
      const int blk = MODE == 2 ? j * NW + wave : wave * per + j;
      v[u] = img[(size_t)blk * 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[blockIdx.x * NT + threadIdx.x] = acc.x;
}

template <int NT, int MODE>
static void run(const float4* img, int nblk, float* out, int nwg, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((stream_kernel<NT, MODE>), dim3(nwg), dim3(NT), 0, 0, img, nblk, out);
  hipEventRecord(e0, 0);
  const int reps = 200;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<NT, MODE>), dim3(nwg), dim3(NT), 0, 0, img, nblk, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, kb = nblk;
  printf("%-28s threads %4d  %4d KB per WG, %3d WGs: %7.2f us per launch  -> %6.1f GB/s per CU (incl. launch gap), %5.1f B/clk at 2.1 GHz\n", name, NT, nblk,
         nwg, us, kb * 1024 / (us * 1e-6) / 1e9, kb * 1024 / (us * 1e-6) / 2.1e9);
}

int main() {
  const int maxblk = 512;
  float4* img; float* out;
  hipMalloc(&img, (size_t)maxblk * 1024);
  hipMalloc(&out, 256 * 1024 * 4);
  std::vector<float> h((size_t)maxblk * 256, 1.0f);
  hipMemcpy(img, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int nblk : {64, 128, 192, 256, 512}) {
    run<256, 0>(img, nblk, out, 256, "same order");
    run<256, 1>(img, nblk, out, 256, "rotated start");
    run<256, 2>(img, nblk, out, 256, "rotated, wave-interleaved");
    run<512, 0>(img, nblk, out, 256, "same order");
    run<512, 1>(img, nblk, out, 256, "rotated start");
    run<512, 2>(img, nblk, out, 256, "rotated, wave-interleaved");
    run<1024, 1>(img, nblk, out, 256, "rotated start");
  }
  // an empty-ish launch for the gap
  run<256, 0>(img, 16, out, 256, "16 KB (launch gap probe)");
  return 0;
}
