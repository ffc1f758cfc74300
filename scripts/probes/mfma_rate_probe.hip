// How fast does one wave issue v_mfma_f32_32x32x2_f32 on this part, alone and with the LDS operand feed of the MLP
// kernels?  Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 mfma_rate_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, long long* cyc, int iters) {
  __shared__ float s_w[64 * 97];
  for (int i = threadIdx.x; i < 64 * 97; i += 256) s_w[i] = 0.001f * (i & 63);
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  f32x16 a0 = {0}, a1 = {0};
  float b = 1.0f + lane;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      float w0, w1;
      if (MODE == 0) { w0 = b; w1 = b; }
      else { w0 = s_w[l31 * 97 + k + 4 * h]; w1 = s_w[(32 + l31) * 97 + k + 4 * h]; }
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, b, a0, 0, 0, 0);
      if (MODE != 2) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, b, a1, 0, 0, 0);
      else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, b, a0, 0, 0, 0);   // one dependent chain
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, int iters) {
  float* out; long long* cyc;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<blocks, 256>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> c(blocks);
  hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  const double mfmas = 64.0 * iters;            // per wave
  printf("%-34s blocks %4d  %8.1f us  shader clock ticks/MFMA %.1f  wall ns/MFMA/wave %.2f  => %.1f TFLOP/s\n", name, blocks,
         ms * 1e3, c[0] / mfmas, ms * 1e6 / mfmas, blocks * 4 * mfmas * 4096 / (ms * 1e-3) * 1e-12);
  hipFree(out); hipFree(cyc);
}

int main() {
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("reported clock %d kHz\n", clk);
  for (int blocks : {256, 512, 1024}) {
    run<0>("regs only, 2 accumulators", blocks, 200);
    run<1>("LDS feed, 2 accumulators", blocks, 200);
    run<2>("LDS feed, 1 dependent chain", blocks, 200);
  }
  return 0;
}
