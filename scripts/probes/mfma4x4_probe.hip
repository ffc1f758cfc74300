// Operand / result layout of v_mfma_f32_4x4x1_16b_f32 on gfx950, read off the device:
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma4x4_probe.hip -o /tmp/mfma4x4_probe && /tmp/mfma4x4_probe
// Pass 1: A(lane) = lane, B = 1 -> every D element names its A lane; pass 2: A = 1, B(lane) = lane -> its B lane.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void probe(float* out) {
  const int lane = threadIdx.x;
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)lane, 1.f, z, 0, 0, 0);
  const f32x4 db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, (float)lane, z, 0, 0, 0);
  for (int r = 0; r < 4; ++r) { out[r * 64 + lane] = da[r]; out[256 + r * 64 + lane] = db[r]; }
}
int main() {
  float* d;
  if (hipMalloc(&d, 512 * sizeof(float)) != hipSuccess) return 1;
  probe<<<1, 64>>>(d);
  float h[512];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int r = 0; r < 4; ++r) {
    printf("reg %d: (A lane, B lane) per result lane:", r);
    for (int l = 0; l < 64; ++l) printf(" %d:(%d,%d)", l, (int)h[r * 64 + l], (int)h[256 + r * 64 + l]);
    printf("\n");
  }
  return 0;
}
