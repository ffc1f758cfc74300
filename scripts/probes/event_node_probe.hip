// Can a kernel inside a captured hipGraph be timed with external event-record nodes (ROCm 7.2, gfx950)?
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/event_node_probe.hip -o /tmp/event_node_probe && /tmp/event_node_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); } } while (0)
__global__ void spin(float* p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
int main() {
  int rv = 0; CK(hipRuntimeGetVersion(&rv)); printf("HIP runtime version %d\n", rv);
  float* d; CK(hipMalloc(&d, 4096));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int mode = 0; mode < 3; ++mode) {
    // mode 0: events created with default flags, capture mode global; 1: thread-local capture; 2: relaxed
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipStreamCaptureMode cm = mode == 0 ? hipStreamCaptureModeGlobal : mode == 1 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed;
    CK(hipStreamBeginCapture(s, cm));
    hipError_t ea = hipEventRecordWithFlags(a, s, hipEventRecordExternal);
    spin<<<1, 256, 0, s>>>(d, 200000);
    hipError_t eb = hipEventRecordWithFlags(b, s, hipEventRecordExternal);
    hipGraph_t g; CK(hipStreamEndCapture(s, &g));
    printf("mode %d: record a -> %s, record b -> %s\n", mode, hipGetErrorString(ea), hipGetErrorString(eb));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("  graph nodes: %zu\n", nn);
    hipGraphExec_t x; CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) {
      CK(hipGraphLaunch(x, s)); CK(hipStreamSynchronize(s));
      float ms = -1; hipError_t ee = hipEventElapsedTime(&ms, a, b);
      printf("  replay %d: elapsed -> %s, %.3f ms\n", r, hipGetErrorString(ee), ms);
    }
    CK(hipGraphExecDestroy(x)); CK(hipGraphDestroy(g)); CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    (void)hipGetLastError();
  }
  // event-record nodes added BY HAND to the graph under capture (hipStreamGetCaptureInfo_v2 + hipGraphAddEventRecordNode +
  // hipStreamUpdateCaptureDependencies): the route for a runtime whose hipEventRecordWithFlags refuses the external flag
  {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    auto add = [&](hipEvent_t ev) {
      hipStreamCaptureStatus st; unsigned long long id = 0; hipGraph_t cg = nullptr; const hipGraphNode_t* deps = nullptr; size_t nd = 0;
      hipError_t e = hipStreamGetCaptureInfo_v2(s, &st, &id, &cg, &deps, &nd);
      if (e != hipSuccess) return e;
      hipGraphNode_t node;
      e = hipGraphAddEventRecordNode(&node, cg, deps, nd, ev);
      if (e != hipSuccess) return e;
      return hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies);
    };
    hipError_t ea = add(a);
    spin<<<1, 256, 0, s>>>(d, 200000);
    hipError_t eb = add(b);
    hipGraph_t g; CK(hipStreamEndCapture(s, &g));
    printf("hand-added nodes during capture: %s / %s\n", hipGetErrorString(ea), hipGetErrorString(eb));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("  graph nodes: %zu\n", nn);
    hipGraphExec_t x; CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) {
      CK(hipGraphLaunch(x, s)); CK(hipStreamSynchronize(s));
      float ms = -1; hipError_t ee = hipEventElapsedTime(&ms, a, b);
      printf("  replay %d: elapsed -> %s, %.3f ms\n", r, hipGetErrorString(ee), ms);
    }
    (void)hipGetLastError();
  }
  // plain (non-external) record inside capture, for comparison
  {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    hipError_t ea = hipEventRecord(a, s);
    spin<<<1, 256, 0, s>>>(d, 200000);
    hipError_t eb = hipEventRecord(b, s);
    hipGraph_t g; CK(hipStreamEndCapture(s, &g));
    printf("plain record in capture: %s / %s\n", hipGetErrorString(ea), hipGetErrorString(eb));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("  graph nodes: %zu\n", nn);
    CK(hipGraphDestroy(g));
  }
  // explicit event-record nodes added to a graph built by hand
  {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    hipGraphNode_t na, nk, nb;
    CK(hipGraphAddEventRecordNode(&na, g, nullptr, 0, a));
    hipKernelNodeParams kp = {}; int n = 200000; void* args[2] = {&d, &n};
    kp.func = (void*)spin; kp.gridDim = dim3(1); kp.blockDim = dim3(256); kp.kernelParams = args;
    CK(hipGraphAddKernelNode(&nk, g, &na, 1, &kp));
    CK(hipGraphAddEventRecordNode(&nb, g, &nk, 1, b));
    hipGraphExec_t x; CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) {
      CK(hipGraphLaunch(x, s)); CK(hipStreamSynchronize(s));
      float ms = -1; hipError_t ee = hipEventElapsedTime(&ms, a, b);
      printf("  hand-built replay %d: elapsed -> %s, %.3f ms\n", r, hipGetErrorString(ee), ms);
    }
  }
  return 0;
}
