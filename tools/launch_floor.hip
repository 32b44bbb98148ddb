// What does a kernel boundary cost on MI355X?  N dependent launches of a trivial kernel on one stream, issued (a) directly, (b) as a
// captured hipGraph; the kernel either touches no memory, or reads + writes one cache line per block, or streams a 4 MB buffer.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__global__ void k_empty(float* p, int n) { if (n < 0) p[0] = 1.f; }
__global__ void k_line(float* p, int n) { p[blockIdx.x * 32 + (threadIdx.x & 31)] += 1.f; }
__global__ void k_stream(float* p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] += 1.f;
}

template <typename F> void bench(const char* name, F launch, int n) {
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) launch(st);
    hipStreamSynchronize(st);
    // (a) direct
    auto h0 = std::chrono::steady_clock::now();
    hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) launch(st);
    hipEventRecord(e1, st);
    auto h1 = std::chrono::steady_clock::now();
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double host_us = std::chrono::duration<double, std::micro>(h1 - h0).count() / n;
    // (b) graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) launch(st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float m; hipEventElapsedTime(&m, e0, e1); if (m < best) best = m;
    }
    printf("%-44s direct %6.2f us/kernel (host issue %5.2f us)   graph %6.2f us/kernel\n", name, ms * 1e3 / n, host_us, best * 1e3 / n);
}

int main() {
    float* p; hipMalloc(&p, 64 << 20); hipMemset(p, 0, 64 << 20);
    const int n = 400;
    bench("empty kernel, 1 block", [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p, 0); }, n);
    bench("empty kernel, 1024 blocks x 256", [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, p, 0); }, n);
    bench("1 line RMW per block, 256 blocks", [&](hipStream_t s) { hipLaunchKernelGGL(k_line, dim3(256), dim3(64), 0, s, p, 0); }, n);
    bench("1 line RMW per block, 2048 blocks", [&](hipStream_t s) { hipLaunchKernelGGL(k_line, dim3(2048), dim3(64), 0, s, p, 0); }, n);
    bench("stream 4 MB RMW (1024 x 256)", [&](hipStream_t s) { hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, s, p, 1 << 20); }, n);
    bench("stream 32 MB RMW (2048 x 256)", [&](hipStream_t s) { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s, p, 8 << 20); }, n);
    return 0;
}
