// Launch / copy / sync latencies that bound a SMALL search call (the reference's 4 queries per batch_search):
// what do an upload, a kernel, a second dependent kernel, a device-wide fence and the wake-up of the host cost?
// build: hipcc --offload-arch=gfx950 -O3 -w scripts/latency_lab.hip -o build/latency_lab   run: build/latency_lab
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <functional>

__global__ void k_empty(uint32_t* p) {
    if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1;
}
__global__ void k_read3(const uint32_t* in, uint32_t* out) {  // a dependent chain of 3 loads from `in`
    uint32_t a = in[threadIdx.x & 3];
    uint32_t b = in[8 + (a & 3)];
    uint32_t c = in[16 + (b & 3)];
    if (threadIdx.x == 0) out[blockIdx.x] = c;
}
__global__ void k_fence(uint32_t* p) {
    p[blockIdx.x * 64 + (threadIdx.x & 63)] = threadIdx.x;
    __threadfence();
    if (threadIdx.x == 0) atomicAdd(p + 65536, 1u);
}
__global__ void k_spin(uint32_t* p, int iters) {
    uint32_t x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
    if (x == 42) p[0] = x;
}

static double time_us(int reps, const std::function<void()>& f) {
    for (int i = 0; i < 20; ++i) f();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) f();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main() {
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    uint32_t *d = nullptr, *h_pin = nullptr, *h_map = nullptr, *d_map = nullptr;
    (void)hipMalloc(&d, 1 << 20);
    (void)hipMemset(d, 0, 1 << 20);
    (void)hipHostMalloc(&h_pin, 1 << 16, hipHostMallocDefault);
    (void)hipHostMalloc(&h_map, 1 << 16, hipHostMallocMapped);
    (void)hipHostGetDevicePointer((void**)&d_map, h_map, 0);
    for (int i = 0; i < 1024; ++i) h_pin[i] = h_map[i] = i & 3;
    const int R = 2000;
    auto sync = [&] { (void)hipStreamSynchronize(st); };
    auto row = [&](const char* what, const std::function<void()>& f) { printf("%-62s %7.2f us\n", what, time_us(R, f)); };
    row("stream sync on an idle stream", [&] { sync(); });
    row("empty kernel (1 wg) + sync", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, d); sync(); });
    row("empty kernel (16 wg x 512) + sync", [&] { hipLaunchKernelGGL(k_empty, dim3(16), dim3(512), 0, st, d); sync(); });
    row("empty kernel (256 wg x 512) + sync", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, d); sync(); });
    row("two dependent empty kernels + sync", [&] { hipLaunchKernelGGL(k_empty, dim3(16), dim3(512), 0, st, d); hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, st, d); sync(); });
    row("H2D 512 B from pinned (async) + sync", [&] { (void)hipMemcpyAsync(d, h_pin, 512, hipMemcpyHostToDevice, st); sync(); });
    row("H2D 512 B + empty kernel + sync", [&] { (void)hipMemcpyAsync(d, h_pin, 512, hipMemcpyHostToDevice, st); hipLaunchKernelGGL(k_empty, dim3(16), dim3(512), 0, st, d); sync(); });
    row("H2D 4 KB + empty kernel + sync", [&] { (void)hipMemcpyAsync(d, h_pin, 4096, hipMemcpyHostToDevice, st); hipLaunchKernelGGL(k_empty, dim3(16), dim3(512), 0, st, d); sync(); });
    row("kernel reading 3 dependent words from DEVICE memory + sync", [&] { hipLaunchKernelGGL(k_read3, dim3(16), dim3(512), 0, st, d + 1024, d); sync(); });
    row("kernel reading 3 dependent words from MAPPED HOST memory + sync", [&] { hipLaunchKernelGGL(k_read3, dim3(16), dim3(512), 0, st, d_map, d); sync(); });
    row("kernel writing its result to MAPPED HOST memory + sync", [&] { hipLaunchKernelGGL(k_read3, dim3(16), dim3(512), 0, st, d + 1024, d_map + 2048); sync(); });
    row("kernel + D2H 512 B to pinned (async) + sync", [&] { hipLaunchKernelGGL(k_empty, dim3(16), dim3(512), 0, st, d); (void)hipMemcpyAsync(h_pin + 4096, d, 512, hipMemcpyDeviceToHost, st); sync(); });
    row("kernel with a __threadfence per thread, 16 wg + sync", [&] { hipLaunchKernelGGL(k_fence, dim3(16), dim3(512), 0, st, d + 8192); sync(); });
    row("kernel with a __threadfence per thread, 256 wg + sync", [&] { hipLaunchKernelGGL(k_fence, dim3(256), dim3(512), 0, st, d + 8192); sync(); });
    row("kernel of ~6000 dependent integer ops per thread + sync", [&] { hipLaunchKernelGGL(k_spin, dim3(16), dim3(512), 0, st, d, 6000); sync(); });
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    row("event + kernel + event + sync", [&] { (void)hipEventRecord(e0, st); hipLaunchKernelGGL(k_empty, dim3(16), dim3(512), 0, st, d); (void)hipEventRecord(e1, st); sync(); });
    return 0;
}
