// null_stream_probe.hip -- what work on the NULL stream costs on this runtime when blocking streams exist (rt_render_overlapped
// keeps two): launch + synchronise of an empty kernel on the null stream, a synchronous 4-byte hipMemcpy, and hipDeviceSynchronize,
// (a) with no other stream, (b) with two idle blocking streams, (c) right after work on them, (d) with two non-blocking streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void spin_kernel(int* p, int n) { int a = 0; for (int i = 0; i < n; i++) a += i * i; if (a == 12345) *p = a; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int measure(const char* what, hipStream_t s0, hipStream_t s1, int* d, int* h)
{
    const int n = 200;
    for (int i = 0; i < 20; i++) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, d); CK(hipStreamSynchronize(0)); }
    double t = now();
    for (int i = 0; i < n; i++) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, d); CK(hipStreamSynchronize(0)); }
    double launch_sync = (now() - t) / n * 1e6;
    t = now();
    for (int i = 0; i < n; i++) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, d); CK(hipDeviceSynchronize()); }
    double launch_devsync = (now() - t) / n * 1e6;
    t = now();
    for (int i = 0; i < n; i++) CK(hipMemcpy(h, d, 4, hipMemcpyDeviceToHost));
    double memcpy_us = (now() - t) / n * 1e6;
    double pair = -1, pair_null = -1, pair_then_null = -1;
    if (s0) {
        for (int i = 0; i < 20; i++) { hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s0, d, 2000); hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s1, d, 2000); CK(hipDeviceSynchronize()); }
        t = now();
        for (int i = 0; i < n; i++) { hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s0, d, 2000); hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s1, d, 2000); CK(hipDeviceSynchronize()); }
        pair = (now() - t) / n * 1e6;
        t = now();
        for (int i = 0; i < n; i++) { hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s0, d, 2000); hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s1, d, 2000); CK(hipMemcpy(h, d, 4, hipMemcpyDeviceToHost)); }
        pair_then_null = (now() - t) / n * 1e6;
    }
    t = now();
    for (int i = 0; i < n; i++) { hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, 0, d, 2000); hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, 0, d, 2000); CK(hipDeviceSynchronize()); }
    pair_null = (now() - t) / n * 1e6;
    printf("%-44s null launch+streamsync %7.1f us  null launch+devsync %7.1f us  sync memcpy 4 B %7.1f us  two spin kernels: on the two streams + devsync %7.1f, + sync memcpy %7.1f, both on null + devsync %7.1f\n",
           what, launch_sync, launch_devsync, memcpy_us, pair, pair_then_null, pair_null);
    return 0;
}
int main()
{
    int *d = nullptr, *h = nullptr;
    CK(hipMalloc((void**)&d, 4)); CK(hipHostMalloc((void**)&h, 4));
    if (measure("no other stream", nullptr, nullptr, d, h)) return 1;
    hipStream_t nb[2], bl[2];
    for (auto& s : nb) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (measure("two NON-blocking streams", nb[0], nb[1], d, h)) return 1;
    for (auto& s : bl) CK(hipStreamCreateWithFlags(&s, hipStreamDefault));
    if (measure("two BLOCKING streams (+ the non-blocking)", bl[0], bl[1], d, h)) return 1;
    for (auto& s : nb) CK(hipStreamDestroy(s));
    if (measure("two BLOCKING streams only", bl[0], bl[1], d, h)) return 1;
    for (auto& s : bl) CK(hipStreamDestroy(s));
    if (measure("streams destroyed again", nullptr, nullptr, d, h)) return 1;
    return 0;
}
