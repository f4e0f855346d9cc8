// kernarg_probe.hip -- does this runtime take kernel arguments above 4 KB (a RenderParams with 128 frames is 11 KB)?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
template <int N> struct Big { float v[N]; int* out; };
template <int N> __global__ void k(const Big<N> p) { if (threadIdx.x == 0) p.out[blockIdx.x] = (int)p.v[blockIdx.x * 7 % N]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <int N> int run(int* d)
{
    Big<N> b; for (int i = 0; i < N; i++) b.v[i] = (float)i; b.out = d;
    hipLaunchKernelGGL(k<N>, dim3(64), dim3(64), 0, 0, b);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    int h[64]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    bool ok = true; for (int i = 0; i < 64; i++) ok = ok && h[i] == i * 7 % N;
    double t = now();
    for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(k<N>, dim3(64), dim3(64), 0, 0, b);
    double issue = (now() - t) / 2000 * 1e6;
    CK(hipDeviceSynchronize());
    double total = (now() - t) / 2000 * 1e6;
    printf("kernarg %6zu B: %s, issue %.2f us per launch, %.2f us per launch drained\n", sizeof(Big<N>), ok ? "ok" : "WRONG", issue, total);
    return 0;
}
int main() { int* d; CK(hipMalloc((void**)&d, 256)); return run<64>(d) || run<700>(d) || run<1000>(d) || run<3000>(d) || run<4000>(d) || run<8000>(d); }
