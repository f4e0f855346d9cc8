// l1_calibration.hip -- how many vector-L1 (TCP) accesses a CU of gfx950 sustains per clock for THIS project's fetch.
//
// bench.py prices the primary kernel's `roofline.fractions.l1` as TCP_TOTAL_CACHE_ACCESSES per second against a peak.  Rounds
// 1-4 ASSUMED that peak (256 CUs x one 64-B access per clock x 2.4 GHz = 39.3 TB/s); MI355X_MICROARCH.md states no vector-L1
// rate.  This measures it, the way tools/fetch_calibration.hip measured what FETCH_SIZE reports: the traversal's vector
// fetch is a GATHER OF 64-BYTE RECORDS -- every lane reads the four float4 of one record (four global_load_dwordx4 per wave
// and iteration, rt_kernels.hip trace_loop) -- and the number of DIFFERENT records the 64 lanes of a wave-instruction hold
// varies from 1 to 64.  Here: the same four loads from a table that stays L1-resident (256 records = 16 KiB of the 32 KiB), with
// exactly D = 1, 2, 4, 8, 16, 32, 64 different records per wave-instruction, 8 waves per SIMD (the render kernel's residency),
// every CU busy; each wave also reads the shader clock around its loop.  One launch per D, so that a rocprofv3 --pmc pass gives
// TCP_TOTAL_CACHE_ACCESSES per launch (tools/l1_calibration.sh divides by the instruction count printed here).
//   hipcc --offload-arch=gfx950 -O3 -o l1_calibration tools/l1_calibration.hip && ./l1_calibration
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

constexpr int kRecords = 256;           // 16 KiB table: resident in every CU's 32 KiB L1 after the first touch
constexpr int kBlock = 256;
constexpr int kUnroll = 2;              // (the asm block below is written for 2: 8 loads = 32 of the 64 VGPRs a wave has at 8 waves per SIMD) independent record fetches in flight per lane (the render kernel has one; the ceiling needs more)

// D different records per wave-instruction: lane l belongs to group l % D; the group's record changes every iteration
template <int D>
__global__ __launch_bounds__(kBlock, 8) void gather_l1_kernel(const float4* __restrict__ table, int iters, float* __restrict__ sink,
                                                             unsigned long long* __restrict__ cycles, unsigned long long* __restrict__ ticks)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const unsigned group = (unsigned)(lane % D);
    unsigned rec = (group * 37u + (unsigned)wave * 11u) & (kRecords - 1);
    float acc = 0.0f;
    // touch the whole table once so that the timed loop never misses
    for (int i = lane; i < kRecords * 4; i += 64) acc += table[i].x;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();    // constant-rate counter (hipDeviceAttributeWallClockRate)
    const unsigned long long t0 = __builtin_readcyclecounter();        // s_memtime
    for (int it = 0; it < iters; it++) {
        float4 v[kUnroll][4];
        const float4* p[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            // D distinct values per wave: group * 37 (odd) is a bijection mod 256, the other terms are the same for every lane
            const unsigned r = (rec + (unsigned)u * 64u) & (kRecords - 1);
            p[u] = table + (size_t)r * 4;
        }
        // The instructions themselves (the compiler narrows a float4 load of which one component is used to a dword load), all
        // eight and their wait in ONE statement with early-clobber results: the compiler must not reuse a register an
        // in-flight load still writes, and it cannot see that these loads complete asynchronously.
        asm volatile(
            "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:16\n\t"
            "global_load_dwordx4 %2, %8, off offset:32\n\tglobal_load_dwordx4 %3, %8, off offset:48\n\t"
            "global_load_dwordx4 %4, %9, off\n\tglobal_load_dwordx4 %5, %9, off offset:16\n\t"
            "global_load_dwordx4 %6, %9, off offset:32\n\tglobal_load_dwordx4 %7, %9, off offset:48\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[1][2]), "=&v"(v[1][3])
            : "v"(p[0]), "v"(p[1])
            : "memory");
#pragma unroll
        for (int u = 0; u < kUnroll; u++) acc += v[u][0].x + v[u][1].y + v[u][2].z + v[u][3].w;
        rec = (rec + 3u) & (kRecords - 1);                     // every group moves on by the same step: the groups stay D different records
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { cycles[wave] = t1 - t0; ticks[wave] = r1 - r0; }
    if (acc == 12345.678f) *sink = acc;
}

template <int D>
int run(const float4* table, float* sink, unsigned long long* d_cycles, unsigned long long* d_ticks, int blocks, int iters, bool last)
{
    const int waves = blocks * (kBlock / 64);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(gather_l1_kernel<D>, dim3(blocks), dim3(kBlock), 0, 0, table, iters / 8, sink, d_cycles, d_ticks);   // warm-up (clock, code)
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(gather_l1_kernel<D>, dim3(blocks), dim3(kBlock), 0, 0, table, iters, sink, d_cycles, d_ticks);
    CK(hipEventRecord(b, 0));
    CK(hipDeviceSynchronize());
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> cyc((size_t)waves);
    CK(hipMemcpy(cyc.data(), d_cycles, cyc.size() * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> tick((size_t)waves);
    CK(hipMemcpy(tick.data(), d_ticks, tick.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0.0, mean_ticks = 0.0; unsigned long long mx = 0;
    for (auto c : cyc) { mean += (double)c; mx = c > mx ? c : mx; }
    for (auto c : tick) mean_ticks += (double)c;
    mean /= (double)waves; mean_ticks /= (double)waves;
    const double loads = (double)waves * iters * kUnroll * 4;                       // global_load_dwordx4 wave-instructions of the timed launch
    printf("%s{\"distinct_records_per_wave_instruction\": %d, \"timed_launch_ms\": %.5f, \"wave_load_instructions\": %.0f, "
           "\"mean_wave_loop_memtime\": %.1f, \"max_wave_loop_memtime\": %llu, \"mean_wave_loop_realtime_ticks\": %.1f, \"waves\": %d, \"iters\": %d}%s\n",
           D == 1 ? "{\"launches\": [" : "", D, ms, loads, mean, mx, mean_ticks, waves, iters, last ? "]}" : ",");
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;                                 // 8 workgroups of 4 waves per CU = 8 waves per SIMD, one round
    const int iters = 4096;
    float4* table = nullptr; float* sink = nullptr; unsigned long long* d_cycles = nullptr;
    CK(hipMalloc((void**)&table, (size_t)kRecords * 64));
    CK(hipMemset(table, 0, (size_t)kRecords * 64));
    CK(hipMalloc((void**)&sink, 4));
    CK(hipMalloc((void**)&d_cycles, (size_t)blocks * (kBlock / 64) * 8));
    unsigned long long* d_ticks = nullptr;
    CK(hipMalloc((void**)&d_ticks, (size_t)blocks * (kBlock / 64) * 8));
    int wall_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    fprintf(stderr, "l1_calibration: %d CUs, clockRate %d kHz, %d blocks x %d threads, %d iterations x %d records x 4 loads per lane\n",
            cus, prop.clockRate, blocks, kBlock, iters, kUnroll);
    if (run<1>(table, sink, d_cycles, d_ticks, blocks, iters, false)) return 1;
    if (run<2>(table, sink, d_cycles, d_ticks, blocks, iters, false)) return 1;
    if (run<4>(table, sink, d_cycles, d_ticks, blocks, iters, false)) return 1;
    if (run<8>(table, sink, d_cycles, d_ticks, blocks, iters, false)) return 1;
    if (run<16>(table, sink, d_cycles, d_ticks, blocks, iters, false)) return 1;
    if (run<32>(table, sink, d_cycles, d_ticks, blocks, iters, false)) return 1;
    if (run<64>(table, sink, d_cycles, d_ticks, blocks, iters, true)) return 1;
    fprintf(stderr, "l1_calibration: cus=%d clock_khz=%d wall_clock_khz=%d\n", cus, prop.clockRate, wall_khz);
    return 0;
}
