// fetch_calibration.hip -- what rocprofv3's FETCH_SIZE reports on gfx950 for THIS project's access pattern.
//
// MI355X_MICROARCH.md ("HBM"): FETCH_SIZE reads exactly half the bytes of a wide coalesced streaming read and "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute".  The
// traversal's pattern is a GATHER OF 64-BYTE RECORDS: every lane reads the four float4 of one record at an unrelated address
// (rt_kernels.hip trace_loop, the vector-memory path).  Three launches with known byte counts:
//   stream   N bytes read once, 16 B per lane, consecutive lanes consecutive addresses        (the guide's calibration case)
//   gather   R records of 64 B, each read exactly once, in a random order                     (the traversal's pattern, cold)
//   regather the same gather over a table of 32 MiB, eight times over                         (the traversal's pattern, cache-resident)
// Build and run (tools/fetch_calibration.sh does both under rocprofv3 --pmc FETCH_SIZE):
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calibration tools/fetch_calibration.hip && ./fetch_calibration
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

__global__ void stream_kernel(const float4* __restrict__ src, size_t n, float* __restrict__ sink)
{
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) *sink = acc;                         // (keeps the loads alive)
}

__global__ void gather_kernel(const float4* __restrict__ records, const uint32_t* __restrict__ index, size_t n, float* __restrict__ sink)
{
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4* r = records + (size_t)index[i] * 4;       // one 64-B record per lane, four 16-B loads: the traversal's fetch
        const float4 a = r[0], b = r[1], c = r[2], d = r[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) *sink = acc;
}

int main()
{
    const size_t stream_bytes = (size_t)1 << 30;                // 1 GiB, read once
    const size_t cold_records = (size_t)1 << 24;                // 16 Mi records = 1 GiB, each read once
    const size_t warm_records = (size_t)1 << 19, warm_passes = 8;   // 32 MiB table (fits every L2 + the Infinity Cache), 8 passes
    float4* table = nullptr; uint32_t* index = nullptr; float* sink = nullptr;
    CK(hipMalloc((void**)&table, stream_bytes));
    CK(hipMemset(table, 0, stream_bytes));
    CK(hipMalloc((void**)&index, cold_records * sizeof(uint32_t)));
    CK(hipMalloc((void**)&sink, 4));
    std::vector<uint32_t> perm(cold_records);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 gen(1);
    for (size_t i = perm.size() - 1; i > 0; i--) std::swap(perm[i], perm[gen() % (i + 1)]);
    CK(hipMemcpy(index, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
    // a 1 GiB streamed write first: nothing of the table is left in any cache
    CK(hipMemset(table, 0, stream_bytes));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, table, stream_bytes / 16, sink);
    CK(hipDeviceSynchronize());
    CK(hipMemset(table, 0, stream_bytes));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(gather_kernel, dim3(4096), dim3(256), 0, 0, table, index, cold_records, sink);
    CK(hipDeviceSynchronize());
    // warm: indices into the first 32 MiB only, several passes in one launch
    std::vector<uint32_t> w(warm_records * warm_passes);
    for (size_t p = 0; p < warm_passes; p++) {
        std::vector<uint32_t> q(warm_records);
        std::iota(q.begin(), q.end(), 0u);
        for (size_t i = q.size() - 1; i > 0; i--) std::swap(q[i], q[gen() % (i + 1)]);
        std::copy(q.begin(), q.end(), w.begin() + p * warm_records);
    }
    CK(hipMemcpy(index, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(gather_kernel, dim3(4096), dim3(256), 0, 0, table, index, w.size(), sink);
    CK(hipDeviceSynchronize());
    printf("{\"stream_bytes\": %zu, \"gather_cold_record_bytes\": %zu, \"gather_cold_index_bytes\": %zu, \"gather_warm_record_bytes_algorithmic\": %zu, "
           "\"gather_warm_table_bytes\": %zu, \"gather_warm_index_bytes\": %zu}\n",
           stream_bytes, cold_records * 64, cold_records * 4, w.size() * 64, warm_records * 64, w.size() * 4);
    return 0;
}
