// randsec.hip -- microbenchmark: rate of random 16-byte loads (one 64-byte sector each) and random
// 32-bit atomic adds as a function of the footprint (L2 4 MB/XCD, Infinity Cache 256 MB, HBM) and of the
// loads in flight per lane.  Used to size the minimizer directory.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13; return x; }

template <int INFLIGHT, int BYTES>
__global__ __launch_bounds__(64, 8) void rd_kernel(const uint4 *__restrict__ tab, uint32_t n_sectors, uint32_t iters, uint32_t *out)
{
    uint32_t s = mix(blockIdx.x * 64u + threadIdx.x + 12345u);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[INFLIGHT][BYTES / 16];
#pragma unroll
        for (int u = 0; u < INFLIGHT; u++) {
            s = mix(s + it * 0x632BE5ABu + u);
            const uint32_t sec = (uint32_t)(((uint64_t)s * n_sectors) >> 32);
#pragma unroll
            for (int q = 0; q < BYTES / 16; q++) v[u][q] = tab[(uint64_t)sec * 4 + q];
        }
#pragma unroll
        for (int u = 0; u < INFLIGHT; u++)
#pragma unroll
            for (int q = 0; q < BYTES / 16; q++) acc += v[u][q].x ^ v[u][q].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int INFLIGHT>
__global__ __launch_bounds__(64, 8) void at_kernel(uint32_t *__restrict__ tab, uint32_t n_words, uint32_t iters)
{
    uint32_t s = mix(blockIdx.x * 64u + threadIdx.x + 999u);
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < INFLIGHT; u++) {
            s = mix(s + it * 0x632BE5ABu + u);
            atomicAdd(&tab[(uint32_t)(((uint64_t)s * n_words) >> 32)], 1u);
        }
    }
}

int main()
{
    const uint64_t max_bytes = 2ull << 30;
    uint4 *tab; uint32_t *out;
    CK(hipMalloc((void **)&tab, max_bytes)); CK(hipMalloc((void **)&out, 64));
    CK(hipMemset(tab, 1, max_bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint64_t foots[] = {2ull << 20, 16ull << 20, 64ull << 20, 128ull << 20, 200ull << 20, 400ull << 20, 800ull << 20, 2ull << 30};
    const unsigned blocks = 256 * 32 * 4;   // 4 rounds of resident one-wave blocks
    printf("kind inflight bytes footprint_MB  Gaccess/s  GB/s(64B sectors)\n");
    auto run = [&](const char *kind, int inflight, int bytes, uint64_t fb, auto launch) {
        const uint32_t iters = 256 / inflight;
        launch(iters); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); launch(iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double n = (double)blocks * 64 * iters * inflight;
        printf("%s %d %d %8.0f  %8.2f  %8.1f\n", kind, inflight, bytes, fb / 1048576.0, n / ms / 1e6, n * 64 / ms / 1e6);
    };
    for (uint64_t fb : foots) {
        const uint32_t ns = (uint32_t)(fb / 64);
        run("rd", 1, 16, fb, [&](uint32_t it) { hipLaunchKernelGGL((rd_kernel<1, 16>), dim3(blocks), dim3(64), 0, 0, tab, ns, it, out); });
        run("rd", 2, 16, fb, [&](uint32_t it) { hipLaunchKernelGGL((rd_kernel<2, 16>), dim3(blocks), dim3(64), 0, 0, tab, ns, it, out); });
        run("rd", 4, 16, fb, [&](uint32_t it) { hipLaunchKernelGGL((rd_kernel<4, 16>), dim3(blocks), dim3(64), 0, 0, tab, ns, it, out); });
        run("rd", 8, 16, fb, [&](uint32_t it) { hipLaunchKernelGGL((rd_kernel<8, 16>), dim3(blocks), dim3(64), 0, 0, tab, ns, it, out); });
        run("rd", 2, 64, fb, [&](uint32_t it) { hipLaunchKernelGGL((rd_kernel<2, 64>), dim3(blocks), dim3(64), 0, 0, tab, ns, it, out); });
        run("rd", 4, 32, fb, [&](uint32_t it) { hipLaunchKernelGGL((rd_kernel<4, 32>), dim3(blocks), dim3(64), 0, 0, tab, ns, it, out); });
        run("at", 2, 4, fb, [&](uint32_t it) { hipLaunchKernelGGL((at_kernel<2>), dim3(blocks), dim3(64), 0, 0, (uint32_t *)tab, (uint32_t)(fb / 4), it); });
        run("at", 8, 4, fb, [&](uint32_t it) { hipLaunchKernelGGL((at_kernel<8>), dim3(blocks), dim3(64), 0, 0, (uint32_t *)tab, (uint32_t)(fb / 4), it); });
    }
    return 0;
}
