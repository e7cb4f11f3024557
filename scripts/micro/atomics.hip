// atomics.hip -- microbenchmark (round 4): what a 32-bit global atomicAdd costs on MI355X as a function of how the lanes of
// ONE wave instruction spread over 64-byte lines, for the cluster scan's counters (profiles/r04_ab_log.md 1).
//   G lanes of a wave share a random line, A of them active (consecutive dwords); 64 / G lines per instruction.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/atomics.hip -o /tmp/atomics && /tmp/atomics
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13; return x; }

// MODE 0: one instruction per iteration; 1: the active lanes' adds as TWO instructions (even dwords, then odd dwords of the
// same line); 2: u64 adds (lane pairs -> one 8-byte add by the even lane); 3: workgroup-scope adds
template <int G, int A, int MODE>
__global__ __launch_bounds__(64, 8) void at_kernel(uint32_t *__restrict__ tab, uint32_t n_lines, uint32_t iters)
{
    const uint32_t t = threadIdx.x, grp = t / G, j = t % G;
    uint32_t s = mix((blockIdx.x * 64u + grp) * 7919u + 999u);
    for (uint32_t it = 0; it < iters; it++) {
        s = mix(s + it * 0x632BE5ABu);
        const uint32_t line = (uint32_t)(((uint64_t)s * n_lines) >> 32);
        uint32_t *p = tab + (uint64_t)line * 16 + j;
        if (j < A) {
            if (MODE == 0) atomicAdd(p, 1u);
            else if (MODE == 1) { if (!(j & 1)) atomicAdd(p, 1u); }
            else if (MODE == 2) { if (!(j & 1)) atomicAdd(reinterpret_cast<unsigned long long *>(p), 0x100000001ull); }
            else __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (MODE == 1 && j < A && (j & 1)) atomicAdd(p, 1u);
    }
}
// every wave walks its own stretch of lines in order (what a binned read set does to a locus: the same lines again and again
// from neighbouring waves is REP > 1: REP consecutive blocks share a stretch)
template <int G, int A>
__global__ __launch_bounds__(64, 8) void seq_kernel(uint32_t *__restrict__ tab, uint32_t n_lines, uint32_t iters, uint32_t rep)
{
    const uint32_t t = threadIdx.x, grp = t / G, j = t % G;
    const uint32_t base = mix((blockIdx.x / rep) * 7919u + 17u) % (n_lines - iters * (64 / G) - 1);
    for (uint32_t it = 0; it < iters; it++) {
        uint32_t *p = tab + (uint64_t)(base + it * (64 / G) + grp) * 16 + j;
        if (j < A) atomicAdd(p, 1u);
    }
}

int main(int argc, char **argv)
{
    const uint64_t bytes = (argc > 1 ? atoll(argv[1]) : 40) << 20;
    uint32_t *tab;
    CK(hipMalloc((void **)&tab, bytes));
    CK(hipMemset(tab, 0, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned blocks = 256 * 32 * 4;
    const uint32_t iters = 128, n_lines = (uint32_t)(bytes / 64);
    printf("footprint %llu MB, %u one-wave blocks x %u iterations\n", (unsigned long long)(bytes >> 20), blocks, iters);
    printf("%-34s %10s %12s %12s %12s\n", "pattern", "ms", "G instr/s", "G lines/s", "G lane-ops/s");
    auto report = [&](const char *name, int G, int A, int instr_per_it, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double n_it = (double)blocks * iters;
        printf("%-34s %10.3f %12.2f %12.2f %12.2f\n", name, ms, n_it * instr_per_it / ms * 1e-6, n_it * (64 / G) / ms * 1e-6, n_it * (64 / G) * A / ms * 1e-6);
    };
#define RUN(G, A, M, name, ipi) report(name, G, A, ipi, [&] { hipLaunchKernelGGL((at_kernel<G, A, M>), dim3(blocks), dim3(64), 0, 0, tab, n_lines, iters); })
    RUN(1, 1, 0, "64 lines x 1 lane", 1);
    RUN(2, 2, 0, "32 lines x 2 lanes", 1);
    RUN(4, 4, 0, "16 lines x 4 lanes", 1);
    RUN(8, 8, 0, "8 lines x 8 lanes", 1);
    RUN(16, 16, 0, "4 lines x 16 lanes", 1);
    RUN(16, 9, 0, "4 lines x 9 lanes", 1);
    RUN(16, 2, 0, "4 lines x 2 lanes", 1);
    RUN(16, 1, 0, "4 lines x 1 lane", 1);
    RUN(64, 16, 0, "1 line x 16 lanes", 1);
    RUN(64, 1, 0, "1 line x 1 lane", 1);
    RUN(16, 16, 1, "4 lines x 16 lanes in 2 instr", 2);
    RUN(16, 2, 1, "4 lines x 2 lanes in 2 instr", 2);
    RUN(16, 16, 2, "4 lines x 8 u64 adds", 1);
    RUN(1, 1, 3, "64 lines x 1 lane, wg scope", 1);
    RUN(16, 16, 3, "4 lines x 16 lanes, wg scope", 1);
#define SEQ(G, A, R, name) report(name, G, A, 1, [&] { hipLaunchKernelGGL((seq_kernel<G, A>), dim3(blocks), dim3(64), 0, 0, tab, n_lines, iters, R); })
    SEQ(16, 16, 1, "sequential 4 lines x 16");
    SEQ(16, 16, 8, "sequential 4 x 16, 8 blocks share");
    SEQ(16, 16, 64, "sequential 4 x 16, 64 blocks share");
    SEQ(1, 1, 1, "sequential 64 lines x 1");
    return 0;
}
