// ss_build_dev.hip -- the minimizer-paged index of ss_mini.hip built ON THE DEVICE.
//
// The host build (ss_mini.hip build_mini: minimizers, partition + sort, buckets + items, page placement) takes ~0.9 s for
// the 25 M rows of an E. coli tree on the 16 CPUs a GPU box grants -- most of the first call on a new database (the
// image is cached afterwards, strainscan_amd/db.py).  The same index, byte for byte (tests hash the exported images of
// both builds), from sorts, prefix sums and a handful of streaming kernels:
//   1  per valid row: minimizer m-mer, its offset, the partition (top 8 bits of h = mix30(minimizer)): A = part | mini | off
//   2  order (A, k-mer, row): two stable radix sorts (by k-mer, then by A) starting from row order
//   3  flags and prefix sums: distinct k-mers, minimizer groups, k-mers per group -> inline items or a bucket per
//      group -> bucket slots and page items per group, exclusive sums = their places in d_mkeys / the item list
//   4  one thread per distinct k-mer: owner row (the LAST row allowed to own it: dict overwrite at identify.py:94),
//      bucket slot + slot_of_row, or an inline item; one thread per bucket: offset mask, "multi" flag, header, item
//   5  items in page order: stable radix sort by h (items of one minimizer stay in (offset, k-mer) order)
//   6  placement.  The host puts every item, in that order, into the first page at or behind its home page that is not
//      full, inside its partition's page range.  With the items sorted by home page that is pos_i = max(8 home_i,
//      pos_(i-1) + 1) in units of slots: a prefix MAXIMUM of (8 home_i - i) (partitions kept apart by a large multiple of
//      the partition number).  Items that leave their partition's range -- a handful, if any -- are placed by the host
//      afterwards, serially, exactly as the host build does; the same checks (no run of full pages as long as
//      n_pages / 1024, last page not full) decide whether the table has to grow
//   7  pages written by one thread per item (+ slot_of_row of inline items' rows), Bloom filter, counters zeroed
// Anything unusual (no device memory, more than 2^31 rows, a HIP error) returns SS_ERANGE / SS_EHIP and the caller falls
// back to the host build.  SS_BUILD=host forces the host build.
#include "ss_common.h"
#include "ss_scan_dev.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <vector>

namespace {

constexpr int PB = 8, NP = 1 << PB;                      // partitions of the host build (page ranges)
constexpr long long PART_BIG = 1ll << 44;                // keeps the partitions apart in the prefix maximum

__global__ void valid_kernel(const uint8_t *__restrict__ flags, uint64_t n, uint32_t *__restrict__ v)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (flags[i] & SS_ROW_VALID) ? 1u : 0u;
}

// entries of the valid rows, in row order
__global__ void ents_kernel(const uint64_t *__restrict__ keys, const uint8_t *__restrict__ flags, const uint32_t *__restrict__ pos, uint64_t n, int k,
                            uint64_t *__restrict__ eA, uint64_t *__restrict__ eKey, uint32_t *__restrict__ eRow, uint32_t *__restrict__ idx)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !(flags[i] & SS_ROW_VALID)) return;
    uint32_t o;
    const uint32_t mx = ss::mini_of_key(keys[i], k, &o);
    const uint32_t p = pos[i];
    eA[p] = ((uint64_t)(ss::mix30(mx) >> (30 - PB)) << 35) | ((uint64_t)mx << 5) | o;
    eKey[p] = keys[i];
    eRow[p] = (uint32_t)i;
    idx[p] = p;
}

__global__ void gather64_kernel(const uint64_t *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t n, uint64_t *__restrict__ dst)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[j] = src[perm[j]];
}
__global__ void gather32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t n, uint32_t *__restrict__ dst)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[j] = src[perm[j]];
}

// newd[j] = a new distinct k-mer starts at j; newg[j] = a new minimizer starts at j
__global__ void flags_kernel(const uint64_t *__restrict__ sA, const uint64_t *__restrict__ sKey, uint32_t n, uint32_t *__restrict__ newd, uint32_t *__restrict__ newg)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    newd[j] = (j == 0 || sKey[j] != sKey[j - 1]) ? 1u : 0u;
    newg[j] = (j == 0 || (sA[j] >> 5) != (sA[j - 1] >> 5)) ? 1u : 0u;
}

// dStart[d] = first row of distinct k-mer d; gStartD[g] = first distinct k-mer of group g (inclusive sums in dIdx1 / gIdx1)
__global__ void starts_kernel(const uint32_t *__restrict__ newd, const uint32_t *__restrict__ newg, const uint32_t *__restrict__ dIdx1,
                              const uint32_t *__restrict__ gIdx1, uint32_t n, uint32_t *__restrict__ dStart, uint32_t *__restrict__ gStartD)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (newd[j]) dStart[dIdx1[j] - 1] = j;
    if (newg[j]) gStartD[gIdx1[j] - 1] = dIdx1[j] - 1;
}

// k-mers of the minimizers that own one or two (choose_inline_max); a grid of a few hundred workgroups strides over the
// groups: one atomic per wave on ONE word is cheap only when the waves are few
__global__ __launch_bounds__(256) void small_groups_kernel(const uint32_t *__restrict__ gStartD, uint32_t n_groups, unsigned long long *__restrict__ out)
{
    unsigned long long sum = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t nd = gStartD[g + 1] - gStartD[g];
        sum += nd <= 2u ? nd : 0u;
    }
    for (int o = 32; o; o >>= 1) sum += __shfl_down(sum, o, 64);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(out, sum);
}

__global__ void group_sizes_kernel(const uint32_t *__restrict__ gStartD, uint32_t n_groups, uint32_t inline_max, uint32_t *__restrict__ gslots,
                                   uint32_t *__restrict__ gitems)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const uint32_t nd = gStartD[g + 1] - gStartD[g];
    gslots[g] = nd > inline_max ? 1u + nd : 0u;
    gitems[g] = nd > inline_max ? 1u : nd;
}

struct Items { uint32_t *h, *lo, *e0, *e1; uint16_t *mid; uint8_t *hi8; };

// one thread per distinct k-mer
__global__ void distinct_kernel(const uint64_t *__restrict__ sA, const uint64_t *__restrict__ sKey, const uint32_t *__restrict__ sRow,
                                const uint8_t *__restrict__ flags, int upper_keys, const uint32_t *__restrict__ dStart, const uint32_t *__restrict__ gIdx1,
                                const uint32_t *__restrict__ gStartD, const uint32_t *__restrict__ gSlotBase, const uint32_t *__restrict__ gItemBase,
                                uint32_t n_dist, uint32_t inline_max, uint64_t *__restrict__ mkeys, uint32_t *__restrict__ slot_of_row,
                                uint8_t *__restrict__ row_valid, Items it, unsigned long long *__restrict__ orphans)
{
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_dist) return;
    const uint32_t j0 = dStart[d], j1 = dStart[d + 1];
    const uint32_t g = gIdx1[j0] - 1, r = d - gStartD[g], nd = gStartD[g + 1] - gStartD[g];
    const uint64_t key = sKey[j0], a = sA[j0];
    const uint32_t o = (uint32_t)(a & 31u), mini = (uint32_t)(a >> 5) & ss::M30, h = ss::mix30(mini);
    long long owner = -1;
    for (uint32_t j = j0; j < j1; j++) {
        const uint32_t row = sRow[j];
        if (upper_keys == 1 || !(flags[row] & SS_ROW_LOWER)) owner = row;           // rows ascend within equal k-mers
    }
    if (owner >= 0) row_valid[owner] = 1;
    else atomicAdd(orphans, 1ull);
    if (nd > inline_max) {
        const uint32_t slot = gSlotBase[g] + 1u + r;
        mkeys[slot] = key;
        for (uint32_t j = j0; j < j1; j++) slot_of_row[sRow[j]] = slot;
    } else {
        const uint32_t i = gItemBase[g] + r;
        it.h[i] = h;
        it.lo[i] = ss::flank_of_key(key, o);
        it.mid[i] = (uint16_t)(((h >> 8) & 0xFFFu) << 4);
        it.hi8[i] = (uint8_t)(16u - o);
        it.e0[i] = j0;
        it.e1[i] = j1;
    }
}

// one thread per minimizer with a bucket: offset mask, "multi", header, the page item that refers to the bucket
__global__ void bucket_kernel(const uint64_t *__restrict__ sA, const uint32_t *__restrict__ dStart, const uint32_t *__restrict__ gStartD,
                              const uint32_t *__restrict__ gSlotBase, const uint32_t *__restrict__ gItemBase, uint32_t n_groups, uint32_t inline_max,
                              uint64_t *__restrict__ mkeys, Items it)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const uint32_t d0 = gStartD[g], d1 = gStartD[g + 1], nd = d1 - d0;
    if (nd <= inline_max) return;
    uint32_t mask = 0, multi = 0;
    for (uint32_t d = d0; d < d1; d++) {
        const uint32_t o = (uint32_t)(sA[dStart[d]] & 31u);
        if ((mask >> o) & 1u) multi = 1u;
        mask |= 1u << o;
    }
    const uint32_t mini = (uint32_t)(sA[dStart[d0]] >> 5) & ss::M30, h = ss::mix30(mini), hslot = gSlotBase[g], i = gItemBase[g];
    mkeys[hslot] = ((uint64_t)nd << 32) | (multi ? ss::HDR_MULTI : 0u) | mask;
    it.h[i] = h;
    it.lo[i] = (multi << 31) | hslot;
    it.mid[i] = (uint16_t)(mask & 0xFFFFu);
    it.hi8[i] = (uint8_t)(0x80u | ((mask >> 16) << 6) | ((h >> 8) & 0x3Fu));
    it.e0[i] = 0;
    it.e1[i] = 0;
}

__global__ void iota_kernel(uint32_t *v, uint32_t n)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) v[j] = j;
}

// w_i = 8 home_i - i + part_i * PART_BIG (items in page order)
__global__ void place_key_kernel(const uint32_t *__restrict__ h_sorted, uint32_t n_items, uint32_t n_pages, long long *__restrict__ w)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const uint32_t h = h_sorted[i];
    w[i] = 8ll * (long long)ss::page_of(h, n_pages) - (long long)i + (long long)(h >> (30 - PB)) * PART_BIG;
}

__device__ __forceinline__ uint64_t lo_of(uint32_t pt, uint32_t n_pages) { return pt >= (uint32_t)NP ? n_pages : ss::page_of(pt << (30 - PB), n_pages); }

// pos_i from the prefix maximum; items beyond their partition's pages are listed as spills, the others fill their page
__global__ void place_pos_kernel(const uint32_t *__restrict__ h_sorted, const long long *__restrict__ wmax, uint32_t n_items, uint32_t n_pages,
                                 uint64_t *__restrict__ pos, uint32_t *__restrict__ fill, uint32_t *__restrict__ spill, uint32_t spill_cap,
                                 uint32_t *__restrict__ n_spill)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const uint32_t pt = h_sorted[i] >> (30 - PB);
    const uint64_t p = (uint64_t)(wmax[i] - (long long)pt * PART_BIG + (long long)i);
    if (p >= 8ull * lo_of(pt + 1, n_pages)) {
        const uint32_t s = atomicAdd(n_spill, 1u);
        if (s < spill_cap) spill[s] = i;
        pos[i] = ~0ull;
    } else {
        pos[i] = p;
        atomicAdd(&fill[p >> 3], 1u);
    }
}

__global__ void pages_init_kernel(uint4 *pages, uint64_t n_alloc)
{
    const uint64_t pg = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pg >= n_alloc) return;
    const uint32_t t = 0xFFFFFFFFu, e = 0x7F7F7F7Fu;                // PG_EMPTY_TAG x 8, PG_EMPTY_HI x 8
    pages[pg * 4] = make_uint4(t, t, e, e);
    pages[pg * 4 + 1] = make_uint4(0, 0, 0, 0);
    pages[pg * 4 + 2] = make_uint4(0, 0, 0, 0);
    pages[pg * 4 + 3] = make_uint4(0, 0, 0, 0);
}

// one thread per item (page order): its slot's four fields, and the counter of the rows of an inline item
__global__ void pages_write_kernel(const uint32_t *__restrict__ order, Items it, const uint64_t *__restrict__ pos, uint32_t n_items, uint64_t n_mslots,
                                   const uint32_t *__restrict__ sRow, uint8_t *__restrict__ pages, uint32_t *__restrict__ slot_of_row)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const uint64_t p = pos[i];
    const uint32_t src = order[i];
    const uint64_t pg = p >> 3;
    const uint32_t sl = (uint32_t)(p & 7u);
    uint8_t *pp = pages + pg * 64;
    pp[sl] = (uint8_t)(it.h[src] & 0xFFu);
    pp[8 + sl] = it.hi8[src];
    *reinterpret_cast<uint32_t *>(pp + 16 + 4 * sl) = it.lo[src];
    *reinterpret_cast<uint16_t *>(pp + 48 + 2 * sl) = it.mid[src];
    const uint32_t slot = (uint32_t)(n_mslots + pg * 8 + sl);
    for (uint32_t q = it.e0[src]; q < it.e1[src]; q++) slot_of_row[sRow[q]] = slot;
}

__global__ void bloom_kernel(const uint32_t *__restrict__ h_sorted, uint32_t n_items, int bits, uint32_t *__restrict__ bloom)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const uint32_t hb = h_sorted[i] >> (30 - bits);
    atomicOr(&bloom[hb >> 5], 1u << (hb & 31u));
}

struct MaxOp { __host__ __device__ long long operator()(long long a, long long b) const { return a > b ? a : b; } };

struct Pool {                        // device scratch of one build, freed together
    std::vector<void *> p;
    ~Pool() { for (void *q : p) hipFree(q); }
    template <typename T> T *get(uint64_t n)
    {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<uint64_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        p.push_back(q);
        return static_cast<T *>(q);
    }
};

unsigned blocks_for(uint64_t n) { return (unsigned)std::max<uint64_t>(1, (n + 255) / 256); }

}  // namespace

namespace ss {

int build_mini_dev(ss_db *db, const uint64_t *keys, const uint8_t *flags, uint64_t n_rows, int upper_keys)
{
    if (n_rows == 0 || n_rows >= 0x7FFFFFF0ull) return SS_ERANGE;
    const int k = db->k;
    static const bool trace = getenv("SS_BUILD_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        hipDeviceSynchronize();
        fprintf(stderr, "[build-dev] %-26s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    uint32_t inline_max = 2;                        // (decided once the minimizers' sizes are known: choose_inline_max)
    double lambda = 2.0;
    if (const char *e = getenv("SS_PAGE_LAMBDA")) lambda = std::max(0.25, std::min(7.8, atof(e)));
    Pool P;
#define SS_D(call) do { if ((call) != hipSuccess) { ss::set_last_error(#call, __FILE__, __LINE__, hipGetLastError()); return SS_EHIP; } } while (0)
#define SS_P(ptr) do { if (!(ptr)) return SS_ENOMEM; } while (0)
    const uint32_t n = (uint32_t)n_rows;
    uint64_t *d_keys = P.get<uint64_t>(n);
    uint8_t *d_flags = P.get<uint8_t>(n);
    uint32_t *d_v = P.get<uint32_t>(n), *d_pos = P.get<uint32_t>(n + 1);
    SS_P(d_keys); SS_P(d_flags); SS_P(d_v); SS_P(d_pos);
    SS_D(hipMemcpy(d_keys, keys, (uint64_t)n * 8, hipMemcpyHostToDevice));
    SS_D(hipMemcpy(d_flags, flags, n, hipMemcpyHostToDevice));
    // hipcub scratch: sized for the largest call (sorts of n 64-bit keys)
    size_t tb = 0, t2 = 0;
    {
        uint64_t *k64 = nullptr; uint32_t *v32 = nullptr; long long *ll = nullptr;
        hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k64, k64, v32, v32, (int)n, 0, 64);
        hipcub::DeviceScan::InclusiveSum(nullptr, t2, v32, v32, (int)n); tb = std::max(tb, t2);
        hipcub::DeviceScan::ExclusiveSum(nullptr, t2, v32, v32, (int)n + 1); tb = std::max(tb, t2);
        hipcub::DeviceScan::InclusiveScan(nullptr, t2, ll, ll, MaxOp(), (int)n); tb = std::max(tb, t2);
        hipcub::DeviceRadixSort::SortPairs(nullptr, t2, v32, v32, v32, v32, (int)n, 0, 30); tb = std::max(tb, t2);
    }
    char *d_tmp = P.get<char>(tb + 256);
    SS_P(d_tmp);
    // 1. valid rows -> entries in row order
    hipLaunchKernelGGL(valid_kernel, dim3(blocks_for(n)), dim3(256), 0, 0, d_flags, (uint64_t)n, d_v);
    t2 = tb;
    SS_D(hipcub::DeviceScan::ExclusiveSum(d_tmp, t2, d_v, d_pos, (int)n));
    uint32_t last_pos = 0, last_v = 0;
    SS_D(hipMemcpy(&last_pos, d_pos + (n - 1), 4, hipMemcpyDeviceToHost));
    SS_D(hipMemcpy(&last_v, d_v + (n - 1), 4, hipMemcpyDeviceToHost));
    const uint32_t nv = last_pos + last_v;
    if (nv == 0) return SS_ERANGE;                                  // (an empty index: the host build knows what to do)
    uint64_t *eA = P.get<uint64_t>(nv), *eKey = P.get<uint64_t>(nv), *sA = P.get<uint64_t>(nv), *sKey = P.get<uint64_t>(nv), *k64tmp = P.get<uint64_t>(nv);
    uint32_t *eRow = P.get<uint32_t>(nv), *idx = P.get<uint32_t>(nv), *perm1 = P.get<uint32_t>(nv), *perm2 = P.get<uint32_t>(nv), *sRow = P.get<uint32_t>(nv);
    SS_P(eA); SS_P(eKey); SS_P(sA); SS_P(sKey); SS_P(k64tmp); SS_P(eRow); SS_P(idx); SS_P(perm1); SS_P(perm2); SS_P(sRow);
    hipLaunchKernelGGL(ents_kernel, dim3(blocks_for(n)), dim3(256), 0, 0, d_keys, d_flags, d_pos, (uint64_t)n, k, eA, eKey, eRow, idx);
    lap("1 minimizers");
    // 2. order (A, k-mer, row): stable sort by k-mer (the entries are in row order), then stable sort by A
    t2 = tb;
    SS_D(hipcub::DeviceRadixSort::SortPairs(d_tmp, t2, eKey, k64tmp, idx, perm1, (int)nv, 0, 62));
    hipLaunchKernelGGL(gather64_kernel, dim3(blocks_for(nv)), dim3(256), 0, 0, eA, perm1, nv, sKey);      // (sKey used as scratch: A in k-mer order)
    t2 = tb;
    SS_D(hipcub::DeviceRadixSort::SortPairs(d_tmp, t2, sKey, sA, perm1, perm2, (int)nv, 0, 43));
    hipLaunchKernelGGL(gather64_kernel, dim3(blocks_for(nv)), dim3(256), 0, 0, eKey, perm2, nv, sKey);
    hipLaunchKernelGGL(gather32_kernel, dim3(blocks_for(nv)), dim3(256), 0, 0, eRow, perm2, nv, sRow);
    lap("2 sort");
    // 3. distinct k-mers, minimizer groups, places
    uint32_t *newd = idx, *newg = perm1, *dIdx1 = perm2, *gIdx1 = eRow;      // (the sort's index arrays are free again)
    hipLaunchKernelGGL(flags_kernel, dim3(blocks_for(nv)), dim3(256), 0, 0, sA, sKey, nv, newd, newg);
    t2 = tb;
    SS_D(hipcub::DeviceScan::InclusiveSum(d_tmp, t2, newd, dIdx1, (int)nv));
    t2 = tb;
    SS_D(hipcub::DeviceScan::InclusiveSum(d_tmp, t2, newg, gIdx1, (int)nv));
    uint32_t n_dist = 0, n_groups = 0;
    SS_D(hipMemcpy(&n_dist, dIdx1 + (nv - 1), 4, hipMemcpyDeviceToHost));
    SS_D(hipMemcpy(&n_groups, gIdx1 + (nv - 1), 4, hipMemcpyDeviceToHost));
    uint32_t *dStart = P.get<uint32_t>((uint64_t)n_dist + 1), *gStartD = P.get<uint32_t>((uint64_t)n_groups + 1);
    uint32_t *gslots = P.get<uint32_t>((uint64_t)n_groups + 1), *gitems = P.get<uint32_t>((uint64_t)n_groups + 1);
    uint32_t *gSlotBase = P.get<uint32_t>((uint64_t)n_groups + 1), *gItemBase = P.get<uint32_t>((uint64_t)n_groups + 1);
    SS_P(dStart); SS_P(gStartD); SS_P(gslots); SS_P(gitems); SS_P(gSlotBase); SS_P(gItemBase);
    hipLaunchKernelGGL(starts_kernel, dim3(blocks_for(nv)), dim3(256), 0, 0, newd, newg, dIdx1, gIdx1, nv, dStart, gStartD);
    SS_D(hipMemcpy(dStart + n_dist, &nv, 4, hipMemcpyHostToDevice));
    SS_D(hipMemcpy(gStartD + n_groups, &n_dist, 4, hipMemcpyHostToDevice));
    {
        unsigned long long *d_small = reinterpret_cast<unsigned long long *>(gslots), small = 0;      // (scratch: gslots is written next)
        static_assert(sizeof(unsigned long long) == 8, "two words of gslots");
        SS_D(hipMemset(d_small, 0, 8));
        hipLaunchKernelGGL(small_groups_kernel, dim3(std::min<unsigned>(blocks_for(n_groups), 512u)), dim3(256), 0, 0, gStartD, n_groups, d_small);
        SS_D(hipMemcpy(&small, d_small, 8, hipMemcpyDeviceToHost));
        inline_max = choose_inline_max(small, n_dist);
    }
    hipLaunchKernelGGL(group_sizes_kernel, dim3(blocks_for(n_groups)), dim3(256), 0, 0, gStartD, n_groups, inline_max, gslots, gitems);
    SS_D(hipMemset(gslots + n_groups, 0, 4));
    SS_D(hipMemset(gitems + n_groups, 0, 4));
    t2 = tb;
    SS_D(hipcub::DeviceScan::ExclusiveSum(d_tmp, t2, gslots, gSlotBase, (int)n_groups + 1));
    t2 = tb;
    SS_D(hipcub::DeviceScan::ExclusiveSum(d_tmp, t2, gitems, gItemBase, (int)n_groups + 1));
    uint32_t raw_slots = 0, n_items = 0;
    SS_D(hipMemcpy(&raw_slots, gSlotBase + n_groups, 4, hipMemcpyDeviceToHost));
    SS_D(hipMemcpy(&n_items, gItemBase + n_groups, 4, hipMemcpyDeviceToHost));
    const uint64_t n_mslots = std::max<uint64_t>(1, raw_slots);
    if (n_mslots >= (uint64_t)START_MASK) return SS_ERANGE;
    lap("3 groups");
    // 4. buckets, owners, items.  The index arrays stay on the device: they ARE the database image.
    const uint64_t nr = std::max<uint64_t>(1, n_rows);
    uint64_t *d_mkeys = nullptr;
    uint32_t *d_slot_of_row = nullptr;
    uint8_t *d_row_valid = nullptr;
    auto fail = [&](int rc) { hipFree(d_mkeys); hipFree(d_slot_of_row); hipFree(d_row_valid); return rc; };
#define SS_F(call) do { if ((call) != hipSuccess) { ss::set_last_error(#call, __FILE__, __LINE__, hipGetLastError()); return fail(SS_EHIP); } } while (0)
    SS_F(hipMalloc((void **)&d_mkeys, n_mslots * 8));
    SS_F(hipMalloc((void **)&d_slot_of_row, nr * 4));
    SS_F(hipMalloc((void **)&d_row_valid, nr));
    SS_F(hipMemset(d_mkeys, 0, n_mslots * 8));
    SS_F(hipMemset(d_slot_of_row, 0xFF, nr * 4));
    SS_F(hipMemset(d_row_valid, 0, nr));
    Items it;
    it.h = P.get<uint32_t>(n_items); it.lo = P.get<uint32_t>(n_items); it.e0 = P.get<uint32_t>(n_items); it.e1 = P.get<uint32_t>(n_items);
    it.mid = P.get<uint16_t>(n_items); it.hi8 = P.get<uint8_t>(n_items);
    unsigned long long *d_orph = P.get<unsigned long long>(1);
    if (!it.h || !it.lo || !it.e0 || !it.e1 || !it.mid || !it.hi8 || !d_orph) return fail(SS_ENOMEM);
    SS_F(hipMemset(d_orph, 0, 8));
    hipLaunchKernelGGL(distinct_kernel, dim3(blocks_for(n_dist)), dim3(256), 0, 0, sA, sKey, sRow, d_flags, upper_keys, dStart, gIdx1, gStartD, gSlotBase, gItemBase,
                       n_dist, inline_max, d_mkeys, d_slot_of_row, d_row_valid, it, d_orph);
    hipLaunchKernelGGL(bucket_kernel, dim3(blocks_for(n_groups)), dim3(256), 0, 0, sA, dStart, gStartD, gSlotBase, gItemBase, n_groups, inline_max, d_mkeys, it);
    unsigned long long orphans = 0;
    SS_F(hipMemcpy(&orphans, d_orph, 8, hipMemcpyDeviceToHost));
    if (orphans && upper_keys == 0) return fail(SS_EKEY);
    lap("4 buckets + items");
    // 5. items in page order (stable by h)
    uint32_t *ord_in = P.get<uint32_t>(n_items), *order = P.get<uint32_t>(n_items), *h_sorted = P.get<uint32_t>(n_items);
    if (!ord_in || !order || !h_sorted) return fail(SS_ENOMEM);
    hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n_items)), dim3(256), 0, 0, ord_in, n_items);
    t2 = tb;
    SS_F(hipcub::DeviceRadixSort::SortPairs(d_tmp, t2, it.h, h_sorted, ord_in, order, (int)n_items, 0, 30));
    lap("5 page order");
    // 6. placement (the table grows until the checks hold, as in the host build)
    uint64_t n_pages = std::max<uint64_t>(PG_MIN_PAGES, (uint64_t)((double)n_items / lambda) + 1), n_alloc = 0;
    long long *d_w = P.get<long long>(n_items), *d_wmax = P.get<long long>(n_items);
    uint64_t *d_ipos = P.get<uint64_t>(n_items);
    constexpr uint32_t SPILL_CAP = 1u << 16;
    uint32_t *d_spill = P.get<uint32_t>(SPILL_CAP), *d_nspill = P.get<uint32_t>(1);
    if (!d_w || !d_wmax || !d_ipos || !d_spill || !d_nspill) return fail(SS_ENOMEM);
    uint32_t *d_fill = nullptr;
    std::vector<uint32_t> fill;
    for (;; n_pages += n_pages / 4) {
        if (n_mslots + (n_pages + n_pages / 1024) * PG_SLOTS >= 0xFFFFFFF0ull) { hipFree(d_fill); return fail(SS_ERANGE); }
        const uint64_t D = n_pages / 1024;
        n_alloc = n_pages + D;
        hipFree(d_fill);
        d_fill = nullptr;
        if (hipMalloc((void **)&d_fill, n_alloc * 4) != hipSuccess) return fail(SS_ENOMEM);
        bool good = hipMemset(d_fill, 0, n_alloc * 4) == hipSuccess && hipMemset(d_nspill, 0, 4) == hipSuccess;
        hipLaunchKernelGGL(place_key_kernel, dim3(blocks_for(n_items)), dim3(256), 0, 0, h_sorted, n_items, (uint32_t)n_pages, d_w);
        t2 = tb;
        good = good && hipcub::DeviceScan::InclusiveScan(d_tmp, t2, d_w, d_wmax, MaxOp(), (int)n_items) == hipSuccess;
        hipLaunchKernelGGL(place_pos_kernel, dim3(blocks_for(n_items)), dim3(256), 0, 0, h_sorted, d_wmax, n_items, (uint32_t)n_pages, d_ipos, d_fill, d_spill,
                           SPILL_CAP, d_nspill);
        uint32_t n_spill = 0;
        good = good && hipMemcpy(&n_spill, d_nspill, 4, hipMemcpyDeviceToHost) == hipSuccess;
        fill.resize(n_alloc);
        good = good && hipMemcpy(fill.data(), d_fill, n_alloc * 4, hipMemcpyDeviceToHost) == hipSuccess;
        if (!good) { hipFree(d_fill); ss::set_last_error("build_mini_dev placement", __FILE__, __LINE__, hipGetLastError()); return fail(SS_EHIP); }
        bool ok = n_spill <= SPILL_CAP;
        if (ok && n_spill) {
            // the items that left their partition's pages: placed serially, in item order, at or behind the next partition's
            // first page (or their home page if that lies further on) -- ss_mini.hip's spill pass
            std::vector<uint32_t> sp(n_spill), sh(n_spill);
            std::vector<uint64_t> spos(n_spill);
            ok = hipMemcpy(sp.data(), d_spill, (uint64_t)n_spill * 4, hipMemcpyDeviceToHost) == hipSuccess;
            std::sort(sp.begin(), sp.end());
            for (uint32_t s = 0; s < n_spill && ok; s++) ok = hipMemcpy(&sh[s], h_sorted + sp[s], 4, hipMemcpyDeviceToHost) == hipSuccess;
            auto lo_h = [&](uint32_t pt) -> uint64_t { return pt >= (uint32_t)NP ? n_pages : page_of(pt << (30 - PB), (uint32_t)n_pages); };
            for (uint32_t s = 0; s < n_spill && ok; s++) {
                const uint32_t pt = sh[s] >> (30 - PB);
                uint64_t pg = std::max<uint64_t>(page_of(sh[s], (uint32_t)n_pages), lo_h(pt + 1));
                while (pg < n_alloc && fill[pg] == PG_SLOTS) pg++;
                if (pg >= n_alloc) { ok = false; break; }
                spos[s] = pg * 8 + fill[pg]++;
            }
            for (uint32_t s = 0; s < n_spill && ok; s++) ok = hipMemcpy(d_ipos + sp[s], &spos[s], 8, hipMemcpyHostToDevice) == hipSuccess;
        }
        uint64_t run = 0, longest = 0;
        for (uint64_t pg = 0; pg < n_alloc && ok; pg++) {
            run = fill[pg] == PG_SLOTS ? run + 1 : 0;
            longest = std::max(longest, run);
        }
        if (trace) fprintf(stderr, "[build-dev] %llu pages: %u items left their partition's pages, longest run of full pages %llu (limit %llu)%s\n",
                           (unsigned long long)n_pages, n_spill, (unsigned long long)longest, (unsigned long long)D,
                           ok && longest < D && fill[n_alloc - 1] < PG_SLOTS ? "" : " -> the table grows");
        if (ok && longest < D && fill[n_alloc - 1] < PG_SLOTS) break;
    }
    hipFree(d_fill);
    lap("6 placement");
    // 7. pages, Bloom filter, counters
    uint8_t *d_pages = nullptr;
    uint32_t *d_counts = nullptr, *d_bloom = nullptr;
    const uint64_t n_slots = n_mslots + n_alloc * PG_SLOTS;
    auto fail2 = [&](int rc) { hipFree(d_pages); hipFree(d_counts); hipFree(d_bloom); return fail(rc); };
#define SS_G(call) do { if ((call) != hipSuccess) { ss::set_last_error(#call, __FILE__, __LINE__, hipGetLastError()); return fail2(SS_EHIP); } } while (0)
    SS_G(hipMalloc((void **)&d_pages, n_alloc * 64));
    SS_G(hipMalloc((void **)&d_counts, n_slots * 4));
    hipLaunchKernelGGL(pages_init_kernel, dim3(blocks_for(n_alloc)), dim3(256), 0, 0, reinterpret_cast<uint4 *>(d_pages), n_alloc);
    hipLaunchKernelGGL(pages_write_kernel, dim3(blocks_for(n_items)), dim3(256), 0, 0, order, it, d_ipos, n_items, n_mslots, sRow, d_pages, d_slot_of_row);
    SS_G(hipMemset(d_counts, 0, n_slots * 4));
    int bits = 10;
    while (bits < 25 && (1ull << bits) < 8ull * n_groups) bits++;
    if ((1ull << bits) < 4ull * n_groups) bits = 0;
    if (const char *bb = getenv("SS_BLOOM_BITS")) bits = atoi(bb);
    uint64_t bloom_bytes = 0;
    if (bits >= 10 && bits <= 30) {
        bloom_bytes = (1ull << bits) / 8;
        SS_G(hipMalloc((void **)&d_bloom, bloom_bytes));
        SS_G(hipMemset(d_bloom, 0, bloom_bytes));
        hipLaunchKernelGGL(bloom_kernel, dim3(blocks_for(n_items)), dim3(256), 0, 0, h_sorted, n_items, bits, d_bloom);
    }
    SS_G(hipGetLastError());
    SS_G(hipDeviceSynchronize());
    lap("7 pages + bloom");
#undef SS_D
#undef SS_P
#undef SS_F
#undef SS_G
    db->n_distinct = n_dist;
    db->n_mslots = n_mslots;
    db->n_inline = ((uint64_t)n_dist + n_items - raw_slots) / 2;
    db->n_slots = n_slots;
    db->n_dir = (uint32_t)n_pages;
    db->n_dir_alloc = (uint32_t)n_alloc;
    db->dirbits = 0;
    db->n_buckets = n_groups;
    db->capacity = n_slots;
    db->d_mkeys = d_mkeys;
    db->d_dir = reinterpret_cast<uint64_t *>(d_pages);
    db->d_counts = d_counts;
    db->d_slot_of_row = d_slot_of_row;
    db->d_row_valid = d_row_valid;
    db->d_bloom = d_bloom;
    db->bloom_bits = d_bloom ? (uint32_t)bits : 0;
    db->device_bytes = n_mslots * 8 + n_slots * 4 + n_alloc * 64 + nr * 5 + bloom_bytes;
    return mark_solid(db);
}

}  // namespace ss
