// ss_reorder.hip -- a resident read set in LOCALITY order.
//
// What bounds the scan of a table of sampled node sets is the number of random 64-byte sectors its page lookups pull
// from memory (ss_mini.hip): ~13 lookups per read, 54 G/s, all the memory system gives.  But a sample
// covers its genomes many times: the reads that start within the same 17 bases of a genome have the same minimizer in
// their first k-mer and share nearly all of their other minimizers too.  In file order such reads are millions of
// records apart -- every lookup misses the 4 MB L2; when they are neighbours, the first read of a group pays the
// sector and the others find it in L2.  Counting does not care about the order of the records (integer sums), and the
// reads are parsed and shipped once per sample but scanned several times (tree scan, one scan per identified
// cluster, two more with -b: identify.py:409, Vote_Strain_L2_Lasso_new_sp.py:354-372, identify_low_depth.py:119,124;
// or one sample against the databases of many species).
// So a resident read set can be rewritten with its records sorted by the minimizer (the 30-bit m-mer, same ordering key
// as the index) of their first 31 bases: find the record starts and ends (two streaming passes, 16 bytes per lane),
// key every record, radix-sort (key, record) pairs, prefix-sum the lengths, copy.  9.7 ms per 20 M reads on MI355X
// (copy 4.6, the rest 0.6-1.1 each); the scan of 20 M reads of a 70/20/10 three-strain sample then takes 3.8 instead of
// 5.7 ms (sampled table) / 2.8 instead of 3.4 ms (contiguous).  The gain grows with the coverage (these reads cover
// their genomes 400/115/60 fold; at one-fold coverage it is nil), and it is won per scan while the ordering is paid
// once: it pays from about five scans of the same resident sample on -- one sample against many databases -- and does
// not for the CLI's one tree scan plus a few small cluster scans.  Hence OPT-IN: SS_READS_ORDER=locality for
// ss_reads_load, the `order` argument of ss_reads_from_flat_dev; bench.py reports both orders.
#include "ss_common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <vector>

namespace {

constexpr int RB = 4096;                    // bytes of a slab handled by one workgroup of the boundary passes
constexpr uint32_t KEY_NONE = 0x40000000u;  // records shorter than a k-mer or with a non-ACGT base in their first 31

// newline mask of 16 bytes (bit i = byte i is '\n'), SWAR zero-byte test on w ^ 0x0A0A0A0A
__device__ __forceinline__ uint32_t nl_mask16(const uint4 v)
{
    uint32_t m = 0;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t x = w[d] ^ 0x0A0A0A0Au;
        const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;      // 0x80 where the byte was '\n'
        m |= (((z >> 7) | (z >> 14) | (z >> 21) | (z >> 28)) & 0xFu) << (4 * d);
    }
    return m;
}

// pass 1: record starts per workgroup; pass 2 (WRITE): their positions and the positions of the record ends.
// A workgroup = 256 threads x 16 bytes; a record starts at a base that follows a '\n' (or the slab's first byte) and
// ends at the first '\n' behind it.
template <bool WRITE>
__global__ __launch_bounds__(256) void bounds_kernel(const char *__restrict__ b, uint64_t n, uint32_t *__restrict__ counts,
                                                     const uint64_t *__restrict__ base, uint64_t *__restrict__ starts,
                                                     uint64_t *__restrict__ ends)
{
    __shared__ uint32_t s_wave[4][2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * RB + (uint64_t)t * 16;
    uint32_t nl = 0xFFFFu, prev = 1u;                           // beyond the slab: newlines
    if (i0 < n) {                                               // n is a multiple of 16 (padded blocks)
        nl = nl_mask16(*reinterpret_cast<const uint4 *>(b + i0));
        prev = i0 == 0 ? 1u : (uint32_t)(b[i0 - 1] == '\n');
    }
    const uint32_t before = ((nl << 1) | prev) & 0xFFFFu;       // bit i = byte i - 1 is a newline
    const uint32_t st = ~nl & before & 0xFFFFu, en = nl & ~before & 0xFFFFu;
    uint32_t cs = (uint32_t)__popc(st), ce = (uint32_t)__popc(en);
    // inclusive wave scan of (cs, ce) packed in one word (<= 16 per thread, 1024 per wave)
    uint32_t pk = cs | (ce << 16);
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)pk, off, 64);
        if (lane >= off) pk += o;
    }
    if (lane == 63) { s_wave[wave][0] = pk & 0xFFFFu; s_wave[wave][1] = pk >> 16; }
    __syncthreads();
    if (!WRITE) {
        if (t == 0) counts[blockIdx.x] = s_wave[0][0] + s_wave[1][0] + s_wave[2][0] + s_wave[3][0];
        return;
    }
    uint64_t s_off = base[blockIdx.x], e_off = s_off;
    // a record's end may lie in a later workgroup than its start: the ends before this workgroup = the starts before it
    // minus the record that is still open at its first byte
    const uint64_t lo = (uint64_t)blockIdx.x * RB;
    if (lo > 0 && lo < n && b[lo - 1] != '\n') e_off -= 1;
    for (int w = 0; w < wave; w++) { s_off += s_wave[w][0]; e_off += s_wave[w][1]; }
    s_off += (pk & 0xFFFFu) - cs;
    e_off += (pk >> 16) - ce;
    for (uint32_t m = st; m; m &= m - 1) starts[s_off++] = i0 + (uint32_t)__ffs(m) - 1u;
    for (uint32_t m = en; m; m &= m - 1) ends[e_off++] = i0 + (uint32_t)__ffs(m) - 1u;
}

// key of a record = the minimizer (30-bit m-mer, ordering key of ss_mini.hip, leftmost on ties) of its first 31 bases
__global__ void keys_kernel(const char *__restrict__ b, const uint64_t *__restrict__ starts, const uint64_t *__restrict__ ends,
                            uint32_t n_rec, uint32_t *__restrict__ keys, uint32_t *__restrict__ idx, uint64_t *__restrict__ len1)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    const uint64_t s = starts[r], len = ends[r] - s;
    idx[r] = r;
    len1[r] = len + 1;
    uint32_t key = KEY_NONE;
    if (len >= 31) {
        uint64_t km = 0;
        bool ok = true;
        unsigned char by[32];
        __builtin_memcpy(by, b + s, 32);                    // the slab is padded: 32 bytes from a record start are inside it
        for (int j = 0; j < 31; j++) {
            const int c = ss::base_code(by[j]);
            ok = ok && c >= 0;
            km |= (uint64_t)(c & 3) << (2 * j);
        }
        if (ok) {
            uint32_t best = 0xFFFFFFFFu, bx = 0;
            for (int i = 0; i < 17; i++) {
                const uint32_t x = (uint32_t)(km >> (2 * i)) & 0x3FFFFFFFu;
                const uint32_t h = ((x & 0xFFFFFFu) * (0x4F1BBu << 5) + 0x7F4A7C00u) & ~31u;
                if (h < best) { best = h; bx = x; }
            }
            key = bx;
        }
    }
    keys[r] = key;
}

__global__ void gather_len_kernel(const uint64_t *__restrict__ len1, const uint32_t *__restrict__ order, uint32_t n_rec,
                                  uint64_t *__restrict__ out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_rec) out[j] = len1[order[j]];
}

// record order[j] -> dst + off[j], followed by its '\n'; 16 lanes per record, 16 (unaligned) bytes per lane and round
__global__ __launch_bounds__(256) void copy_records_kernel(const char *__restrict__ src, const uint64_t *__restrict__ starts,
                                                           const uint64_t *__restrict__ len1, const uint32_t *__restrict__ order,
                                                           const uint64_t *__restrict__ off, uint32_t n_rec, char *__restrict__ dst)
{
    const uint32_t j = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (j >= n_rec) return;
    const uint32_t r = order[j];
    const uint64_t s = starts[r], l1 = len1[r], o = off[j], len = l1 - 1;
    for (uint64_t c = (uint64_t)(threadIdx.x & 15) * 16; c < l1; c += 256) {
        if (c + 16 <= len) {
            uint4 v;
            __builtin_memcpy(&v, src + s + c, 16);
            __builtin_memcpy(dst + o + c, &v, 16);
        } else {
            for (uint64_t i = c; i < l1 && i < c + 16; i++) dst[o + i] = i == len ? '\n' : src[s + i];
        }
    }
}

int order_slab(ss_reads::Slab &sl)
{
    const uint64_t n = sl.used;
    if (n < 64) return SS_OK;
    const unsigned nb = (unsigned)((n + RB - 1) / RB);
    static const bool trace = getenv("SS_INGEST_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        hipDeviceSynchronize();
        fprintf(stderr, "[reorder] %-22s at %.4f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    uint32_t *d_cnt = nullptr, *d_keys = nullptr, *d_keys2 = nullptr, *d_idx = nullptr, *d_ord = nullptr;
    uint64_t *d_base = nullptr, *d_starts = nullptr, *d_ends = nullptr, *d_len1 = nullptr, *d_lsort = nullptr, *d_off = nullptr;
    void *d_tmp = nullptr;
    char *d_new = nullptr, *d_small = nullptr, *d_big = nullptr;       // two allocations hold all the scratch (a hipMalloc costs ~0.2 ms)
    auto cleanup = [&] { hipFree(d_small); hipFree(d_big); };
#define SS_R(call) do { if ((call) != hipSuccess) { ss::set_last_error(#call, __FILE__, __LINE__, hipGetLastError()); cleanup(); hipFree(d_new); return SS_EHIP; } } while (0)
    size_t tb0 = 0;
    SS_R(hipcub::DeviceScan::ExclusiveSum(nullptr, tb0, d_cnt, d_base, (int)nb));
    const uint64_t small_bytes = (((uint64_t)nb * 4 + 255) & ~255ull) + (((uint64_t)nb + 1) * 8 + 255 & ~255ull) + std::max<size_t>(tb0, 256);
    SS_R(hipMalloc((void **)&d_small, small_bytes));
    d_cnt = (uint32_t *)d_small;
    d_base = (uint64_t *)(d_small + (((uint64_t)nb * 4 + 255) & ~255ull));
    d_tmp = (char *)d_base + (((uint64_t)nb + 1) * 8 + 255 & ~255ull);
    hipLaunchKernelGGL((bounds_kernel<false>), dim3(nb), dim3(256), 0, 0, sl.d, n, d_cnt, (const uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr);
    SS_R(hipcub::DeviceScan::ExclusiveSum(d_tmp, tb0, d_cnt, d_base, (int)nb));
    uint64_t last_base = 0;
    uint32_t last_cnt = 0;
    SS_R(hipMemcpy(&last_base, d_base + (nb - 1), 8, hipMemcpyDeviceToHost));
    SS_R(hipMemcpy(&last_cnt, d_cnt + (nb - 1), 4, hipMemcpyDeviceToHost));
    lap("count starts");
    const uint64_t n_rec64 = last_base + last_cnt;
    if (n_rec64 < 2 || n_rec64 >= 0x7FFFFFF0ull) { cleanup(); return SS_OK; }      // nothing to order / too many for 32-bit record numbers
    const uint32_t n_rec = (uint32_t)n_rec64;
    size_t tb1 = 0, tb2 = 0;
    SS_R(hipcub::DeviceRadixSort::SortPairs(nullptr, tb1, d_keys, d_keys2, d_idx, d_ord, (int)n_rec, 0, 31));
    SS_R(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, d_lsort, d_off, (int)n_rec));
    const uint64_t a8 = ((uint64_t)n_rec * 8 + 255) & ~255ull, a4 = ((uint64_t)n_rec * 4 + 255) & ~255ull;
    const size_t tbig = std::max<size_t>(std::max(tb1, tb2), 256);
    SS_R(hipMalloc((void **)&d_big, 5 * a8 + 4 * a4 + tbig));
    d_starts = (uint64_t *)d_big; d_ends = (uint64_t *)(d_big + a8); d_len1 = (uint64_t *)(d_big + 2 * a8);
    d_lsort = (uint64_t *)(d_big + 3 * a8); d_off = (uint64_t *)(d_big + 4 * a8);
    d_keys = (uint32_t *)(d_big + 5 * a8); d_keys2 = (uint32_t *)(d_big + 5 * a8 + a4); d_idx = (uint32_t *)(d_big + 5 * a8 + 2 * a4);
    d_ord = (uint32_t *)(d_big + 5 * a8 + 3 * a4);
    void *d_tmp2 = d_big + 5 * a8 + 4 * a4;
    lap("allocations");
    hipLaunchKernelGGL((bounds_kernel<true>), dim3(nb), dim3(256), 0, 0, sl.d, n, d_cnt, d_base, d_starts, d_ends);
    lap("starts + ends");
    const unsigned rb = (n_rec + 255) / 256;
    hipLaunchKernelGGL(keys_kernel, dim3(rb), dim3(256), 0, 0, sl.d, d_starts, d_ends, n_rec, d_keys, d_idx, d_len1);
    lap("keys");
    SS_R(hipcub::DeviceRadixSort::SortPairs(d_tmp2, tb1, d_keys, d_keys2, d_idx, d_ord, (int)n_rec, 0, 31));
    lap("sort");
    hipLaunchKernelGGL(gather_len_kernel, dim3(rb), dim3(256), 0, 0, d_len1, d_ord, n_rec, d_lsort);
    SS_R(hipcub::DeviceScan::ExclusiveSum(d_tmp2, tb2, d_lsort, d_off, (int)n_rec));
    uint64_t off_last = 0, len_last = 0;
    SS_R(hipMemcpy(&off_last, d_off + (n_rec - 1), 8, hipMemcpyDeviceToHost));
    SS_R(hipMemcpy(&len_last, d_lsort + (n_rec - 1), 8, hipMemcpyDeviceToHost));
    const uint64_t total = off_last + len_last, cap = (total + 15) & ~15ull;
    if (total > n) { cleanup(); return SS_EINVAL; }                                 // cannot happen: every record + one '\n' was in the slab
    lap("offsets");
    SS_R(hipMalloc((void **)&d_new, std::max<uint64_t>(cap, 16)));
    lap("new slab");
    hipLaunchKernelGGL(copy_records_kernel, dim3((n_rec + 15) / 16), dim3(256), 0, 0, sl.d, d_starts, d_len1, d_ord, d_off, n_rec, d_new);
    SS_R(hipMemset(d_new + total, '\n', cap - total));
    SS_R(hipDeviceSynchronize());
    lap("copy");
#undef SS_R
    cleanup();
    hipFree(sl.d);
    lap("frees");
    sl.d = d_new;
    sl.cap = std::max<uint64_t>(cap, 16);
    sl.used = cap;
    return SS_OK;
}

}  // namespace

namespace ss {

int reads_order_for_locality(ss_reads *R, bool force)
{
    if (!R) return SS_EINVAL;
    if (!force) {
        const char *e = getenv("SS_READS_ORDER");
        if (!e || strcmp(e, "locality")) return SS_OK;          // default: file order (see the header comment)
    }
    uint64_t bytes = 0;
    for (auto &sl : R->slabs) {
        const int rc = order_slab(sl);
        if (rc) return rc;
        bytes += sl.cap;
    }
    R->device_bytes = bytes;
    return SS_OK;
}

}  // namespace ss
