// ss_reorder.hip -- a resident read set in LOCALITY order.
//
// What bounds the scan of a table of sampled node sets is the number of random 64-byte sectors its page lookups pull
// from memory (ss_mini.hip): ~13 lookups per read, 54 G/s, all the memory system gives.  But a sample
// covers its genomes many times: the reads that start within the same 17 bases of a genome have the same minimizer in
// their first k-mer and share nearly all of their other minimizers too.  In file order such reads are millions of
// records apart -- every lookup misses the 4 MB L2; when they are processed at about the same time, the first read of a
// group pays the sector and the others find it in L2.  Counting does not care about the order of the records (integer
// sums), and the reads are parsed and shipped once per sample but scanned several times (tree scan, one scan per
// identified cluster, two more with -b: identify.py:409, Vote_Strain_L2_Lasso_new_sp.py:354-372,
// identify_low_depth.py:119,124).
//
// "About the same time" is all that is needed: the chip has ~8 K scan waves = ~50 K reads in flight, so a total order
// buys nothing over BINS of a few thousand reads.  Round 2 sorted (hipcub radix sort of 20 M (key, record) pairs, five
// passes over per-record arrays, a gather copy at 1.3 TB/s: 9.7 ms per 20 M reads); now the records are binned in two
// streaming passes over the slab, with no per-record array at all:
//   bin  = top SS_ORDER_BITS (12) bits of h = mix30(minimizer of the record's first 31 bases) -- the hash that addresses the
//          index pages (ss_mini.hip), so a bin's FIRST lookups also walk the page table in ascending order;
//          records without a first k-mer (shorter, or a non-ACGT base in it) go to one extra bin at the end
//   pass 1  count_kernel: find the record starts of a 4 KB tile (16 bytes per lane, SWAR newline masks), the end of each
//           record (suffix minimum over the tile + a 512-byte halo), its bin; atomicAdd of the record's slot size to the
//           bin's byte count.  A slot = record + '\n', padded with '\n' to 8 bytes (152 bytes for a 150-base read: nothing
//           added) so that every piece of the copy is an aligned store
//   scan    exclusive prefix over the 4097 bin sizes (one workgroup)
//   pass 2  the same discovery again (cheaper than storing and re-reading 16 bytes per record), a returning atomicAdd
//           on the bin's cursor claims the slot, and the WAVE copies its records together: the records' 16-byte pieces are
//           numbered across the wave (prefix sum of the piece counts), every lane finds the record of its piece by
//           binary search in LDS, loads 16 unaligned bytes, pads behind the record's end with '\n' and stores them
//           aligned -- ~70 pieces for the ~7 records of a wave's 1 KB, two rounds of full-width loads and stores.
// Order inside a bin is whatever the atomics decide (not reproducible run to run; the multiset of records is, and so
// is every count).  ON by default for resident read sets (SS_READS_ORDER=file keeps the file order): it costs ~2-3 ms per
// 20 M reads against ~80 ms of parsing and PCIe for the same reads, and every scan of the set is then 0-35 % faster
// depending on the coverage of the sample (profiles/r03_locality_sweep.json).
#include "ss_common.h"
#include "ss_scan_dev.h"

#include <algorithm>
#include <chrono>
#include <vector>

namespace {

constexpr int RB = 4096;                    // bytes of a slab owned by one workgroup (256 lanes x 16 bytes)
constexpr int HALO = 512;                   // bytes behind the tile searched (in parallel) for the end of its last record
constexpr int MAX_BITS = 16;
constexpr uint32_t NO_NL = 0xFFFFu;         // "no newline" as a tile-relative position

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

// 16 bytes at b + i; bytes at or beyond n read as '\n'
__device__ __forceinline__ uint4 load16_nl(const char *__restrict__ b, uint64_t i, uint64_t n)
{
    if (i + 16 <= n) { uint4 v; __builtin_memcpy(&v, b + i, 16); return v; }
    uint32_t w[4] = {0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};
    for (int k = 0; k < 16 && i + k < n; k++) {
        const uint32_t ch = (uint8_t)b[i + k];
        w[k >> 2] = (w[k >> 2] & ~(0xFFu << (8 * (k & 3)))) | (ch << (8 * (k & 3)));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// bin of a record: top `bits` bits of mix30(minimizer of its first 31 bases) (ordering key of the index, leftmost on
// ties); 1 << bits when the record has no first k-mer.  `b + s .. + 32` is inside the buffer (callers check).
__device__ __forceinline__ uint32_t record_bin(const char *__restrict__ b, uint64_t s, uint64_t len, int bits)
{
    if (len < 31) return 1u << bits;
    uint4 q[2];
    __builtin_memcpy(q, b + s, 32);
    const uint32_t w[8] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w};
    uint64_t km = 0;
    uint32_t bad = 0;
#pragma unroll
    for (int d = 0; d < 8; d++) {
        const uint32_t c = (w[d] >> 1) & 0x03030303u;
        const uint32_t letter = __builtin_amdgcn_perm(0u, 0x47544341u, c);          // code -> 'A' 'C' 'T' 'G'
        uint32_t diff = (w[d] & 0xDFDFDFDFu) ^ letter;
        if (d == 7) diff &= 0x00FFFFFFu;                                            // byte 31 is not part of the k-mer
        bad |= diff;
        km |= (uint64_t)__builtin_amdgcn_udot4(c, 0x40100401u, 0u, false) << (8 * d);
    }
    if (bad) return 1u << bits;
    km &= 0x3FFFFFFFFFFFFFFFull;
    uint32_t best = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 31 - ss::MINI_M + 1; i++) {
        const uint32_t x = (uint32_t)(km >> (2 * i));
        best = min(best, (ss::mmkey(x) & ss::KEY_MASK) | (uint32_t)i);              // (key, position): leftmost on ties
    }
    const uint32_t x = (uint32_t)(km >> (2 * (best & 31u))) & ss::M30;
    return ss::mix30(x) >> (30 - bits);
}

__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// One kernel, two uses.  WRITE = false: bins[bin] += slot bytes of every record that STARTS in this tile.
// WRITE = true: bins[] holds the bins' cursors (exclusive prefix of the counts): claim and copy.
template <bool WRITE>
__global__ __launch_bounds__(256) void bin_kernel(const char *__restrict__ b, uint64_t n, int bits,
                                                  unsigned long long *__restrict__ bins, char *__restrict__ dst)
{
    __shared__ uint32_t s_first[5];                     // first newline (tile-relative) of waves 0..3 and of the halo
    __shared__ uint64_t s_src[4][64], s_dst[4][64];
    __shared__ uint32_t s_len[4][64], s_pend[4][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint64_t tile0 = (uint64_t)blockIdx.x * RB, i0 = tile0 + (uint64_t)t * 16;
    uint32_t nl = 0xFFFFu, prev = 1u;                   // beyond the buffer: newlines
    if (i0 < n) {
        nl = nl_mask16(load16_nl(b, i0, n));
        prev = i0 == 0 ? 1u : (uint32_t)(b[i0 - 1] == '\n');
    }
    // first newline at or behind every lane's chunk: suffix minimum over the wave, then over the later waves and the halo
    uint32_t mine = nl ? (uint32_t)t * 16u + (uint32_t)__builtin_ctz(nl) : NO_NL, suf = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_down((int)suf, off, 64);
        if (lane + off < 64) suf = min(suf, o);
    }
    if (lane == 0) s_first[wave] = suf;
    if (wave == 0) {                                    // halo: HALO bytes behind the tile, 32 lanes x 16 bytes
        uint32_t h = NO_NL;
        const uint64_t j0 = tile0 + RB + (uint64_t)lane * 16;
        if (lane < HALO / 16) {
            if (j0 < n) {
                const uint32_t m = nl_mask16(load16_nl(b, j0, n));         // (the buffer's end reads as a newline)
                if (m) h = RB + (uint32_t)lane * 16u + (uint32_t)__builtin_ctz(m);
            } else {
                h = RB + (uint32_t)lane * 16u;
            }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) h = min(h, (uint32_t)__shfl_xor((int)h, off, 64));
        if (lane == 0) s_first[4] = h;
    }
    __syncthreads();
    uint32_t later = (uint32_t)__shfl_down((int)suf, 1, 64);               // first newline behind this lane's chunk
    if (lane == 63) later = NO_NL;
    for (int w = wave + 1; w < 5; w++) later = min(later, s_first[w]);

    const uint32_t before = ((nl << 1) | prev) & 0xFFFFu;                  // bit i = byte i - 1 is a newline
    uint32_t st = ~nl & before & 0xFFFFu;                                  // record starts in this chunk
    while (__any(st != 0)) {
        // ---- every lane takes its next record start (usually there is one round: a 16-byte chunk starts <= 1 read)
        const bool has = st != 0;
        uint64_t s = 0, len = 0;
        uint32_t slot = 0, bin = 0;
        if (has) {
            const uint32_t bit = (uint32_t)__builtin_ctz(st);
            st &= st - 1;
            s = i0 + bit;
            const uint32_t up = nl & ~((2u << bit) - 1u);
            uint64_t e;
            if (up) e = i0 + (uint32_t)__builtin_ctz(up);
            else if (later != NO_NL) e = tile0 + later;
            else {                                                         // a record longer than the halo: walk on
                e = tile0 + RB + HALO;
                while (e < n) {
                    const uint32_t m = nl_mask16(load16_nl(b, e, n));
                    if (m) { e += (uint32_t)__builtin_ctz(m); break; }
                    e += 16;
                }
                e = min(e, n);
            }
            len = e - s;
            slot = (uint32_t)((len + 1 + 7) & ~7ull);                      // (a record of 4 GB does not exist: blocks are cut far below)
            bin = (s + 32 <= n) ? record_bin(b, s, len, bits) : (1u << bits);
        }
        if (!WRITE) {
            if (has) atomicAdd(&bins[bin], (unsigned long long)slot);
            continue;
        }
        // ---- claim the slots, then copy the wave's records together, 16 bytes per lane and round
        uint64_t d0 = 0;
        if (has) d0 = atomicAdd(&bins[bin], (unsigned long long)slot);
        const uint64_t mask = __ballot(has);
        const int rank = __popcll(mask & ((1ull << lane) - 1ull)), n_rec = __popcll(mask);
        const uint32_t pieces = has ? (slot + 15u) >> 4 : 0u;
        const uint32_t pend = wave_incl_sum(pieces, lane);
        const uint32_t total = (uint32_t)__shfl((int)pend, 63, 64);
        if (has) {
            s_src[wave][rank] = s; s_dst[wave][rank] = d0; s_len[wave][rank] = (uint32_t)min(len, (uint64_t)0xFFFFFFFFu);
            s_pend[wave][rank] = pend;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                                // lgkmcnt(0): the LDS stores above are visible to the wave
        for (uint32_t p = (uint32_t)lane; p < total; p += 64) {
            int lo = 0, hi = n_rec - 1;                                    // first record whose inclusive piece count exceeds p
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_pend[wave][mid] <= p) lo = mid + 1; else hi = mid;
            }
            const uint32_t rlen = s_len[wave][lo], rslot = (rlen + 1u + 7u) & ~7u;
            const uint32_t first = s_pend[wave][lo] - ((rslot + 15u) >> 4);
            const uint32_t c = (p - first) * 16u;
            const uint64_t src = s_src[wave][lo] + c, out = s_dst[wave][lo] + c;
            uint32_t w[4] = {0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};
            const int keep = (int)min(16u, rlen > c ? rlen - c : 0u);      // record bytes in this piece
            if (keep > 0) {
                if (src + 16 <= n) {
                    uint4 v;
                    __builtin_memcpy(&v, b + src, 16);
                    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
                } else {
                    for (int i = 0; i < keep; i++) {
                        const uint32_t ch = (uint8_t)b[src + i];
                        w[i >> 2] = (w[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | (ch << (8 * (i & 3)));
                    }
                }
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const int k = keep - 4 * d;
                    if (k <= 0) w[d] = 0x0A0A0A0Au;
                    else if (k < 4) { const uint32_t m = (1u << (8 * k)) - 1u; w[d] = (w[d] & m) | (0x0A0A0A0Au & ~m); }
                }
            }
            char *o8 = static_cast<char *>(__builtin_assume_aligned(dst + out, 8));        // slots are multiples of 8 bytes
            if (c + 16 <= rslot) __builtin_memcpy(o8, w, 16);
            else __builtin_memcpy(o8, w, 8);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// exclusive prefix of the bin sizes (in place); total -> *total; one workgroup
__global__ __launch_bounds__(1024) void bin_scan_kernel(unsigned long long *__restrict__ bins, uint32_t n_bins, unsigned long long *total)
{
    __shared__ unsigned long long s_part[1024];
    const int t = threadIdx.x;
    const uint32_t per = (n_bins + 1023u) / 1024u, a = min(n_bins, (uint32_t)t * per), e = min(n_bins, a + per);
    unsigned long long sum = 0;
    for (uint32_t i = a; i < e; i++) sum += bins[i];
    s_part[t] = sum;
    __syncthreads();
    if (t == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; i++) { const unsigned long long v = s_part[i]; s_part[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = s_part[t];
    for (uint32_t i = a; i < e; i++) { const unsigned long long v = bins[i]; bins[i] = run; run += v; }
}

int order_bits()
{
    static const int bits = [] {
        const char *e = getenv("SS_ORDER_BITS");
        const int v = e ? atoi(e) : 12;
        return std::min(MAX_BITS, std::max(1, v));
    }();
    return bits;
}

}  // namespace

namespace ss {

// src[0, n) (a flat base block on the device) -> a new buffer with the records
// binned; *out_d (hipMalloc'ed), *out_used (multiple of 16, '\n' padded), *out_cap.
int order_flat_dev(const char *src, uint64_t n, char **out_d, uint64_t *out_used, uint64_t *out_cap)
{
    *out_d = nullptr; *out_used = 0; *out_cap = 0;
    const int bits = order_bits();
    const uint32_t n_bins = (1u << bits) + 1u;
    static const bool trace = getenv("SS_INGEST_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        hipDeviceSynchronize();
        fprintf(stderr, "[reorder] %-22s at %.4f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    unsigned long long *d_bins = nullptr;
    char *d_new = nullptr;
#define SS_R(call) do { if ((call) != hipSuccess) { ss::set_last_error(#call, __FILE__, __LINE__, hipGetLastError()); hipFree(d_bins); hipFree(d_new); return SS_EHIP; } } while (0)
    SS_R(hipMalloc((void **)&d_bins, ((uint64_t)n_bins + 1) * 8));
    SS_R(hipMemsetAsync(d_bins, 0, ((uint64_t)n_bins + 1) * 8, 0));
    const unsigned nb = (unsigned)((n + RB - 1) / RB);
    hipLaunchKernelGGL((bin_kernel<false>), dim3(nb), dim3(256), 0, 0, src, n, bits, d_bins, (char *)nullptr);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, 0, d_bins, n_bins, d_bins + n_bins);
    unsigned long long total = 0;
    SS_R(hipMemcpy(&total, d_bins + n_bins, 8, hipMemcpyDeviceToHost));
    lap("count + scan");
    const uint64_t cap = std::max<uint64_t>((total + 15) & ~15ull, 16);
    SS_R(hipMalloc((void **)&d_new, cap));
    lap("new slab");
    hipLaunchKernelGGL((bin_kernel<true>), dim3(nb), dim3(256), 0, 0, src, n, bits, d_bins, d_new);
    if (cap > total) SS_R(hipMemsetAsync(d_new + total, '\n', cap - total, 0));
    SS_R(hipGetLastError());
    SS_R(hipDeviceSynchronize());
    lap("copy");
#undef SS_R
    hipFree(d_bins);
    *out_d = d_new; *out_used = cap; *out_cap = cap;
    return SS_OK;
}

bool reads_order_wanted()
{
    const char *e = getenv("SS_READS_ORDER");
    return !(e && (!strcmp(e, "file") || !strcmp(e, "0") || !strcmp(e, "off")));
}

int reads_order_for_locality(ss_reads *R, bool force)
{
    if (!R) return SS_EINVAL;
    if (!force && !reads_order_wanted()) return SS_OK;
    uint64_t bytes = 0;
    for (auto &sl : R->slabs) {
        if (sl.used >= 64) {
            char *d = nullptr;
            uint64_t used = 0, cap = 0;
            const int rc = order_flat_dev(sl.d, sl.used, &d, &used, &cap);
            if (rc) return rc;
            hipFree(sl.d);
            sl.d = d; sl.used = used; sl.cap = cap;
        }
        bytes += sl.cap;
    }
    R->device_bytes = bytes;
    return SS_OK;
}

}  // namespace ss
