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
//   bin  = top 12 to 22 bits (order_bits) of h = mix30(minimizer of the record's first 31 bases) -- the hash that addresses the
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

#include <mutex>
#include "ss_scan_dev.h"

#include <algorithm>
#include <chrono>
#include <vector>

namespace {

#ifndef SS_ORDER_CH
#define SS_ORDER_CH 64
#endif
constexpr int CH = SS_ORDER_CH;             // bytes of a slab owned by one lane (64 or 128): that many / 16 loads in flight
constexpr int RB = 256 * CH;                // ... and by one workgroup (16 KB: what bounds these passes is the chain of dependent
                                            // round trips of a workgroup -- load, neighbours, key bytes, atomic -- not its instructions)
constexpr int HALO = 512;                   // bytes behind the tile searched (in parallel) for the end of its last record
static_assert(22 <= 23, "a table entry has 24 bits for the bin; order_bits stays at or below 22");
constexpr uint32_t NO_NL = 0xFFFFFFFFu;     // "no newline" as a tile-relative position
constexpr int TCAP = 2 * CH;                // record starts per tile that the record table holds (reads of ~125 bases and more)

// newline mask of 16 bytes (bit i = byte i is '\n'): exact SWAR zero-byte test on w ^ 0x0A0A0A0A leaves 0x80 in the bytes
// that were '\n'; byte dot products with weights 1, 2, 4, 8 (x 16 for the odd dwords) gather the flags, 128 x the mask
__device__ __forceinline__ uint32_t nl_mask16(const uint4 v)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t z[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t x = w[d] ^ 0x0A0A0A0Au;
        z[d] = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    }
    const uint32_t lo = __builtin_amdgcn_udot4(z[1], 0x80402010u, __builtin_amdgcn_udot4(z[0], 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(z[3], 0x80402010u, __builtin_amdgcn_udot4(z[2], 0x08040201u, 0u, false), false);
    return (lo >> 7) | ((hi >> 7) << 8);
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

// ---- what a workgroup knows about its 4 KB tile -------------------------------------------------------------------
// Per-tile record table, written by the first pass and read by the second (so the second neither looks for newlines
// nor keys anything): entry = start in the tile (14 bits) | length (26 bits) | bin (24 bits); a tile with more than
// TCAP record starts (reads shorter than ~125 bases), or a record of 64 MB, says T_OVERFLOW and is discovered again.
constexpr uint32_t T_OVERFLOW = 0xFFFFFFFFu;
constexpr int START_BITS = CH == 64 ? 14 : 15, LEN_BITS = 40 - START_BITS;
constexpr uint32_t LEN_LIMIT = 1u << LEN_BITS;
static_assert(RB == 1 << START_BITS, "a table entry holds the start within the tile in START_BITS bits");

struct TileLds {
    uint32_t first[5];                      // first newline (tile-relative) of waves 0..3 and of the halo
    uint32_t wcnt[4];                       // records taken by each wave in the current round
    uint32_t more;                          // some lane has another record start left
    uint64_t src[256], dst[256];            // the round's records: where they start, where they go
    uint32_t len[256], bin[256], pend[256]; // length, bin, inclusive count of 16-byte pieces
};

#if SS_ORDER_CH == 128
typedef unsigned __int128 mask_t;           // one bit per byte of the lane's chunk
__device__ __forceinline__ uint32_t mask_ctz(mask_t m) { const uint64_t lo = (uint64_t)m; return lo ? (uint32_t)__builtin_ctzll(lo) : 64u + (uint32_t)__builtin_ctzll((uint64_t)(m >> 64)); }
#else
typedef uint64_t mask_t;
__device__ __forceinline__ uint32_t mask_ctz(mask_t m) { return (uint32_t)__builtin_ctzll(m); }
#endif
struct TileState { mask_t nl, st; uint32_t later; uint64_t tile0, i0; };

// loads the lane's 64 bytes, finds the record starts in them and the first newline behind them (tile + halo)
__device__ __forceinline__ void tile_setup(const char *__restrict__ b, uint64_t n, TileLds &L, TileState &S)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    S.tile0 = (uint64_t)blockIdx.x * RB;
    S.i0 = S.tile0 + (uint64_t)t * CH;
    mask_t nl = ~(mask_t)0;                             // beyond the buffer: newlines
    uint32_t prev = 1u;
    if (S.i0 < n) {
        uint4 v[CH / 16];
#pragma unroll
        for (int k = 0; k < CH / 16; k++) v[k] = load16_nl(b, S.i0 + 16u * k, n);
        prev = S.i0 == 0 ? 1u : (uint32_t)(b[S.i0 - 1] == '\n');
        nl = 0;
#pragma unroll
        for (int k = 0; k < CH / 16; k++) nl |= (mask_t)nl_mask16(v[k]) << (16 * k);
    }
    // first newline at or behind every lane's chunk: suffix minimum over the wave, then over the later waves and the halo
    uint32_t suf = nl ? (uint32_t)t * CH + mask_ctz(nl) : NO_NL;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_down((int)suf, off, 64);
        if (lane + off < 64) suf = min(suf, o);
    }
    if (lane == 0) L.first[wave] = suf;
    if (wave == 0) {                                    // halo: HALO bytes behind the tile, 32 lanes x 16 bytes
        uint32_t h = NO_NL;
        const uint64_t j0 = S.tile0 + RB + (uint64_t)lane * 16;
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
        if (lane == 0) L.first[4] = h;
    }
    __syncthreads();
    uint32_t later = (uint32_t)__shfl_down((int)suf, 1, 64);               // first newline behind this lane's chunk
    if (lane == 63) later = NO_NL;
    for (int w = wave + 1; w < 5; w++) later = min(later, L.first[w]);
    const mask_t before = (nl << 1) | prev;                                // bit i = byte i - 1 is a newline
    S.nl = nl; S.later = later;
    S.st = ~nl & before;                                                   // record starts in this chunk
}

// One round: every lane hands in its next record start (a 64-byte chunk starts at most one read, so there is usually one
// round); the records are numbered across the workgroup and land in L.src / L.len.  Returns their number; L.more tells
// whether another round is needed.
__device__ __forceinline__ uint32_t tile_round(const char *__restrict__ b, uint64_t n, TileLds &L, TileState &S)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool has = S.st != 0;
    uint64_t s = 0, len = 0;
    if (has) {
        const uint32_t bit = mask_ctz(S.st);
        S.st &= S.st - 1;
        s = S.i0 + bit;
        const mask_t up = bit + 1 < (uint32_t)CH ? S.nl & ~((((mask_t)1) << (bit + 1)) - 1) : (mask_t)0;
        uint64_t e;
        if (up) e = S.i0 + mask_ctz(up);
        else if (S.later != NO_NL) e = S.tile0 + S.later;
        else {                                                             // a record longer than the halo: walk on
            e = S.tile0 + RB + HALO;
            while (e < n) {
                const uint32_t m = nl_mask16(load16_nl(b, e, n));
                if (m) { e += (uint32_t)__builtin_ctz(m); break; }
                e += 16;
            }
            e = min(e, n);
        }
        len = e - s;
    }
    const uint64_t mask = __ballot(has);
    const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) L.wcnt[wave] = (uint32_t)__popcll(mask);
    if (t == 0) L.more = 0;
    __syncthreads();
    uint32_t base = 0, cnt = 0;
    for (int w = 0; w < 4; w++) { const uint32_t c = L.wcnt[w]; if (w < wave) base += c; cnt += c; }
    if (has) { L.src[base + rank] = s; L.len[base + rank] = (uint32_t)min(len, (uint64_t)0xFFFFFFFFu); }
    if (S.st != 0) L.more = 1;
    __syncthreads();
    return cnt;
}

#ifndef SS_SLOT_ALIGN
#define SS_SLOT_ALIGN 8u
#endif
__host__ __device__ __forceinline__ uint32_t slot_of(uint32_t len) { return (len + 1u + (SS_SLOT_ALIGN - 1u)) & ~(SS_SLOT_ALIGN - 1u); }     // record + '\n', padded to 8 bytes

// ---- pass 1: bytes per bin, and the tile's record table -----------------------------------------------------------------
__global__ __launch_bounds__(256, 8) __attribute__((amdgpu_num_sgpr(80))) void count_kernel(const char *__restrict__ b, uint64_t n, int bits, unsigned long long *__restrict__ hist,
                                                    uint32_t *__restrict__ tab_cnt, unsigned long long *__restrict__ tab,
                                                    unsigned long long *__restrict__ n_overflow)
{
    __shared__ TileLds L;
    TileState S;
    tile_setup(b, n, L, S);
    const int t = threadIdx.x;
    bool first = true;
    while (true) {
        const uint32_t cnt = tile_round(b, n, L, S);
        const bool more = L.more != 0;
        uint32_t bin = 0, len = 0;
        bool big = false;
        if ((uint32_t)t < cnt) {
            const uint64_t s = L.src[t];
            len = L.len[t];
            bin = (s + 32 <= n) ? record_bin(b, s, len, bits) : (1u << bits);
            atomicAdd(&hist[bin], (unsigned long long)slot_of(len));
            big = len >= LEN_LIMIT;
        }
        if (first) {
            const bool fits = !more && cnt <= (uint32_t)TCAP && !__syncthreads_or(big);
            if (fits && (uint32_t)t < cnt)
                tab[(uint64_t)blockIdx.x * TCAP + t] = (unsigned long long)(L.src[t] - S.tile0) | ((unsigned long long)len << START_BITS) |
                                                        ((unsigned long long)bin << (START_BITS + LEN_BITS));
            if (t == 0) {
                tab_cnt[blockIdx.x] = fits ? cnt : T_OVERFLOW;
                if (!fits) atomicAdd(n_overflow, 1ull);
            }
        }
        first = false;
        if (!more) break;
        __syncthreads();
    }
}

// ---- pass 2: claim the slots, copy -----------------------------------------------------------------------------------------
// The workgroup copies its records together: their 16-byte pieces are numbered across the workgroup (prefix sum of the
// piece counts), every lane finds the record of its piece by binary search in LDS, loads 16 unaligned bytes, pads behind
// the record's end with '\n' and stores them aligned (slots are multiples of 8 bytes).
__device__ __forceinline__ void copy_round(const char *__restrict__ b, uint64_t n, char *__restrict__ dst, TileLds &L, uint32_t cnt)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t pieces = (uint32_t)t < cnt ? (slot_of(L.len[t]) + 15u) >> 4 : 0u;
    uint32_t pend = wave_incl_sum(pieces, lane);
    if (lane == 63) L.wcnt[wave] = pend;
    __syncthreads();
    uint32_t total = 0;
    for (int w = 0; w < 4; w++) { const uint32_t c = L.wcnt[w]; if (w < wave) pend += c; total += c; }
    if ((uint32_t)t < cnt) L.pend[t] = pend;
    __syncthreads();
    for (uint32_t p = (uint32_t)t; p < total; p += 256) {
        int lo = 0, hi = (int)cnt - 1;                                     // first record whose inclusive piece count exceeds p
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (L.pend[mid] <= p) lo = mid + 1; else hi = mid;
        }
        const uint32_t rlen = L.len[lo], rslot = slot_of(rlen);
        const uint32_t c = (p - (L.pend[lo] - ((rslot + 15u) >> 4))) * 16u;
        const uint64_t src = L.src[lo] + c, out = L.dst[lo] + c;
        uint32_t w[4] = {0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};
        const int keep = (int)min(16u, rlen > c ? rlen - c : 0u);          // record bytes in this piece
        if (keep > 0) {
            const uint4 v = load16_nl(b, src, n);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            if (keep < 16) {
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const int k = keep - 4 * d;
                    if (k <= 0) w[d] = 0x0A0A0A0Au;
                    else if (k < 4) { const uint32_t m = (1u << (8 * k)) - 1u; w[d] = (w[d] & m) | (0x0A0A0A0Au & ~m); }
                }
            }
        }
        char *o8 = static_cast<char *>(__builtin_assume_aligned(dst + out, 8));
        if (c + 16 <= rslot) __builtin_memcpy(o8, w, 16);
        else __builtin_memcpy(o8, w, 8);
    }
}

// one piece: the record's bytes from `v`, '\n' behind its end, stored aligned (slots are multiples of 8 bytes)
__device__ __forceinline__ void store_piece(char *__restrict__ dst, uint64_t out, uint4 v, uint32_t rlen, uint32_t rslot, uint32_t c)
{
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
    const int keep = (int)min(16u, rlen > c ? rlen - c : 0u);              // record bytes in this piece
    if (keep < 16) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int k = keep - 4 * d;
            if (k <= 0) w[d] = 0x0A0A0A0Au;
            else if (k < 4) { const uint32_t m = (1u << (8 * k)) - 1u; w[d] = (w[d] & m) | (0x0A0A0A0Au & ~m); }
        }
    }
    char *o8 = static_cast<char *>(__builtin_assume_aligned(dst + out, 8));
#ifdef SS_PLACE_NT      // (A/B builds: the slab is written once and read by a later kernel -- stores that do not allocate in L2)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef u32x2 u32x2_a8 __attribute__((aligned(8)));
    __builtin_nontemporal_store((u32x2){w[0], w[1]}, reinterpret_cast<u32x2_a8 *>(o8));
    if (c + 16 <= rslot) __builtin_nontemporal_store((u32x2){w[2], w[3]}, reinterpret_cast<u32x2_a8 *>(o8 + 8));
#else
    if (c + 16 <= rslot) __builtin_memcpy(o8, w, 16);
    else __builtin_memcpy(o8, w, 8);
#endif
}

// the usual tile: everything is in the table.  The chain of dependent round trips is what bounds this pass, so it is kept
// short: table entries and count together; the returning atomics that claim the slots are ISSUED, and while they are on
// their way every lane finds its pieces (up to PRE of them: ~1100 pieces per tile of 150-base reads over 256 lanes) and
// loads their bytes -- neither needs the destination; only the stores wait for it.
constexpr int PRE = 6;
__global__ __launch_bounds__(256, 8) __attribute__((amdgpu_num_sgpr(80))) void place_kernel(
    const char *__restrict__ b, uint64_t n, unsigned long long *__restrict__ cursor, const uint32_t *__restrict__ tab_cnt,
    const unsigned long long *__restrict__ tab, char *__restrict__ dst)
{
    __shared__ TileLds L;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    static_assert(TCAP <= 256, "one table entry per thread");
    const unsigned long long e = t < TCAP ? tab[(uint64_t)blockIdx.x * TCAP + t] : 0ull;     // (in flight together with the count)
    const uint32_t known = tab_cnt[blockIdx.x];
    if (known == T_OVERFLOW || known == 0) return;
    uint64_t d0 = 0;
    uint32_t pieces = 0;
    if ((uint32_t)t < known) {
        const uint32_t len = (uint32_t)(e >> START_BITS) & (LEN_LIMIT - 1u);
        L.src[t] = (uint64_t)blockIdx.x * RB + (uint32_t)(e & (uint32_t)(RB - 1));
        L.len[t] = len;
        d0 = atomicAdd(&cursor[(uint32_t)(e >> (START_BITS + LEN_BITS))], (unsigned long long)slot_of(len));      // (answer needed at the stores)
        pieces = (slot_of(len) + 15u) >> 4;
    }
    uint32_t pend = wave_incl_sum(pieces, lane);
    if (lane == 63) L.wcnt[wave] = pend;
    __syncthreads();
    uint32_t total = 0;
    for (int w = 0; w < 4; w++) { const uint32_t c = L.wcnt[w]; if (w < wave) pend += c; total += c; }
    if ((uint32_t)t < known) L.pend[t] = pend;
    __syncthreads();
    uint4 v[PRE];
    uint32_t rec[PRE], off[PRE];
#pragma unroll
    for (int r = 0; r < PRE; r++) {
        const uint32_t p = (uint32_t)t + 256u * r;
        rec[r] = 0; off[r] = 0;
        v[r] = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
        if (p < total) {
            int lo = 0, hi = (int)known - 1;                               // first record whose inclusive piece count exceeds p
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (L.pend[mid] <= p) lo = mid + 1; else hi = mid;
            }
            const uint32_t rlen = L.len[lo], c = (p - (L.pend[lo] - ((slot_of(rlen) + 15u) >> 4))) * 16u;
            rec[r] = (uint32_t)lo; off[r] = c;
            if (rlen > c) v[r] = load16_nl(b, L.src[lo] + c, n);
        }
    }
    if ((uint32_t)t < known) L.dst[t] = d0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PRE; r++) {
        const uint32_t p = (uint32_t)t + 256u * r;
        if (p < total) {
            const uint32_t rlen = L.len[rec[r]];
            store_piece(dst, L.dst[rec[r]] + off[r], v[r], rlen, slot_of(rlen), off[r]);
        }
    }
    for (uint32_t p = (uint32_t)t + 256u * PRE; p < total; p += 256) {    // long records: the rest, piece by piece
        int lo = 0, hi = (int)known - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (L.pend[mid] <= p) lo = mid + 1; else hi = mid;
        }
        const uint32_t rlen = L.len[lo], rslot = slot_of(rlen), c = (p - (L.pend[lo] - ((rslot + 15u) >> 4))) * 16u;
        const uint4 x = rlen > c ? load16_nl(b, L.src[lo] + c, n) : make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
        store_piece(dst, L.dst[lo] + c, x, rlen, rslot, c);
    }
}

// a tile whose records did not fit the table (short reads, several record starts in one lane's chunk): found and keyed again
__global__ __launch_bounds__(256) void place_again_kernel(const char *__restrict__ b, uint64_t n, int bits, unsigned long long *__restrict__ cursor,
                                                          const uint32_t *__restrict__ tab_cnt, char *__restrict__ dst)
{
    __shared__ TileLds L;
    const int t = threadIdx.x;
    if (tab_cnt[blockIdx.x] != T_OVERFLOW) return;
    TileState S;
    tile_setup(b, n, L, S);
    while (true) {
        const uint32_t cnt = tile_round(b, n, L, S);
        const bool more = L.more != 0;
        if ((uint32_t)t < cnt) {
            const uint64_t s = L.src[t];
            const uint32_t len = L.len[t];
            const uint32_t bin = (s + 32 <= n) ? record_bin(b, s, len, bits) : (1u << bits);
            L.dst[t] = atomicAdd(&cursor[bin], (unsigned long long)slot_of(len));
        }
        __syncthreads();
        copy_round(b, n, dst, L, cnt);
        if (!more) break;
        __syncthreads();
    }
}

// ---- records of ONE length (what a sequencer writes: every read 150 bases) ---------------------------------------------
// The general passes above spend their time finding out where records begin and end -- newline masks, a suffix minimum
// over the tile, a halo, record tables in LDS, a binary search per copied piece: chains of dependent round trips that hold
// them at 2.4-2.5 TB/s (profiles/r05_sampled_kernel_stats.csv: 1.28 + 2.43 ms per 20 M reads).  When every record of the
// slab has the same length L (the slab is n_rec x (L + 1) bytes, the first newline says L) record i starts at i * (L + 1)
// and nothing has to be found: a WAVE owns 64 consecutive records (9.7 KB for L = 150), no LDS, no barrier.
//   count_fixed   streams the wave's span in 16-byte pieces and CHECKS it -- a newline at offset L of every record and
//                 nowhere else; one violation anywhere sets a flag and the caller runs the general passes instead -- and
//                 every lane keys its own record from its first 32 bytes (in cache by then); bin sizes by atomicAdd, the
//                 record's bin kept (4 bytes per record) for the second pass
//   place_fixed   the returning atomicAdd on the bin's cursor is issued first; while it is on its way the lanes load the
//                 span's pieces (numbered across the wave: consecutive lanes, consecutive 16 bytes); the destination of
//                 a piece's record comes from the owning lane by a wave shuffle; aligned stores, padded with '\n'
constexpr uint32_t FIX_MIN_L = 32, FIX_MAX_L = 1023;

__global__ __launch_bounds__(256, 8) void count_fixed_kernel(const char *__restrict__ b, uint64_t n, uint64_t n_rec, uint32_t L, uint32_t magic_l1, int bits,
                                                             unsigned long long *__restrict__ hist, uint32_t *__restrict__ bins,
                                                             unsigned long long *__restrict__ not_fixed)
{
    const int lane = threadIdx.x & 63;
    const uint64_t r0 = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u;
    if (r0 >= n_rec) return;
    const uint32_t L1 = L + 1u, nr = (uint32_t)min((uint64_t)64, n_rec - r0), span = nr * L1;
    const uint64_t base = r0 * L1;
    bool bad = false;
#ifndef SS_COUNT_NOVERIFY          // (A/B builds only: what the count pass costs when it reads each record's first bytes and nothing else)
    for (uint32_t off = (uint32_t)lane * 16u; off < span; off += 1024u) {
        const uint32_t m = nl_mask16(load16_nl(b, base + off, n));
        const uint32_t pos = off - __umulhi(off, magic_l1) * L1;           // offset of the piece's first byte within its record
        const uint32_t valid = span - off >= 16u ? 0xFFFFu : (1u << (span - off)) - 1u;
        const uint32_t want = L - pos < 16u ? 1u << (L - pos) : 0u;        // (L >= 32: at most one record end in 16 bytes)
        bad |= ((m ^ want) & valid) != 0u;
    }
#endif
    if (r0 + nr == n_rec) {                                                // behind the last record: newlines only (padding)
        for (uint64_t i = base + span + (uint32_t)lane; i < n; i += 64) bad |= b[i] != '\n';
    }
    if (__ballot(bad)) { if (lane == 0) atomicOr(not_fixed, 1ull); return; }
    if ((uint32_t)lane < nr) {
        const uint32_t bin = record_bin(b, base + (uint64_t)lane * L1, L, bits);      // (s + 32 <= s + L + 1 <= n)
        bins[r0 + lane] = bin;
        atomicAdd(&hist[bin], (unsigned long long)slot_of(L));
    }
}

constexpr int FPRE = 5;                     // pieces a lane has in flight: 64 lanes x 5 = the 640 pieces of 64 records of 150 bases
__global__ __launch_bounds__(256, 8) void place_fixed_kernel(const char *__restrict__ b, uint64_t n, uint64_t n_rec, uint32_t L, uint32_t magic_p,
                                                             unsigned long long *__restrict__ cursor, const uint32_t *__restrict__ bins,
                                                             char *__restrict__ dst)
{
    const int lane = threadIdx.x & 63;
    const uint64_t r0 = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u;
    if (r0 >= n_rec) return;
    const uint32_t L1 = L + 1u, nr = (uint32_t)min((uint64_t)64, n_rec - r0), slot = slot_of(L), P = (slot + 15u) >> 4, total = nr * P;
    const uint64_t base = r0 * L1;
    unsigned long long d0 = 0;
#if defined(SS_PLACE_SEQ)          // (A/B builds, results unbinned: what the pass costs as a plain copy -- no atomics, sequential destinations)
    if ((uint32_t)lane < nr) d0 = (r0 + (uint64_t)lane) * slot;
#elif defined(SS_PLACE_HASH)       // (A/B builds, results WRONG: scattered like the real thing, no atomics)
    if ((uint32_t)lane < nr) d0 = ((((r0 + (uint64_t)lane) * 0x9E3779B97F4A7C15ull) >> 20) % n_rec) * slot;
#else
    if ((uint32_t)lane < nr) d0 = atomicAdd(&cursor[bins[r0 + lane]], (unsigned long long)slot);      // (answer needed at the stores)
#endif
    for (uint32_t p0 = 0; p0 < total; p0 += 64u * FPRE) {
        uint4 v[FPRE];
        uint32_t rec[FPRE], off[FPRE];
#pragma unroll
        for (int r = 0; r < FPRE; r++) {
            const uint32_t p = p0 + (uint32_t)lane + 64u * r;
            rec[r] = min(__umulhi(p, magic_p), nr - 1u);
            off[r] = (p - rec[r] * P) * 16u;
            v[r] = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
            if (p < total && L > off[r]) v[r] = load16_nl(b, base + (uint64_t)rec[r] * L1 + off[r], n);
        }
#pragma unroll
        for (int r = 0; r < FPRE; r++) {
            const uint32_t p = p0 + (uint32_t)lane + 64u * r;
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)d0, (int)rec[r], 64), hi = (uint32_t)__shfl((int)(uint32_t)(d0 >> 32), (int)rec[r], 64);
            if (p < total) store_piece(dst, (((uint64_t)hi << 32) | lo) + off[r], v[r], L, slot, off[r]);
        }
    }
}

// ---- exclusive prefix over the bin sizes (up to 4 M of them): block sums, their prefix, local prefixes ----------------------
constexpr int SCAN_PER = 4096;              // entries per workgroup of 1024 threads

__global__ __launch_bounds__(1024) void scan_sums_kernel(const unsigned long long *__restrict__ v, uint32_t n, unsigned long long *__restrict__ sums)
{
    __shared__ unsigned long long s_w[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t i0 = blockIdx.x * SCAN_PER + (uint32_t)t * 4u;
    unsigned long long x = 0;
    for (uint32_t i = i0; i < min(n, i0 + 4u); i++) x += v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += (unsigned long long)__shfl_xor((long long)x, off, 64);
    if (lane == 0) s_w[wave] = x;
    __syncthreads();
    if (t == 0) { unsigned long long r = 0; for (int w = 0; w < 16; w++) r += s_w[w]; sums[blockIdx.x] = r; }
}

__global__ __launch_bounds__(1024) void scan_top_kernel(unsigned long long *__restrict__ sums, uint32_t nb, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long s_part[1024];
    const int t = threadIdx.x;
    const uint32_t per = (nb + 1023u) / 1024u, a = min(nb, (uint32_t)t * per), e = min(nb, a + per);
    unsigned long long sum = 0;
    for (uint32_t i = a; i < e; i++) sum += sums[i];
    s_part[t] = sum;
    __syncthreads();
    if (t == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; i++) { const unsigned long long v = s_part[i]; s_part[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = s_part[t];
    for (uint32_t i = a; i < e; i++) { const unsigned long long v = sums[i]; sums[i] = run; run += v; }
}

__global__ __launch_bounds__(1024) void scan_apply_kernel(unsigned long long *__restrict__ v, uint32_t n, const unsigned long long *__restrict__ sums)
{
    __shared__ unsigned long long s_w[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t i0 = blockIdx.x * SCAN_PER + (uint32_t)t * 4u;
    unsigned long long x[4], mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { x[k] = i0 + k < n ? v[i0 + k] : 0ull; mine += x[k]; }
    unsigned long long incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = (unsigned long long)__shfl_up((long long)incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned long long run = sums[blockIdx.x] + incl - mine;
    for (int w = 0; w < wave; w++) run += s_w[w];
#pragma unroll
    for (int k = 0; k < 4; k++) { if (i0 + k < n) v[i0 + k] = run; run += x[k]; }
}

// bin width: about four records per bin for the block at hand (reads that share their first minimizer then sit in the
// same scan tile or the next), 12 bits at least, 22 at most (4 M counters = 32 MB)
int order_bits(uint64_t n_bytes)
{
    int bits = 12;
    while (bits < 22 && (n_bytes / 152) >> (bits + 2)) bits++;
    return bits;
}

}  // namespace

namespace ss {

static std::mutex g_scr_mu;
static uint64_t g_order_n[2] = {0, 0};        // slabs binned by the one-length passes / by the general ones (ss_reads_order_counters)
static double g_order_ms[3] = {0, 0, 0};      // the last order_flat_dev: count + prefix, allocation of the new slab, place
static char *g_scr = nullptr;            // the scratch of the last call (bin cursors, per-tile record tables), kept for the next
static uint64_t g_scr_cap = 0;

// src[0, n) (a flat base block on the device) -> a new buffer with the records
// binned; *out_d (hipMalloc'ed), *out_used (multiple of 16, '\n' padded), *out_cap.
int order_flat_dev(const char *src, uint64_t n, char **out_d, uint64_t *out_used, uint64_t *out_cap)
{
    *out_d = nullptr; *out_used = 0; *out_cap = 0;
    const int bits = order_bits(n);
    const uint32_t n_bins = (1u << bits) + 1u;
    static const bool trace = getenv("SS_INGEST_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        hipDeviceSynchronize();
        fprintf(stderr, "[reorder] %-22s at %.4f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    const unsigned nb = (unsigned)((n + RB - 1) / RB), nsb = (n_bins + SCAN_PER - 1) / SCAN_PER;
    // one allocation for the scratch: bin sizes / cursors (+ total), block sums of the prefix, per-tile record counts and table
    const uint64_t o_sums = ((uint64_t)n_bins + 2) * 8, o_cnt = o_sums + (((uint64_t)nsb + 1) * 8), o_tab = (o_cnt + (uint64_t)nb * 4 + 255) & ~255ull;
    // records of one length: the first newline says which; the count pass checks every record against it
    uint32_t fix_L = 0;
    uint64_t n_rec = 0;
    static const bool fixed_allowed = [] { const char *e = getenv("SS_ORDER_FIXED"); return !(e && !strcmp(e, "0")); }();
    if (fixed_allowed && n >= 64) {
        char head[FIX_MAX_L + 2];
        const size_t hn = (size_t)std::min<uint64_t>(n, sizeof(head));
        if (hipMemcpy(head, src, hn, hipMemcpyDeviceToHost) != hipSuccess) { ss::set_last_error("hipMemcpy", __FILE__, __LINE__, hipGetLastError()); return SS_EHIP; }
        const void *nl = memchr(head, '\n', hn);
        if (nl) {
            const uint32_t L = (uint32_t)((const char *)nl - head);
            if (L >= FIX_MIN_L && L <= FIX_MAX_L && n / (L + 1) >= 1 && n - (n / (L + 1)) * (L + 1) < 64) { fix_L = L; n_rec = n / (L + 1); }
        }
    }
    // (the record table of the general passes and the 4-byte bins of the one-length passes share a region: whichever is larger)
    const uint64_t scratch = o_tab + std::max<uint64_t>((uint64_t)nb * TCAP * 8, (n_rec * 4 + 255) & ~255ull);
    char *d_scr = nullptr, *d_new = nullptr;
#define SS_R(call) do { if ((call) != hipSuccess) { ss::set_last_error(#call, __FILE__, __LINE__, hipGetLastError()); hipFree(d_scr); hipFree(d_new); return SS_EHIP; } } while (0)
    // (the scratch of the call before is kept -- 0.19 GB for 20 M reads --: two driver calls fewer per sample)
    uint64_t scr_cap = 0;
    {
        std::lock_guard<std::mutex> g(g_scr_mu);
        if (g_scr && g_scr_cap >= scratch) { d_scr = g_scr; scr_cap = g_scr_cap; g_scr = nullptr; g_scr_cap = 0; }
    }
    if (!d_scr) { SS_R(hipMalloc((void **)&d_scr, scratch)); scr_cap = scratch; }
    unsigned long long *d_hist = (unsigned long long *)d_scr, *d_sums = (unsigned long long *)(d_scr + o_sums);
    uint32_t *d_cnt = (uint32_t *)(d_scr + o_cnt);
    unsigned long long *d_tab = (unsigned long long *)(d_scr + o_tab);
    constexpr unsigned pad1 = 0, pad2 = 0;
    unsigned long long tail[2] = {0, 0};                 // bytes of the new slab; tiles that did not fit the table / "not of one length"
    bool fixed = false;
    if (fix_L) {
        const uint32_t L1 = fix_L + 1u, P = (::slot_of(fix_L) + 15u) >> 4;
        const unsigned nbf = (unsigned)((n_rec + 255) / 256);
        SS_R(hipMemsetAsync(d_hist, 0, o_sums, 0));
        hipLaunchKernelGGL(count_fixed_kernel, dim3(nbf), dim3(256), 0, 0, src, n, n_rec, fix_L, (uint32_t)(((1ull << 32) + L1 - 1) / L1), bits, d_hist,
                           (uint32_t *)d_tab, d_hist + n_bins + 1);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(nsb), dim3(1024), 0, 0, d_hist, n_bins, d_sums);
        hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, 0, d_sums, nsb, d_hist + n_bins);
        hipLaunchKernelGGL(scan_apply_kernel, dim3(nsb), dim3(1024), 0, 0, d_hist, n_bins, d_sums);
        SS_R(hipMemcpy(tail, d_hist + n_bins, 16, hipMemcpyDeviceToHost));
        fixed = tail[1] == 0;                            // (else: some record is shorter or longer after all -- the general passes)
        (void)P;
    }
    if (!fixed) {
        SS_R(hipMemsetAsync(d_hist, 0, o_sums, 0));
        hipLaunchKernelGGL(count_kernel, dim3(nb), dim3(256), pad1, 0, src, n, bits, d_hist, d_cnt, d_tab, d_hist + n_bins + 1);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(nsb), dim3(1024), 0, 0, d_hist, n_bins, d_sums);
        hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, 0, d_sums, nsb, d_hist + n_bins);
        hipLaunchKernelGGL(scan_apply_kernel, dim3(nsb), dim3(1024), 0, 0, d_hist, n_bins, d_sums);
        SS_R(hipMemcpy(tail, d_hist + n_bins, 16, hipMemcpyDeviceToHost));
    }
    const unsigned long long total = tail[0];
    lap("count + prefix");
    const auto t_counted = std::chrono::steady_clock::now();
    const uint64_t cap = std::max<uint64_t>((total + 15) & ~15ull, 16);
    uint64_t real_cap = cap;                             // (a kept block may be larger)
    SS_R(ss::big_malloc((void **)&d_new, cap, &real_cap));
    const auto t_alloc = std::chrono::steady_clock::now();
    lap("new slab");
    if (fixed) {
        const uint32_t P = (::slot_of(fix_L) + 15u) >> 4;
        hipLaunchKernelGGL(place_fixed_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, 0, src, n, n_rec, fix_L, (uint32_t)(((1ull << 32) + P - 1) / P),
                           d_hist, (const uint32_t *)d_tab, d_new);
    } else {
        hipLaunchKernelGGL(place_kernel, dim3(nb), dim3(256), pad2, 0, src, n, d_hist, d_cnt, d_tab, d_new);
        if (tail[1]) hipLaunchKernelGGL(place_again_kernel, dim3(nb), dim3(256), 0, 0, src, n, bits, d_hist, d_cnt, d_new);
    }
    if (cap > total) SS_R(hipMemsetAsync(d_new + total, '\n', cap - total, 0));
    SS_R(hipGetLastError());
    SS_R(hipDeviceSynchronize());
    lap("place");
    {
        // where the call's time went (ss_reads_order_timing): the driver's allocation of the new slab is not the kernels' time,
        // and on some boxes a fresh 3 GB allocation takes 60-90 ms
        const auto t_end = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::lock_guard<std::mutex> g(g_scr_mu);
        g_order_ms[0] = ms(t_begin, t_counted); g_order_ms[1] = ms(t_counted, t_alloc); g_order_ms[2] = ms(t_alloc, t_end);
        g_order_n[fixed ? 0 : 1]++;
    }
#undef SS_R
    {
        std::lock_guard<std::mutex> g(g_scr_mu);
        if (!g_scr || g_scr_cap < scr_cap) { std::swap(g_scr, d_scr); std::swap(g_scr_cap, scr_cap); }
    }
    if (d_scr) hipFree(d_scr);
    *out_d = d_new; *out_used = cap; *out_cap = real_cap;
    return SS_OK;
}

void reorder_timing(double out[3])
{
    std::lock_guard<std::mutex> g(g_scr_mu);
    for (int i = 0; i < 3; i++) out[i] = g_order_ms[i];
}

void reorder_counters(uint64_t out[2])
{
    std::lock_guard<std::mutex> g(g_scr_mu);
    out[0] = g_order_n[0]; out[1] = g_order_n[1];
}

void reorder_release()
{
    char *d = nullptr;
    {
        std::lock_guard<std::mutex> g(g_scr_mu);
        std::swap(d, g_scr);
        g_scr_cap = 0;
    }
    if (d) hipFree(d);
}

// Policy: ALWAYS, unless SS_READS_ORDER=file.  Binning 20 M reads costs ~3.6 ms of kernel time once per sample; a tree scan of
// the binned set is 1.8 ms faster than in file order on sampled node sets (5.6 -> 3.8 ms), 0.7 ms on contiguous ones, a
// cluster scan 6 ms (16.8 -> 10.6: the hits of a locus' reads are added up in LDS).  So it pays from the SECOND scan of a sample
// on -- the tree scan + one cluster's scan, or the two scans of -b -- and a sample that is scanned exactly once (every
// identified cluster single-strain) loses ~1.5 ms per 20 M reads, beside ~80 ms of text ingest for the same reads.  The
// loader cannot know which it will be: the clusters are identified by the first scan.
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
    // (asked ONCE: the driver takes ~1 ms to answer, as long as the binning of a 1 M-read file; every slab replaces one of about its size)
    size_t mem_free = 0, mem_total = 0;
    const bool known = hipMemGetInfo(&mem_free, &mem_total) == hipSuccess;
    for (auto &sl : R->slabs) {
        // (the binned copy lives beside the slab until it replaces it: a slab that leaves no room for that stays in file order)
        const bool room = known && mem_free > sl.used + sl.used / 8 + (1ull << 30);
        if (sl.used >= 64 && (room || force)) {
            char *d = nullptr;
            uint64_t used = 0, cap = 0;
            const int rc = order_flat_dev(sl.d, sl.used, &d, &used, &cap);
            if (rc) return rc;
            ss::big_put(sl.d, sl.cap);                    // (kept for the next slab's binned copy)
            sl.d = d; sl.used = used; sl.cap = cap; sl.binned = true;
        }
        bytes += sl.cap;
    }
    R->device_bytes = bytes;
    return SS_OK;
}

}  // namespace ss
