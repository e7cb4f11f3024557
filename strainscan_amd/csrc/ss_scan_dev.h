// Device helpers shared by the scan kernels (flat table: ss_scan.hip, minimizer buckets: ss_mini.hip).
#pragma once
#include "ss_common.h"
#include <stdlib.h>

namespace ss {

// ---- minimizers of the k = 31 index (ss_mini.hip) and of the locality order of resident reads (ss_reorder.hip) ----
constexpr int MINI_M = 15;                       // minimizer length (30 bits)
constexpr uint32_t M30 = 0x3FFFFFFFu;

// Ordering key of a 30-bit m-mer: the m-mer with the smallest key, leftmost on ties, is the
// minimizer of a k-mer.  key = lo24(x) * C1 + C0 with C1 = an odd 19-bit constant << 5 and C0 a
// multiple of 32: ONE v_mad_u32_u24 per m-mer on the device (full rate; the 24-bit multiplier
// ignores the operand's upper bits, so the device never masks the window).  The key orders the
// m-mers by their first 12 bases (a bijection of those 24 bits onto the 27 key bits); m-mers that
// agree there tie and resolve leftmost like any other tie.  The low five bits are zero by
// construction: the kernel adds the m-mer's index there (for free, inside C0), so ONE v_min_u32
// per step compares (key, position).  The constant keeps poly-A from being everybody's minimizer.
// It need not be injective: a bucket is named by the m-mer itself and holds full k-mers.
constexpr uint32_t MMK_C1 = 0x4F1BBu << 5, MMK_C0 = 0x7F4A7C00u;
__host__ __device__ __forceinline__ uint32_t mmkey(uint32_t x)
{
    return (x & 0xFFFFFFu) * MMK_C1 + MMK_C0;
}
constexpr uint32_t KEY_MASK = ~31u;

// Once per RUN: h = mix30(minimizer), a bijection of the 30 bits (odd multipliers modulo 2^30, xor-shifts).
// Page = high half of (h << 2) x n_pages (any table size, no power-of-two rounding); tags = low bits of h.
__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
__host__ __device__ __forceinline__ uint32_t mix30(uint32_t x)
{
    uint32_t h = (x * 0x9E3779B1u) & M30;
    h ^= h >> 15;
    h = (h * 0x2C1B3C6Du) & M30;
    h ^= h >> 14;
    return h;
}

// ---- the page index of ss_mini.hip (built by ss_mini.hip on the host, by ss_build_dev.hip on the device) ----
constexpr uint32_t HDR_MULTI = 1u << 17;
__host__ __device__ __forceinline__ uint32_t page_of(uint32_t h, uint32_t n_pages) { return mulhi32(h << 2, n_pages); }
constexpr uint32_t PG_SLOTS = 8;                          // slots per 64-byte page
constexpr uint32_t PG_MIN_PAGES = 4096;                   // >= 2^11: (page, low 20 bits of h) determines h
// A page, structure of arrays, so that ONE 16-byte load decides almost every lookup:
//   bytes  0.. 7  tag8[8]   h & 0xFF of the slot's minimizer                         (0xFF when empty)
//   bytes  8..15  hi8[8]    inline k-mer: e = 16 - offset (5 bits); bucket reference: 0x80 | mask bit 16 << 6 | h[13:8]
//                           (0x7F when empty: an inline slot with the impossible e = 31)
//   bytes 16..47  lo32[8]   inline: flank32; reference: multi << 31 | bucket start
//   bytes 48..63  mid16[8]  inline: h[19:8] << 4; reference: mask bits 0..15
// Only slots whose tag8 (and e range / reference filter bits) match have their lo32 / mid16 read, from the sector the
// first load has just brought into the L1.
constexpr uint8_t PG_EMPTY_TAG = 0xFF, PG_EMPTY_HI = 0x7F;
constexpr uint32_t START_MASK = 0x3FFFFFFFu;
// bit 30 of a bucket reference's lo32 (round 4): the bucket is SOLID -- one database k-mer per offset, the offsets a contiguous
// range, and the k-mer of offset o + 1 is the k-mer of offset o moved one base to the left: its k-mers are one stretch of
// cnt + 30 bases (a super-k-mer), and a run of a read is verified against that stretch by ONE lane (ss_mini.hip phase 3a)
constexpr uint32_t PG_SOLID = 1u << 30;
// How many database k-mers of one minimizer are kept INLINE in page slots (more go to a bucket).  Two, where a table's
// minimizers mostly own one or two k-mers (the sampled node sets of a tree: one sector answers a lookup); NONE, where such
// minimizers are the exception (every k-mer of a genome: 3 % of them) -- there an inline k-mer's hit is an atomic of its own,
// one (instruction, line) request per hit, while a bucket's hits share lines and, in a cluster scan, the LDS counters: the
// cluster scan 7.4 -> 6.5 ms without inline k-mers, a contiguous tree table unchanged (round 4).  SS_INLINE_MAX overrides.
inline uint32_t choose_inline_max(uint64_t kmers_of_small_minimizers /* <= 2 k-mers */, uint64_t n_distinct)
{
    if (const char *e = getenv("SS_INLINE_MAX")) { const int v = atoi(e); return (uint32_t)(v < 0 ? 0 : v > 8 ? 8 : v); }
    return kmers_of_small_minimizers * 8 < n_distinct ? 0u : 2u;
}
// the 16 bases of a k-mer that are not its minimizer: rotate the 62-bit key right by 2 * offset (the minimizer
// comes to stand in bits 0..29), the upper 32 bits = bases behind the minimizer, then the bases in front of it
__host__ __device__ __forceinline__ uint32_t flank_of_key(uint64_t key, uint32_t off)
{
    const uint64_t M62 = (1ull << 62) - 1;
    const uint64_t r = off ? (((key >> (2 * off)) | (key << (62 - 2 * off))) & M62) : key;
    return (uint32_t)(r >> 30);
}

// the same for a k-mer of any length MINI_M + 2 <= k <= 31 (round 6: `-k` other than 31 on the page index): the key has 2 k bits,
// the flank k - 15 bases (2 k - 30 bits).  k = 31 gives flank_of_key.
constexpr int MINI_K_MIN = MINI_M + 2;           // 17: at least three m-mers per k-mer (below that the flat table serves)
__host__ __device__ __forceinline__ uint32_t flank_of_key_k(uint64_t key, uint32_t off, int k)
{
    const uint64_t M = (1ull << (2 * k)) - 1;
    const uint64_t r = off ? (((key >> (2 * off)) | (key << (2 * k - 2 * off))) & M) : key;
    return (uint32_t)(r >> 30);
}

// minimizer (the m-mer itself) of a k-mer and its LEFTMOST offset inside the k-mer
__host__ __device__ inline uint32_t mini_of_key(uint64_t key, int k, uint32_t *offset)
{
    const int w = k - MINI_M + 1;
    uint32_t best = 0, bo = 0, bx = 0;
    for (int i = 0; i < w; i++) {
        const uint32_t x = (uint32_t)(key >> (2 * i)) & M30;
        const uint32_t h = mmkey(x) & KEY_MASK;
        if (i == 0 || h < best) { best = h; bo = (uint32_t)i; bx = x; }
    }
    *offset = bo;
    return bx;
}

}  // namespace ss

namespace ss { namespace dev {

constexpr int SCAN_THREADS = 256;
constexpr int PPT = 16;                        // k-mer start positions per thread per tile
constexpr int TILE = SCAN_THREADS * PPT;       // bytes of the base stream per tile

// ---------------------------------------------------------------------------------------------
// 16 ASCII bases (4 dwords) -> 32 bits of 2-bit codes (base i at bits 2i) + 16 invalid flags.
// SWAR + byte permute + byte dot products: no per-byte loop, no LDS lookup table, no multiplies.
// ---------------------------------------------------------------------------------------------
// Per dword: code bytes c = (ascii >> 1) & 3; a byte permute looks up the upper-case letter each
// code stands for ('A','C','T','G'), so one xor with the case-folded input leaves a non-zero byte
// exactly where the base is not ACGT/acgt; v_dot4_u32_u8 packs the four 2-bit codes (weights
// 1,4,16,64) and the four flags (weights 1,2,4,8 / 16,..,128 on 0x80 bytes) in one instruction each.
__device__ __forceinline__ void encode16(const uint32_t w[4], uint32_t &code, uint32_t &inv)
{
    uint32_t cb[4], fl[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t c = (w[d] >> 1) & 0x03030303u;
        const uint32_t letter = __builtin_amdgcn_perm(0u, 0x47544341u, c);          // code -> 'A' 'C' 'T' 'G'
        const uint32_t diff = (w[d] & 0xDFDFDFDFu) ^ letter;
        fl[d] = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u;       // 0x80 where the byte is not a base
        cb[d] = __builtin_amdgcn_udot4(c, 0x40100401u, 0u, false);
    }
    code = cb[0] | (cb[1] << 8) | (cb[2] << 16) | (cb[3] << 24);
    const uint32_t lo = __builtin_amdgcn_udot4(fl[1], 0x80402010u, __builtin_amdgcn_udot4(fl[0], 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(fl[3], 0x80402010u, __builtin_amdgcn_udot4(fl[2], 0x08040201u, 0u, false), false);
    inv = (lo | (hi << 8)) >> 7;                                                    // sums are 128 x the flag masks
}

// 16 bytes at `off` of the base stream; bytes at or beyond n read as '\n'.
template <bool ALIGNED>
__device__ __forceinline__ void load16(const uint8_t *__restrict__ bases, uint64_t off, uint64_t n,
                                       uint32_t w[4])
{
    if (ALIGNED && off + 16 <= n) {
        // non-temporal: the base stream is read exactly once, and 3 GB of it must not push the 4 MB Bloom filter (and the
        // directory / bucket sectors, which do get re-used) out of L2: 3.45 vs 3.50 ms for 20 M reads
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(bases + off));
        const uint4 v = make_uint4(q.x, q.y, q.z, q.w);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                uint64_t p = off + d * 4 + b;
                uint32_t c = (p < n) ? bases[p] : 0x0Au;
                x |= c << (8 * b);
            }
            w[d] = x;
        }
    }
}


}}  // namespace ss::dev
