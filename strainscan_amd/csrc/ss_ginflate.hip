// ss_ginflate.hip -- one gzip member inflated by thousands of waves on the GPU.
//
// The reference pipes `zcat` into jellyfish (library/identify.py:81-84).  ss_pgz.hip restates the two-pass scheme of
// Kerbiriou & Chikhi (pugz, 2019) on host threads; on a box whose cgroup grants 16 CPUs that is 11 M reads/s, twenty
// times slower than plain text reaches HBM.  The same scheme on the device:
//   A  sync     the deflate data is cut into chunks of SS_GZ_CHUNK bytes.  Every wave searches its chunk for a block
//               entry: the 64 lanes test 64 consecutive bit positions at a time with a register-only header check
//               (non-final dynamic block, complete code-length code, valid run lengths, complete literal/length
//               and distance codes, end-of-block present); a candidate is confirmed by decoding that whole block
//               (count only) and finding a plausible block header behind it.
//   B  inflate  every wave decodes from its entry to the next chunk's entry, WAVE-UNIFORMLY: all lanes follow the same
//               bit stream (Huffman tables in LDS, compressed bytes staged through LDS), lane 0 stores literals, the
//               64 lanes copy matches together.  What lies in the 32 KB in front of the chunk is unknown: the output
//               is 16-bit symbols, a byte or "byte w of the window" (a copy of a copy keeps the index).  The last 4096
//               symbols are mirrored in an LDS ring so that a match never waits for the wave's own global stores.
//   C  windows  chunk after chunk, the last 32 KB of a chunk are resolved with its window and become the next window;
//   D  bytes    all symbols -> bytes in parallel; CRC-32 of the text by segments (combined on the host).
// Accepted only if every chunk ended exactly on the next one's entry, the stream ended at the member's trailer and
// CRC-32 and ISIZE match; anything else (several members, a chunk that expands more than SS_GZ_RATIO times, a damaged
// file) returns "not handled" and the caller inflates on the host (ss_pgz.hip, libdeflate, zlib), so a wrong text cannot
// get through.
#include "ss_common.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <vector>

namespace {

constexpr uint32_t WSIZE = 32768;
constexpr uint16_t UNRES = 0x8000;          // symbol = UNRES | index into the 32 KB in front of the chunk
#ifndef SS_GZ_RING
#define SS_GZ_RING 2048
#endif
#ifndef SS_GZ_LITBITS
#define SS_GZ_LITBITS 11
#endif
constexpr int RING = SS_GZ_RING;            // most recent symbols of a wave, in LDS
constexpr int STAGE = 2048;                 // compressed bytes staged in LDS at a time (two halves of 1 KB)

__constant__ uint16_t c_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// ---- the per-lane header check ------------------------------------------------------------------------------------
// Could a NON-FINAL DYNAMIC block header start at bit `p`?  One unaligned 16-byte load per lane (the header bits and the
// code-length code's lengths: 17 + at most 57 bits), registers only: block type, HLIT / HDIST in range, and the
// code-length code complete (zlib insists on that).  About one position in 2000 passes; those are then parsed and
// decoded by the whole wave (read_dynamic + inflate_block), which is the real test.
__device__ __forceinline__ bool header_prefilter(const uint8_t *in, uint64_t p)
{
    uint64_t lo, hi;
    const uint8_t *q = in + (p >> 3);
    __builtin_memcpy(&lo, q, 8);
    __builtin_memcpy(&hi, q + 8, 8);
    const int sh = (int)(p & 7);
    if (sh) { lo = (lo >> sh) | (hi << (64 - sh)); hi >>= sh; }
    if ((lo & 7u) != 4u) return false;                           // BFINAL = 0, BTYPE = 2 (bits: 0, then 0 1 LSB first = value 2)
    const uint32_t hlit = (uint32_t)(lo >> 3) & 31u, hdist = (uint32_t)(lo >> 8) & 31u, hclen = ((uint32_t)(lo >> 13) & 15u) + 4u;
    if (hlit > 29u || hdist > 29u) return false;
    // 3-bit lengths from bit 17 on: Kraft sum over the non-zero ones must be exactly 2^7
    uint64_t w = (lo >> 17) | (hi << 47);                        // bits 17.. of the header (at least 57 + 17 <= 128 - 7)
    uint32_t kraft = 0;
    for (uint32_t i = 0; i < hclen; i++) {
        const uint32_t l = (uint32_t)w & 7u;
        w >>= 3;
        if (i == 15) w |= (hi >> (3 * 16 + 17 - 64)) << 0 & 0;   // (no-op: 19 x 3 = 57 bits fit in w's 64)
        kraft += l ? (128u >> l) : 0u;
    }
    return kraft == 128u;
}

// ---- wave-uniform decoder: tables and staging in LDS -------------------------------------------------------------
template <int PB, int MAXSYM>
struct LHuff {
    uint16_t tent[1 << PB];          // symbol | code length << 9 for every PB-bit pattern; 0 = code longer than PB bits
    uint16_t count[16], sorted[MAXSYM];
    int maxlen;
};
struct WaveState {
    LHuff<SS_GZ_LITBITS, 288> lit;
    LHuff<8, 32> dist;
    LHuff<7, 19> clc;
    uint8_t lens[320];
    uint16_t len_base[32], dist_base[32];      // the length / distance tables, copied from constant memory once per wave:
    uint8_t len_extra[32], dist_extra[32];     // a constant-memory load with a data-dependent index costs a memory round trip
    uint16_t ring[RING];
    alignas(16) uint32_t stage[STAGE / 4];
};

// bit reader over the LDS stage (512 dwords = two halves of 1 KB): the wave refills a half with 16-byte loads when the
// reader has left it.  Everything here is called wave-uniformly; LDS operations of one wave execute in order, so no
// barrier is needed between the lanes' stores and the (uniform) loads that follow.
struct SBits {
    const uint32_t *in;      // global, dword aligned, padded with zeros behind the data
    uint64_t n;              // bytes of data
    uint64_t wpos;           // next dword to take (absolute index)
    uint64_t staged_to;      // dword index up to which the stage holds data
    uint64_t buf;
    int cnt;
};
__device__ __forceinline__ void sb_fill_half(WaveState &S, const SBits &b, uint64_t w0)
{
    // dwords [w0, w0 + 256) -> stage; w0 is a multiple of 256
    const int lane = threadIdx.x & 63;
    const uint4 v = *reinterpret_cast<const uint4 *>(b.in + w0 + (uint64_t)lane * 4);
    *reinterpret_cast<uint4 *>(&S.stage[(w0 & 511) + lane * 4]) = v;
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t sb_word(WaveState &S, SBits &b)
{
    // every lane reads the same word: telling the compiler so (readfirstlane) moves the whole bit reader, the code
    // lookups and the block logic onto the scalar unit -- one instruction per cycle beside the other waves' vector work
    const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.stage[b.wpos & 511]);
    b.wpos++;
    if ((b.wpos & 255) == 0) {           // the half just left is refilled with the data behind the other one
        sb_fill_half(S, b, b.staged_to);
        b.staged_to += 256;
    }
    return v;
}
__device__ __forceinline__ void sb_init(WaveState &S, SBits &b, const uint8_t *p, uint64_t n, uint64_t bitpos)
{
    b.in = reinterpret_cast<const uint32_t *>(p); b.n = n;
    b.wpos = bitpos >> 5;
    const uint64_t h0 = b.wpos & ~255ull;
    __builtin_amdgcn_wave_barrier();
    sb_fill_half(S, b, h0);
    sb_fill_half(S, b, h0 + 256);
    b.staged_to = h0 + 512;
    const int skip = (int)(bitpos & 31);
    b.buf = (uint64_t)sb_word(S, b) >> skip;
    b.cnt = 32 - skip;
}
// the same when the stage may still hold the right 2 KB (the sync search tries many positions a few bits apart, and a
// wrong one is usually rejected within a few hundred bits): no global loads then
__device__ __forceinline__ void sb_seek(WaveState &S, SBits &b, const uint8_t *p, uint64_t n, uint64_t bitpos, uint64_t &stage_base)
{
    const uint64_t w = bitpos >> 5;
    if (stage_base != ~0ull && b.staged_to == stage_base + 512 && w >= stage_base && w < stage_base + 200) {
        b.wpos = w;
        const int skip = (int)(bitpos & 31);
        b.buf = (uint64_t)sb_word(S, b) >> skip;
        b.cnt = 32 - skip;
        return;
    }
    sb_init(S, b, p, n, bitpos);
    stage_base = w & ~255ull;
}
// more than 32 bits in the buffer
__device__ __forceinline__ void sb_refill(WaveState &S, SBits &b)
{
    if (b.cnt <= 32) {
        b.buf |= (uint64_t)sb_word(S, b) << b.cnt;
        b.cnt += 32;
    }
}
__device__ __forceinline__ uint32_t sb_peek(SBits &b, int k) { return (uint32_t)(b.buf & ((1ull << k) - 1)); }
__device__ __forceinline__ void sb_drop(SBits &b, int k) { b.buf >>= k; b.cnt -= k; }
__device__ __forceinline__ uint32_t sb_get(WaveState &S, SBits &b, int k)      // k <= 32
{
    if (b.cnt < k) sb_refill(S, b);
    const uint32_t v = sb_peek(b, k);
    sb_drop(b, k);
    return v;
}
__device__ __forceinline__ uint64_t sb_bitpos(const SBits &b) { return b.wpos * 32 - (uint64_t)b.cnt; }
__device__ __forceinline__ bool sb_past_end(const SBits &b) { return sb_bitpos(b) > b.n * 8; }

// canonical code from lengths; lane 0 builds, the wave waits.  0 complete, 1 incomplete, -1 over-subscribed
template <int PB, int MAXSYM>
__device__ int huff_build(LHuff<PB, MAXSYM> &h, const uint8_t *lens, int n)
{
    __shared__ int s_ret;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        for (int i = 0; i < 16; i++) h.count[i] = 0;
        for (int i = 0; i < n; i++) h.count[lens[i]]++;
        int maxlen = 15;
        while (maxlen > 0 && h.count[maxlen] == 0) maxlen--;
        h.maxlen = maxlen;
        int left = 1, ret = 0;
        for (int l = 1; l <= 15; l++) { left <<= 1; left -= h.count[l]; if (left < 0) { ret = -1; break; } }
        if (ret == 0) {
            uint16_t offs[16];
            offs[1] = 0;
            for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
            for (int i = 0; i < n; i++) if (lens[i]) h.sorted[offs[lens[i]]++] = (uint16_t)i;
            ret = left > 0 ? 1 : 0;
        }
        s_ret = ret;
    }
    __syncthreads();
    const int ret = __builtin_amdgcn_readfirstlane(s_ret);      // uniform: a value read from LDS is "divergent" to the compiler,
    if (ret < 0) return ret;                                    // and one divergent branch moves the whole decoder state to VGPRs
    // table: all lanes clear, then lane-parallel fill per code
    for (int e = threadIdx.x & 63; e < (1 << PB); e += 64) h.tent[e] = 0;
    __syncthreads();
    // codes in canonical order: code value of sorted[idx] = first_code[len] + (idx - offs[len])
    {
        int idx0 = 0;
        uint32_t code = 0;
        for (int l = 1; l <= PB; l++) {
            const int c = __builtin_amdgcn_readfirstlane((int)h.count[l]);
            for (int k = threadIdx.x & 63; k < c; k += 64) {
                const uint32_t cd = code + (uint32_t)k;
                uint32_t r = __brev(cd) >> (32 - l);
                const uint16_t sym = h.sorted[idx0 + k];
                for (uint32_t e = r; e < (1u << PB); e += 1u << l) h.tent[e] = (uint16_t)(sym | (l << 9));
            }
            idx0 += c;
            code = (code + (uint32_t)c) << 1;
        }
    }
    __syncthreads();
    return ret;
}
template <int PB, int MAXSYM>
__device__ __forceinline__ int huff_decode(const LHuff<PB, MAXSYM> &h, SBits &b)     // needs >= 15 bits buffered
{
    const uint32_t v = sb_peek(b, 15);
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)h.tent[v & ((1u << PB) - 1)]);
    if (e) { sb_drop(b, (int)(e >> 9)); return (int)(e & 511u); }
    // codes longer than the table's PB bits: bit by bit (rare)
    int code = 0, first = 0, index = 0;
#pragma nounroll
    for (int len = 1; len <= 15; len++) {
        code |= (int)((v >> (len - 1)) & 1u);
        const int c = __builtin_amdgcn_readfirstlane((int)h.count[len]);
        if (code - c < first) { sb_drop(b, len); return __builtin_amdgcn_readfirstlane((int)h.sorted[index + (code - first)]); }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

__device__ bool read_dynamic(WaveState &S, SBits &b, bool &dist_usable)
{
    const int hlit = (int)sb_get(S, b, 5) + 257, hdist = (int)sb_get(S, b, 5) + 1, hclen = (int)sb_get(S, b, 4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    __syncthreads();
    if ((threadIdx.x & 63) < 19) S.lens[threadIdx.x & 63] = 0;
    __syncthreads();
    for (int i = 0; i < hclen; i++) {
        const uint32_t v = sb_get(S, b, 3);
        if ((threadIdx.x & 63) == 0) S.lens[c_cl_order[i]] = (uint8_t)v;
    }
    if (__builtin_amdgcn_readfirstlane(huff_build(S.clc, S.lens, 19)) != 0) return false;   // (a call's result arrives in a VGPR)
    // the code lengths themselves: sequential; every lane decodes, lane 0 stores
    int i = 0;
    __shared__ uint8_t s_all[320];
    while (i < hlit + hdist) {
        if (b.cnt < 22) sb_refill(S, b);
        const int s = huff_decode(S.clc, b);
        if (s < 0 || sb_past_end(b)) return false;
        int rep = 1, val = s;
        if (s == 16) { if (i == 0) return false; val = __builtin_amdgcn_readfirstlane((int)s_all[i - 1]); rep = 3 + (int)sb_get(S, b, 2); }
        else if (s == 17) { val = 0; rep = 3 + (int)sb_get(S, b, 3); }
        else if (s == 18) { val = 0; rep = 11 + (int)sb_get(S, b, 7); }
        if (i + rep > hlit + hdist) return false;
        __syncthreads();
        for (int r = threadIdx.x & 63; r < rep; r += 64) s_all[i + r] = (uint8_t)val;
        __syncthreads();
        i += rep;
    }
    if (__builtin_amdgcn_readfirstlane((int)s_all[256]) == 0) return false;
    for (int k = threadIdx.x & 63; k < hlit + hdist; k += 64) S.lens[k] = s_all[k];
    __syncthreads();
    const int rl = __builtin_amdgcn_readfirstlane(huff_build(S.lit, S.lens, hlit));
    if (rl < 0 || (rl > 0 && __builtin_amdgcn_readfirstlane(S.lit.maxlen) != 1)) return false;
    const int rd = __builtin_amdgcn_readfirstlane(huff_build(S.dist, S.lens + hlit, hdist));
    const int dmax = __builtin_amdgcn_readfirstlane(S.dist.maxlen);
    if (rd < 0 || (rd > 0 && dmax > 1)) return false;
    dist_usable = dmax > 0;
    return !sb_past_end(b);
}

__device__ void fixed_codes(WaveState &S)
{
    __syncthreads();
    for (int i = threadIdx.x & 63; i < 288; i += 64) S.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
    __syncthreads();
    huff_build(S.lit, S.lens, 288);
    __syncthreads();
    for (int i = threadIdx.x & 63; i < 30; i += 64) S.lens[i] = 5;
    __syncthreads();
    huff_build(S.dist, S.lens, 30);
}

// Output of a wave: symbols go to the LDS ring first; whenever 1024 new ones have gathered, the 64 lanes write them to
// global memory together (128 contiguous bytes per store instruction).  A match reads its source from the ring when it
// is at most RING_REACH symbols back (always true for what is not flushed yet) and from global memory otherwise --
// data flushed at least two flushes ago; every flush first waits for the stores of the one before.
constexpr uint32_t RING_REACH = RING - 264;
struct OutState {
    uint16_t *out;           // null: count only
    uint64_t cap, n, flushed;
};
__device__ __forceinline__ void out_flush(WaveState &S, OutState &o, uint64_t upto)
{
    const int lane = threadIdx.x & 63;
    __threadfence_block();
    for (uint64_t i = o.flushed + (uint64_t)lane; i < upto; i += 64) o.out[i] = S.ring[i & (RING - 1)];
    o.flushed = upto;
}

__device__ __forceinline__ void wave_tables(WaveState &S)
{
    const int lane = threadIdx.x & 63;
    if (lane < 29) { S.len_base[lane] = c_len_base[lane]; S.len_extra[lane] = c_len_extra[lane]; }
    if (lane < 30) { S.dist_base[lane] = c_dist_base[lane]; S.dist_extra[lane] = c_dist_extra[lane]; }
    __syncthreads();
}

// One block from the current position.  0 = block done, 1 = final block done, < 0 = error.
__device__ int inflate_block(WaveState &S, SBits &b, OutState &o, bool known_window, uint64_t probe_symbols = ~0ull)
{
    const int lane = threadIdx.x & 63;
    const bool store = o.out != nullptr;
    const uint32_t bfinal = sb_get(S, b, 1), btype = sb_get(S, b, 2);
    if (btype == 3) return -1;
    uint64_t n = o.n;
#define GI_RET(v) do { o.n = n; return (v); } while (0)
    if (btype == 0) {
        sb_drop(b, b.cnt & 7);
        const uint32_t len = sb_get(S, b, 16), nlen = sb_get(S, b, 16);
        if (sb_past_end(b) || (len ^ 0xFFFFu) != nlen) GI_RET(-2);
        if (store && n + len > o.cap) GI_RET(-9);
        for (uint32_t i = 0; i < len; i++) {
            const uint32_t v = sb_get(S, b, 8);
            if (store) {
                if (lane == 0) S.ring[n & (RING - 1)] = (uint16_t)v;
                if (n + 1 - o.flushed >= 1024) { o.n = n + 1; out_flush(S, o, n + 1); }
            }
            n++;
        }
        GI_RET(sb_past_end(b) ? -3 : (int)bfinal);
    }
    bool dist_usable = true;
    if (btype == 1) fixed_codes(S);
    else if (!read_dynamic(S, b, dist_usable)) GI_RET(-4);
    const uint64_t probe_end = probe_symbols == ~0ull ? ~0ull : n + probe_symbols;
    for (;;) {
        if (store) {
            if (n + 264 > o.cap) GI_RET(-9);
            if (n - o.flushed >= 1024) out_flush(S, o, n);
        }
        if (n >= probe_end) GI_RET(2);                     // sync search: the header was valid and this many symbols decoded
        sb_refill(S, b);                                   // > 32 bits: two literal/length codes
        int s = huff_decode(S.lit, b);
        if (s >= 0 && s < 256) {
            if (store && lane == 0) S.ring[n & (RING - 1)] = (uint16_t)s;
            n++;
            s = huff_decode(S.lit, b);
            if (s >= 0 && s < 256) {
                if (store && lane == 0) S.ring[n & (RING - 1)] = (uint16_t)s;
                n++;
                if (sb_past_end(b)) GI_RET(-5);
                continue;
            }
        }
        if (s < 0 || sb_past_end(b)) GI_RET(-5);
        if (s == 256) GI_RET((int)bfinal);
        if (s > 285) GI_RET(-6);
        sb_refill(S, b);                                   // length extra 5 + distance 15 + distance extra 13 = 33 bits
        const int li = s - 257;
        const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.len_base[li]) +
                             sb_get(S, b, __builtin_amdgcn_readfirstlane((int)S.len_extra[li]));
        if (!dist_usable) GI_RET(-7);
        const int ds = huff_decode(S.dist, b);
        if (ds < 0 || ds > 29) GI_RET(-7);
        const uint32_t dist = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.dist_base[ds]) +
                              sb_get(S, b, __builtin_amdgcn_readfirstlane((int)S.dist_extra[ds]));
        if (sb_past_end(b)) GI_RET(-5);
        if (known_window && dist > n) GI_RET(-8);
        if (store) {
            const uint32_t n32 = (uint32_t)n;                  // a chunk's output is far below 2^31 symbols
            for (uint32_t base = 0; base < len; base += 64) {
                const uint32_t i = base + (uint32_t)lane;
                if (i < len) {
                    const uint32_t k = dist >= len ? i : i % dist;          // a match that overlaps itself repeats with period dist
                    const int32_t sp = (int32_t)(n32 - dist + k);
                    uint16_t v;
                    if (sp < 0) v = (uint16_t)(UNRES | (uint32_t)((int32_t)WSIZE + sp));
                    else if (n32 - (uint32_t)sp <= RING_REACH) v = S.ring[sp & (RING - 1)];
                    else v = o.out[sp];
                    // sources lie in front of n, destinations behind it, and a ring slot is never both within one
                    // match (a source that far back is read from global memory): no hazard between the pieces
                    S.ring[(n32 + i) & (RING - 1)] = v;
                }
            }
        }
        n += len;
    }
#undef GI_RET
}

__device__ unsigned g_sync_tries;      // candidates that passed the header check and were decoded (trace)

// ---- kernels ------------------------------------------------------------------------------------------------------
// A: entry point of every chunk (chunk 0: the start of the deflate data)
__global__ __launch_bounds__(64) void sync_kernel(const uint8_t *in, uint64_t in_n, uint64_t data_off, uint64_t chunk_bytes,
                                                  uint32_t n_chunks, uint64_t *entry /* bit position or ~0 */, uint64_t probe)
{
    __shared__ WaveState S;
    const uint32_t c = blockIdx.x;
    const int lane = threadIdx.x & 63;
    if (c == 0) { if (lane == 0) entry[0] = data_off * 8; return; }
    wave_tables(S);
    const uint64_t lo = (data_off + (uint64_t)c * chunk_bytes) * 8, hi = min(in_n * 8, lo + chunk_bytes * 8);
    uint64_t found = ~0ull, stage_base = ~0ull;
    SBits b;
    b.staged_to = 0;
    for (uint64_t p0 = lo; p0 < hi && found == ~0ull; p0 += 64) {
        const uint64_t p = p0 + lane;
        const bool ok = p < hi && header_prefilter(in, p);
        uint64_t m = __ballot(ok);
        while (m && found == ~0ull) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint64_t cand = p0 + (uint64_t)l;
            // confirm with the whole wave: the header parses, the block decodes to its end, and what follows looks like
            // the start of a block
            sb_seek(S, b, in, in_n, cand, stage_base);
            OutState o{nullptr, 0, 0, 0};
            // a complete, valid header (zlib's rules) and SS_GZ_PROBE symbols that decode: a position inside a block
            // passes this with negligible probability, and if one ever does, the chunk will not end on the next entry and
            // the file goes to the host inflater
            const int r = inflate_block(S, b, o, false, probe);
            if (lane == 0) atomicAdd(&g_sync_tries, 1u);
            if (r == 2 || (r == 0 && o.n > 0)) found = cand;
        }
    }
    if (lane == 0) entry[c] = found;
}

// B: every listed chunk from its entry to the next entry (or the end of the stream)
__global__ __launch_bounds__(64) void inflate_kernel(const uint8_t *in, uint64_t in_n, const uint64_t *start, const uint64_t *stop,
                                                     uint32_t n_chunks, uint16_t *sym, const uint64_t *sym_off, const uint64_t *sym_cap,
                                                     uint64_t *out_len, uint64_t *end_bit, int *status)
{
    __shared__ WaveState S;
    const uint32_t c = blockIdx.x;
    if (c >= n_chunks) return;
    const int lane = threadIdx.x & 63;
    wave_tables(S);
    SBits b;
    sb_init(S, b, in, in_n, start[c]);
    OutState o{sym ? sym + sym_off[c] : nullptr, sym_cap[c], 0, 0};
    const uint64_t stop_at = stop[c];
    int st = 0;
    for (;;) {
        const int r = inflate_block(S, b, o, c == 0);
        if (r < 0) { st = r; break; }
        const uint64_t pos = sb_bitpos(b);
        if (r == 1) { st = (stop_at == ~0ull) ? 1 : -20; break; }      // the final block ends the LAST chunk only
        if (pos == stop_at) { st = 0; break; }
        if (pos > stop_at) { st = -21; break; }
    }
    if (st >= 0 && o.out) out_flush(S, o, o.n);
    const uint64_t n = o.n;
    if (lane == 0) { out_len[c] = n; end_bit[c] = sb_bitpos(b); status[c] = st; }
}

// C: windows.  The 32 KB in front of chunk c + 1 are the last 32 KB of chunk c's output, in which a symbol may still
//    point into chunk c's own window, and so on down the chain.  Instead of walking the chain chunk after chunk
//    (thousands of dependent steps), the tails are treated as maps "window of c -> window of c + 1" (an entry is a byte
//    or an index into the previous window) and composed by doubling: after round r every map reaches 2^r chunks back;
//    ~log2(n_chunks) fully parallel rounds, in practice two or three until no index is left.
__global__ __launch_bounds__(256) void tails_kernel(const uint16_t *sym, const uint64_t *sym_off, const uint64_t *out_len,
                                                    uint32_t n_chunks, uint16_t *map)
{
    const uint32_t c = blockIdx.y;                            // map[c] : window of chunk c -> window of chunk c + 1
    const uint64_t L = out_len[c];
    const uint16_t *sy = sym + sym_off[c];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < WSIZE; i += gridDim.x * 256) {
        uint16_t e;
        if (L >= WSIZE - i) e = sy[L - (WSIZE - i)];
        else e = (uint16_t)(UNRES | (i + (uint32_t)L));       // the chunk was shorter than the window: its own window shifts in
        map[(uint64_t)c * WSIZE + i] = e;
    }
}
__global__ __launch_bounds__(256) void compose_kernel(const uint16_t *in, uint16_t *out, uint32_t n_chunks, uint32_t span, uint32_t *more)
{
    const uint32_t c = blockIdx.y;
    bool any = false;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < WSIZE; i += gridDim.x * 256) {
        uint16_t e = in[(uint64_t)c * WSIZE + i];
        if ((e & UNRES) && c >= span) {
            e = in[(uint64_t)(c - span) * WSIZE + (e & (WSIZE - 1))];
            any = any || ((e & UNRES) && c >= 2 * span);
        }
        out[(uint64_t)c * WSIZE + i] = e;
    }
    if (__ballot(any) && (threadIdx.x & 63) == 0) atomicOr(more, 1u);
}
// win[c + 1] = bytes of map[c]; win[0] = nothing (chunk 0 has no unknown window)
__global__ __launch_bounds__(256) void windows_kernel(const uint16_t *map, uint32_t n_chunks, uint8_t *win)
{
    const uint32_t c = blockIdx.y;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < WSIZE; i += gridDim.x * 256)
        win[(uint64_t)c * WSIZE + i] = c == 0 ? (uint8_t)0 : (uint8_t)map[(uint64_t)(c - 1) * WSIZE + i];
}

// D: symbols -> bytes
__global__ __launch_bounds__(256) void bytes_kernel(const uint16_t *sym, const uint64_t *sym_off, const uint64_t *out_len,
                                                    const uint64_t *text_off, const uint8_t *win, uint8_t *text)
{
    const uint32_t c = blockIdx.y;
    const uint64_t L = out_len[c];
    const uint16_t *sy = sym + sym_off[c];
    const uint8_t *w = win + (uint64_t)c * WSIZE;
    uint8_t *dst = text + text_off[c];
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < L; i += (uint64_t)gridDim.x * 256) {
        const uint16_t s = sy[i];
        dst[i] = (s & UNRES) ? w[s & (WSIZE - 1)] : (uint8_t)s;
    }
}

// CRC-32 (zlib's polynomial) of segment s of the text, one lane per segment, byte-wise table in LDS
__global__ __launch_bounds__(64) void crc_kernel(const uint8_t *text, uint64_t n, uint64_t seg, const uint32_t *table, uint32_t *crc)
{
    __shared__ uint32_t tab[256];
    for (int i = threadIdx.x; i < 256; i += 64) tab[i] = table[i];
    __syncthreads();
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t a = s * seg;
    if (a >= n) return;
    const uint64_t e = min(n, a + seg);
    uint32_t k = 0xFFFFFFFFu;
    uint64_t i = a;
    for (; i + 4 <= e; i += 4) {                                 // seg is a multiple of 4 and the text is 4-byte aligned
        const uint32_t w = *reinterpret_cast<const uint32_t *>(text + i);
        k = tab[(k ^ w) & 0xFFu] ^ (k >> 8);
        k = tab[(k ^ (w >> 8)) & 0xFFu] ^ (k >> 8);
        k = tab[(k ^ (w >> 16)) & 0xFFu] ^ (k >> 8);
        k = tab[(k ^ (w >> 24)) & 0xFFu] ^ (k >> 8);
    }
    for (; i < e; i++) k = tab[(k ^ text[i]) & 0xFFu] ^ (k >> 8);
    crc[s] = k ^ 0xFFFFFFFFu;
}

// GF(2) operator "append seg zero bytes" for CRC-32 (zlib's crc32_combine builds it anew for every call; here all segments
// but the last have the same length, so it is built once)
uint32_t gf2_times(const uint32_t *mat, uint32_t vec)
{
    uint32_t sum = 0;
    for (int i = 0; vec; vec >>= 1, i++) if (vec & 1u) sum ^= mat[i];
    return sum;
}
void gf2_square(uint32_t *sq, const uint32_t *mat) { for (int i = 0; i < 32; i++) sq[i] = gf2_times(mat, mat[i]); }
void crc_zero_operator(uint32_t *op /*[32]*/, int log2_bytes)
{
    uint32_t a[32], b[32];
    a[0] = 0xEDB88320u;                                   // one zero BIT
    for (int i = 1; i < 32; i++) a[i] = 1u << (i - 1);
    gf2_square(b, a); gf2_square(a, b); gf2_square(b, a);  // 2, 4, 8 bits = one byte (in b)
    uint32_t *cur = b, *oth = a;
    for (int k = 0; k < log2_bytes; k++) { gf2_square(oth, cur); std::swap(cur, oth); }
    for (int i = 0; i < 32; i++) op[i] = cur[i];
}

uint64_t gzip_header_len(const uint8_t *p, uint64_t n)
{
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return 0;
    const uint8_t flg = p[3];
    if (flg & 0xE0) return 0;
    uint64_t pos = 10;
    if (flg & 4) { if (pos + 2 > n) return 0; pos += 2 + ((uint64_t)p[pos] | (uint64_t)p[pos + 1] << 8); }
    for (int f = 0; f < 2; f++)
        if (flg & (f == 0 ? 8 : 16)) {
            while (pos < n && p[pos]) pos++;
            pos++;
        }
    if (flg & 2) pos += 2;
    return pos + 8 < n ? pos : 0;
}

}  // namespace

namespace ss {

// The file image `in` (host) inflated on the device.  true: *text_dev (hipMalloc) holds *len bytes, verified against the
// trailer.  false: not handled here (the caller inflates on the host).
static std::atomic<uint64_t> g_handled{0}, g_declined{0};

bool gpu_gunzip(const uint8_t *in, uint64_t in_n, char **text_dev, uint64_t *len)
{
    static const bool trace = getenv("SS_INGEST_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        hipDeviceSynchronize();
        fprintf(stderr, "[ginflate] %-18s at %.4f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    auto no = [&](const char *why, long long a = 0) {
        if (trace) fprintf(stderr, "[ginflate] not handled: %s (%lld)\n", why, a);
        g_declined++;
        return false;
    };
    const uint64_t data_off = gzip_header_len(in, in_n);
    if (!data_off) return no("header");
    const uint8_t *tr = in + in_n - 8;
    const uint32_t want_crc = (uint32_t)tr[0] | (uint32_t)tr[1] << 8 | (uint32_t)tr[2] << 16 | (uint32_t)tr[3] << 24;
    const uint32_t want_isize = (uint32_t)tr[4] | (uint32_t)tr[5] << 8 | (uint32_t)tr[6] << 16 | (uint32_t)tr[7] << 24;
    uint64_t chunk_bytes = 32 << 10, ratio = 12;
    if (const char *e = getenv("SS_GZ_CHUNK")) chunk_bytes = std::max<uint64_t>(4096, (uint64_t)atoll(e));
    if (const char *e = getenv("SS_GZ_RATIO")) ratio = std::max<uint64_t>(2, (uint64_t)atoll(e));
    const uint64_t data_n = in_n - 8 - data_off;
    chunk_bytes = std::max<uint64_t>(chunk_bytes, data_n / 32768 + 1);                  // grid dimensions
    const uint32_t n_chunks0 = (uint32_t)std::max<uint64_t>(1, (data_n + chunk_bytes - 1) / chunk_bytes);

    uint8_t *d_in = nullptr, *d_win = nullptr, *d_text = nullptr;
    uint64_t *d_entry = nullptr, *d_meta = nullptr;
    uint16_t *d_sym = nullptr;
    uint32_t *d_crc = nullptr, *d_tab = nullptr;
    int *d_status = nullptr;
    auto cleanup = [&](bool keep_text) {
        hipFree(d_in); hipFree(d_win); hipFree(d_entry); hipFree(d_meta); hipFree(d_sym); hipFree(d_crc); hipFree(d_tab); hipFree(d_status);
        if (!keep_text) hipFree(d_text);
    };
#define GI(call) do { if ((call) != hipSuccess) { cleanup(false); return no(#call); } } while (0)
    GI(hipMalloc((void **)&d_in, in_n + 8192));                 // the stage is filled 1 KB at a time, up to 2 KB ahead
    GI(hipMemcpy(d_in, in, in_n, hipMemcpyHostToDevice));
    GI(hipMemset(d_in + in_n, 0, 8192));
    GI(hipMalloc((void **)&d_entry, (uint64_t)n_chunks0 * 8));
    lap("input on device");
    uint64_t probe = 512;
    if (const char *e = getenv("SS_GZ_PROBE")) probe = (uint64_t)atoll(e);
    hipLaunchKernelGGL(sync_kernel, dim3(n_chunks0), dim3(64), 0, 0, d_in, in_n - 8, data_off, chunk_bytes, n_chunks0, d_entry, probe);
    std::vector<uint64_t> entry(n_chunks0);
    GI(hipMemcpy(entry.data(), d_entry, (uint64_t)n_chunks0 * 8, hipMemcpyDeviceToHost));
    if (trace) {
        unsigned tries = 0;
        hipMemcpyFromSymbol(&tries, HIP_SYMBOL(g_sync_tries), 4);
        fprintf(stderr, "[ginflate] %u chunks, %u candidate blocks decoded\n", n_chunks0, tries);
        tries = 0;
        hipMemcpyToSymbol(HIP_SYMBOL(g_sync_tries), &tries, 4);
    }
    lap("sync");
    // chunks with an entry; a chunk without one belongs to its predecessor
    std::vector<uint64_t> start, stop, off, cap;
    for (uint32_t c = 0; c < n_chunks0; c++)
        if (entry[c] != ~0ull) start.push_back(entry[c]);
    const uint32_t nc = (uint32_t)start.size();
    uint64_t sym_total = 0;
    for (uint32_t c = 0; c < nc; c++) {
        stop.push_back(c + 1 < nc ? start[c + 1] : ~0ull);
        const uint64_t cbits = (c + 1 < nc ? start[c + 1] : (in_n - 8) * 8) - start[c];
        const uint64_t cp = (cbits / 8 + 1) * ratio + 4096;
        off.push_back(sym_total);
        cap.push_back(cp);
        sym_total += cp;
    }
    // meta: start, stop, off, cap, out_len, end_bit, text_off
    GI(hipMalloc((void **)&d_meta, (uint64_t)nc * 8 * 7));
    uint64_t *d_start = d_meta, *d_stop = d_meta + nc, *d_off = d_meta + 2ull * nc, *d_cap = d_meta + 3ull * nc, *d_len = d_meta + 4ull * nc,
             *d_end = d_meta + 5ull * nc, *d_toff = d_meta + 6ull * nc;
    GI(hipMemcpy(d_start, start.data(), (uint64_t)nc * 8, hipMemcpyHostToDevice));
    GI(hipMemcpy(d_stop, stop.data(), (uint64_t)nc * 8, hipMemcpyHostToDevice));
    GI(hipMemcpy(d_off, off.data(), (uint64_t)nc * 8, hipMemcpyHostToDevice));
    GI(hipMemcpy(d_cap, cap.data(), (uint64_t)nc * 8, hipMemcpyHostToDevice));
    {
        // symbols (2 B each, `ratio` per input byte), windows and maps (5 x 32 KB per chunk) and the text must fit
        size_t mem_free = 0, mem_total = 0;
        GI(hipMemGetInfo(&mem_free, &mem_total));
        const uint64_t need = sym_total * 2 + (uint64_t)nc * WSIZE * 5 + ((uint64_t)want_isize | (in_n > (1ull << 32) ? in_n * 4 : 0)) + (64 << 20);
        if (need > mem_free / 2) { cleanup(false); return no("device memory", (long long)(need >> 20)); }
    }
    GI(hipMalloc((void **)&d_sym, sym_total * 2));
    GI(hipMalloc((void **)&d_status, (uint64_t)nc * 4));
    lap("symbol buffers");
    if (getenv("SS_GZ_COUNTONLY")) {         // timing experiment: the decode without any output
        hipLaunchKernelGGL(inflate_kernel, dim3(nc), dim3(64), 0, 0, d_in, in_n - 8, d_start, d_stop, nc, (uint16_t *)nullptr, d_off, d_cap, d_len, d_end, d_status);
        lap("inflate (count only)");
    }
    hipLaunchKernelGGL(inflate_kernel, dim3(nc), dim3(64), 0, 0, d_in, in_n - 8, d_start, d_stop, nc, d_sym, d_off, d_cap, d_len, d_end, d_status);
    std::vector<int> status(nc);
    std::vector<uint64_t> out_len(nc), end_bit(nc), text_off(nc);
    GI(hipMemcpy(status.data(), d_status, (uint64_t)nc * 4, hipMemcpyDeviceToHost));
    GI(hipMemcpy(out_len.data(), d_len, (uint64_t)nc * 8, hipMemcpyDeviceToHost));
    GI(hipMemcpy(end_bit.data(), d_end, (uint64_t)nc * 8, hipMemcpyDeviceToHost));
    lap("inflate");
    uint64_t total = 0;
    for (uint32_t c = 0; c < nc; c++) {
        if (status[c] != (c + 1 == nc ? 1 : 0)) { cleanup(false); return no("chunk status", status[c] * 1000000ll + c); }
        text_off[c] = total;
        total += out_len[c];
    }
    // the stream must end where the trailer begins (after padding to a byte)
    if ((end_bit[nc - 1] + 7) / 8 != in_n - 8) { cleanup(false); return no("stream end", (long long)((end_bit[nc - 1] + 7) / 8)); }
    if ((uint32_t)total != want_isize) { cleanup(false); return no("isize"); }
    GI(hipMemcpy(d_toff, text_off.data(), (uint64_t)nc * 8, hipMemcpyHostToDevice));
    GI(hipMalloc((void **)&d_win, (uint64_t)nc * WSIZE));
    {
        uint16_t *d_map[2] = {nullptr, nullptr};
        uint32_t *d_more = nullptr;
        bool ok = hipMalloc((void **)&d_map[0], (uint64_t)nc * WSIZE * 2) == hipSuccess && hipMalloc((void **)&d_map[1], (uint64_t)nc * WSIZE * 2) == hipSuccess &&
                  hipMalloc((void **)&d_more, 4) == hipSuccess;
        int cur = 0;
        if (ok) {
            hipLaunchKernelGGL(tails_kernel, dim3(8, nc), dim3(256), 0, 0, d_sym, d_off, d_len, nc, d_map[0]);
            for (uint32_t span = 1; span < nc && ok; span *= 2) {
                uint32_t more = 0;
                ok = hipMemset(d_more, 0, 4) == hipSuccess;
                hipLaunchKernelGGL(compose_kernel, dim3(8, nc), dim3(256), 0, 0, d_map[cur], d_map[cur ^ 1], nc, span, d_more);
                cur ^= 1;
                ok = ok && hipMemcpy(&more, d_more, 4, hipMemcpyDeviceToHost) == hipSuccess;
                if (!more) break;
            }
            hipLaunchKernelGGL(windows_kernel, dim3(8, nc), dim3(256), 0, 0, d_map[cur], nc, d_win);
            ok = ok && hipDeviceSynchronize() == hipSuccess;
        }
        hipFree(d_map[0]); hipFree(d_map[1]); hipFree(d_more);
        if (!ok) { cleanup(false); return no("windows"); }
    }
    lap("windows");
    GI(hipMalloc((void **)&d_text, std::max<uint64_t>(total, 16) + 64));
    hipLaunchKernelGGL(bytes_kernel, dim3(64, nc), dim3(256), 0, 0, d_sym, d_off, d_len, d_toff, d_win, d_text);
    lap("bytes");
    // CRC-32 by segments of 16 KB, combined on the host with ONE precomputed operator
    constexpr int SEG_LOG2 = 14;
    const uint64_t seg = 1ull << SEG_LOG2, nseg = (total + seg - 1) / seg;
    std::vector<uint32_t> tab(256);
    for (uint32_t i = 0; i < 256; i++) { uint32_t k = i; for (int j = 0; j < 8; j++) k = (k & 1u) ? 0xEDB88320u ^ (k >> 1) : k >> 1; tab[i] = k; }
    GI(hipMalloc((void **)&d_tab, 1024));
    GI(hipMemcpy(d_tab, tab.data(), 1024, hipMemcpyHostToDevice));
    GI(hipMalloc((void **)&d_crc, std::max<uint64_t>(1, nseg) * 4));
    if (nseg) hipLaunchKernelGGL(crc_kernel, dim3((unsigned)((nseg + 63) / 64)), dim3(64), 0, 0, d_text, total, seg, d_tab, d_crc);
    std::vector<uint32_t> crcs(std::max<uint64_t>(1, nseg));
    if (nseg) GI(hipMemcpy(crcs.data(), d_crc, nseg * 4, hipMemcpyDeviceToHost));
    uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
    {
        uint32_t op[32];
        crc_zero_operator(op, SEG_LOG2);
        for (uint64_t s = 0; s < nseg; s++) {
            const uint64_t l = std::min<uint64_t>(seg, total - s * seg);
            crc = l == seg ? (gf2_times(op, crc) ^ crcs[s]) : (uint32_t)crc32_combine(crc, crcs[s], (z_off_t)l);
        }
    }
    lap("crc");
#undef GI
    if (crc != want_crc) { cleanup(false); return no("crc"); }
    cleanup(true);
    g_handled++;
    *text_dev = (char *)d_text;
    *len = total;
    return true;
}

}  // namespace ss

// members the device inflater has produced / has left to the host inflaters, in this process
extern "C" int ss_gz_gpu_counters(uint64_t *handled, uint64_t *declined)
{
    if (!handled || !declined) return SS_EINVAL;
    *handled = ss::g_handled.load();
    *declined = ss::g_declined.load();
    return SS_OK;
}

extern "C" int ss_gz_inflate_gpu(const char *path, char **text, uint64_t *len)
{
    if (!path || !text || !len) return SS_EINVAL;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return SS_EIO;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 20) { close(fd); return SS_EIO; }
    std::vector<uint8_t> buf((size_t)st.st_size);
    uint64_t got = 0;
    while (got < buf.size()) {
        const ssize_t r = pread(fd, buf.data() + got, buf.size() - got, (off_t)got);
        if (r <= 0) break;
        got += (uint64_t)r;
    }
    close(fd);
    if (got != buf.size()) return SS_EIO;
    char *d = nullptr;
    uint64_t n = 0;
    if (!ss::gpu_gunzip(buf.data(), buf.size(), &d, &n)) return SS_ERANGE;
    char *h = (char *)malloc(std::max<uint64_t>(n, 1));
    if (!h) { hipFree(d); return SS_ENOMEM; }
    const hipError_t e = n ? hipMemcpy(h, d, n, hipMemcpyDeviceToHost) : hipSuccess;
    hipFree(d);
    if (e != hipSuccess) { free(h); return SS_EHIP; }
    *text = h;
    *len = n;
    return SS_OK;
}
