// ss_ginflate.hip -- a gzip file inflated by thousands of waves on the GPU.
//
// The reference pipes `zcat` into jellyfish (library/identify.py:81-84).  ss_pgz.hip restates the two-pass scheme of
// Kerbiriou & Chikhi (pugz, 2019) on host threads; on a box whose cgroup grants 16 CPUs that is 11-14 M reads/s, twenty
// times slower than plain text reaches HBM.  The same scheme on the device:
//   A  sync     the deflate data is cut into chunks of SS_GZ_CHUNK bytes.  Every wave searches its chunk for a block
//               entry: the 64 lanes test 64 consecutive bit positions at a time with a register-only header check
//               (non-final dynamic block, complete code-length code); a survivor's header is parsed by the whole wave
//               (code lengths decoded lane-parallel, canonical codes built with ballots: the Kraft sums reject it
//               before a table is written) and 512 symbols are decoded.
//   A2 subsync  a block is entered every SS_GZ_SPLIT_KB (12 KB) of deflate data as well: Huffman-coded data synchronises
//               itself, so a wave that decodes from any bit inside a block with the block's tables -- all 64 bit offsets of
//               a window as hypotheses -- is left with ONE offset after a few windows: an item's first bit, an entry.
//   B  inflate  every wave decodes from its entry to the next chunk's entry.  The 64 lanes decode whole items (literal,
//               or length + distance with their extra bits) SPECULATIVELY at 64 consecutive bit positions of the
//               LDS-staged input; the wave follows the chain of item lengths (readlane), a prefix sum places the items.
//               What lies in the 32 KB in front of the chunk is unknown: the output is 16-bit symbols, a byte or "byte
//               w of the window" (a copy of a copy keeps the index).  The last 2048 symbols are mirrored in an LDS
//               ring; matches that reach further back read the wave's own output from global memory, all such
//               matches of a window together (one lane per symbol) and while the next window is decoded.
//   C  windows  the last 32 KB of every chunk as a map "my window -> the next chunk's window", composed in two levels;
//   D  bytes    all symbols -> bytes in parallel; CRC-32 of the text by segments (combined on the host).
// B-D run SEGMENT by segment (128 MB of deflate data): a segment's text is complete before the next one starts, so the
// window in front of a segment is the end of the text so far, and the scratch (a reusable arena, ~6 GB) does not grow
// with the file.
// Accepted only if every chunk ended exactly on the next one's entry, the stream ended at the file's last trailer and
// CRC-32 and ISIZE of EVERY member match (lanes joined with `cat` are found member by member: the chunk that meets a
// final block finds trailer and header behind it).  A wrong entry (a position inside a block that passed A) shows as the
// chunk before it running past it and is dropped.  A bgzip file (BGZF: every member names its size) needs no search: its
// members are the chunks.  Anything else (a chunk that expands more than 12 times, more than 8 MB of deflate data without a dynamic
// block's start -- stored blocks only --, a damaged file) returns "not handled" and the caller inflates on the host
// (ss_pgz.hip, libdeflate, zlib), so a wrong text cannot get through.
#include "ss_common.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <functional>
#include <future>

namespace {

constexpr uint32_t WSIZE = 32768;
constexpr uint16_t UNRES = 0x8000;          // symbol = UNRES | index into the 32 KB in front of the chunk
#ifndef SS_GZ_RING
#define SS_GZ_RING 2048
#endif
#ifndef SS_GZ_LITBITS
#define SS_GZ_LITBITS 9
#endif
constexpr int RING = SS_GZ_RING;            // most recent symbols of a wave, in LDS
#ifndef SS_GZ_STAGE
#define SS_GZ_STAGE 1024
#endif
constexpr int STAGE = SS_GZ_STAGE;          // compressed bytes staged in LDS at a time (two halves); LDS decides the waves per SIMD
constexpr uint32_t STAGE_DW = STAGE / 4, HALF_DW = STAGE_DW / 2, STAGE_MASK = STAGE_DW - 1;
static_assert(STAGE == 1024 || STAGE == 2048, "a half is filled by one 8- or 16-byte load per lane");

// length code li = symbol - 257 and distance code ds -> base value and extra bits (RFC 1951 3.2.5), computed: a table
// lookup with a data-dependent index is a memory (or LDS) round trip in the middle of a dependent chain
__device__ __forceinline__ void len_code(int li, uint32_t &base, int &extra)
{
    extra = li < 8 || li == 28 ? 0 : (li >> 2) - 1;
    base = li < 8 ? 3u + (uint32_t)li : li == 28 ? 258u : 3u + ((4u + ((uint32_t)li & 3u)) << extra);
}
__device__ __forceinline__ void dist_code(int ds, uint32_t &base, int &extra)
{
    extra = ds < 4 ? 0 : (ds >> 1) - 1;
    base = ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1u)) << extra);
}
__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// ---- the per-lane header check ------------------------------------------------------------------------------------
// Could a NON-FINAL DYNAMIC block header start at bit `p`?  One unaligned 16-byte load per lane (the header bits and the
// code-length code's lengths: 17 + at most 57 bits), a 64-byte table in LDS: block type, HLIT / HDIST in range, and the
// code-length code complete (zlib insists on that).  About one position in 2000 passes; those are then parsed and
// decoded by the whole wave (read_dynamic + inflate_block), which is the real test.
__device__ __forceinline__ uint4 header_bytes(const uint8_t *in, uint64_t p)
{
    uint4 v;
    __builtin_memcpy(&v, in + (p >> 3), 16);
    return v;
}
// The Kraft sum is taken two fields at a time from a 64-entry table in LDS (kraft2[a | b << 3] = the weights of the lengths a and
// b, 128 >> l for l > 0): ten look-ups instead of a loop of up to 19 steps that a wave runs as long as its longest lane (nearly
// always 19; round 5: sync_kernel 7.3 -> 6.4 ms per 410 MB file, the same candidates).  The fields beyond HCLEN are masked off first.
__device__ __forceinline__ bool header_prefilter_lut(const uint4 v, uint64_t p, const uint8_t *kraft2)
{
    uint64_t lo = (uint64_t)v.x | (uint64_t)v.y << 32, hi = (uint64_t)v.z | (uint64_t)v.w << 32;
    const int sh = (int)(p & 7);
    if (sh) { lo = (lo >> sh) | (hi << (64 - sh)); hi >>= sh; }
    if ((lo & 7u) != 4u) return false;
    const uint32_t hlit = (uint32_t)(lo >> 3) & 31u, hdist = (uint32_t)(lo >> 8) & 31u, hclen = ((uint32_t)(lo >> 13) & 15u) + 4u;
    if (hlit > 29u || hdist > 29u) return false;
    uint64_t w = (lo >> 17) | (hi << 47);                        // the 3-bit lengths, 57 bits at most
    w &= (1ull << (3u * hclen)) - 1ull;                          // (3 * 19 = 57 < 64)
    uint32_t kraft = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) kraft += kraft2[(uint32_t)(w >> (6 * i)) & 63u];
    return kraft == 128u;
}

// ---- per-wave decoder state: tables, ring and staging in LDS -------------------------------------------------------------
template <int PB, int MAXSYM>
struct LHuff {
    uint16_t tent[1 << PB];          // symbol | code length << 9 for every PB-bit pattern; 0 = code longer than PB bits
    // canonical code, for the codes longer than PB bits: ub[l] = the end of the length-l codes among the 15-bit left-justified
    // code values (they are sorted by length: a code's length is the number of ub[] it has reached), delta[l] = where the
    // length-l symbols start in sorted[] minus the first length-l code
    uint16_t ub[16];
    int16_t delta[16];
    uint16_t sorted[MAXSYM];
    int maxlen;
};
struct WaveState {
    union {                              // the code-length code is dead when the literal/length code is built from what it decoded
        LHuff<SS_GZ_LITBITS, 288> lit;
        LHuff<7, 19> clc;
    };
    LHuff<8, 32> dist;
    alignas(16) uint8_t lens[320];       // (code lengths while a header is read; 64 words of scratch while a block's items are decoded)
    alignas(16) uint16_t ring[RING];
    alignas(16) uint32_t stage[STAGE / 4];
#ifdef SS_GZ_PAD_LDS
    uint8_t pad[SS_GZ_PAD_LDS];        // (experiment: fewer waves per SIMD)
#endif
};

// Bit positions over the LDS stage (STAGE_DW dwords = two halves; the wave refills a half with one load per lane when the
// position has left it).  `bp` is the absolute bit position of the next unread bit; the block loop below lets the 64
// lanes decode at bp + lane, the headers are read through a small scalar cache (buf / cnt) of the bits at bp.
// Everything here is called wave-uniformly; LDS operations of one wave execute in order, so no barrier is needed
// between the lanes' stores and the loads that follow.
// Positions are RELATIVE to `base` (a bit position of the file, a multiple of a stage's bits, at or below everything the
// wave will look at) and 32 bits wide (round 4): the window loop compares the position with four limits per window, and
// 64-bit scalars can only be compared on the VALU (no s_cmp_lt_u64) and cost two SGPRs each -- the loop restored ~30 of them
// from spill lanes per window.  A wave stays within ~512 MB of its base (SB_CAP); beyond, the position is "past the end".
constexpr uint32_t SB_CAP = 0xFFF00000u, SB_NEVER = 0xFFFFFFFFu;
struct SBits {
    const uint32_t *in;      // global, dword aligned, padded with zeros behind the data: the dword of relative position 0
    uint64_t base;           // the file's bit position of relative position 0
    uint32_t nbits;          // bits of data from `base` on (capped)
    uint32_t bp;             // next bit
    uint32_t staged_to;      // dword index up to which the stage holds data: it covers [staged_to - STAGE_DW, staged_to)
    uint64_t buf;            // scalar cache: cnt bits from bp on
    int cnt;
};
__device__ __forceinline__ void sb_fill_half(WaveState &S, const SBits &b, uint32_t w0)
{
    // dwords [w0, w0 + HALF_DW) -> stage; w0 is a multiple of HALF_DW
    const int lane = threadIdx.x & 63;
    if constexpr (HALF_DW == 256) {
        const uint4 v = *reinterpret_cast<const uint4 *>(b.in + w0 + (uint64_t)lane * 4);
        *reinterpret_cast<uint4 *>(&S.stage[(w0 & STAGE_MASK) + lane * 4]) = v;
    } else {
        const uint2 v = *reinterpret_cast<const uint2 *>(b.in + w0 + (uint64_t)lane * 2);
        *reinterpret_cast<uint2 *>(&S.stage[(w0 & STAGE_MASK) + lane * 2]) = v;
    }
    __builtin_amdgcn_wave_barrier();
}
// the dwords [bp >> 5, (bp >> 5) + HALF_DW) are in the stage afterwards (a start, a jump or a step back reloads both halves;
// the sync search tries positions a few bits apart and finds its data still there)
__device__ __forceinline__ void sb_stage(WaveState &S, SBits &b)
{
    const uint32_t w = b.bp >> 5;
    if (w >= b.staged_to || w + STAGE_DW < b.staged_to) {
        const uint32_t h0 = w & ~(uint32_t)(HALF_DW - 1);
        __builtin_amdgcn_wave_barrier();
        sb_fill_half(S, b, h0);
        sb_fill_half(S, b, h0 + HALF_DW);
        b.staged_to = h0 + STAGE_DW;
    } else {
        while (w + HALF_DW >= b.staged_to) {     // the half behind the position is refilled with the data behind the other one
            sb_fill_half(S, b, b.staged_to);
            b.staged_to += HALF_DW;
        }
    }
}
// `bitpos`: where the wave starts; `lowest`: the lowest position it will ever seek to (a chunk that begins inside a block
// goes back to that block's header first)
__device__ __forceinline__ void sb_init(SBits &b, const uint8_t *p, uint64_t n, uint64_t bitpos, uint64_t lowest = ~0ull)
{
    const uint64_t base_dw = (min(bitpos, lowest) >> 5) & ~(uint64_t)(STAGE_DW - 1);
    b.in = reinterpret_cast<const uint32_t *>(p) + base_dw;
    b.base = base_dw << 5;
    const uint64_t total = n * 8;
    b.nbits = total <= b.base ? 0u : (uint32_t)min(total - b.base, (uint64_t)SB_CAP);
    b.bp = (uint32_t)min(bitpos - b.base, (uint64_t)SB_NEVER - 1);
    b.staged_to = 0; b.buf = 0; b.cnt = 0;
}
__device__ __forceinline__ void sb_seek(SBits &b, uint32_t rel) { b.bp = rel; b.cnt = 0; }                      // relative
// a position of the file -> relative (SB_NEVER: none, or out of this wave's reach)
__device__ __forceinline__ uint32_t sb_rel(const SBits &b, uint64_t bitpos)
{
    return (bitpos == ~0ull || bitpos - b.base >= (uint64_t)SB_CAP) ? SB_NEVER : (uint32_t)(bitpos - b.base);
}
__device__ __forceinline__ void sb_seek_abs(SBits &b, uint64_t bitpos) { sb_seek(b, sb_rel(b, bitpos)); }
// scalar cache: at least 33 bits from bp on.  Every lane reads the same words: telling the compiler so (readfirstlane)
// keeps the header logic on the scalar unit
__device__ __forceinline__ void sb_load(WaveState &S, SBits &b)
{
    sb_stage(S, b);
    const uint32_t w = b.bp >> 5;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.stage[w & STAGE_MASK]);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.stage[(w + 1) & STAGE_MASK]);
    const int sh = (int)(b.bp & 31);
    b.buf = (((uint64_t)hi << 32) | lo) >> sh;
    b.cnt = 64 - sh;
}
__device__ __forceinline__ void sb_need(WaveState &S, SBits &b, int k) { if (b.cnt < k) sb_load(S, b); }      // k <= 33
__device__ __forceinline__ uint32_t sb_peek(SBits &b, int k) { return (uint32_t)(b.buf & ((1ull << k) - 1)); }
__device__ __forceinline__ void sb_drop(SBits &b, int k) { b.buf >>= k; b.cnt -= k; b.bp += (uint32_t)k; }
__device__ __forceinline__ uint32_t sb_get(WaveState &S, SBits &b, int k)      // k <= 32
{
    sb_need(S, b, k);
    const uint32_t v = sb_peek(b, k);
    sb_drop(b, k);
    return v;
}
__device__ __forceinline__ uint64_t sb_bitpos(const SBits &b) { return b.base + b.bp; }                         // of the file
__device__ __forceinline__ bool sb_past_end(const SBits &b) { return b.bp > b.nbits; }

// canonical code from the lengths lens[0 .. n) (LDS): 0 complete, 1 incomplete, -1 over-subscribed (nothing built).
// The 64 lanes hold the lengths of symbols lane, lane + 64, ...; a ballot per code length counts the codes and ranks a
// lane's symbol among those of its length, so the Kraft sum rejects a wrong header (the sync search sees ~80 per chunk)
// before any table is written.
template <int PB, int MAXSYM, int MAXLEN = 15>
__device__ int huff_build(LHuff<PB, MAXSYM> &h, const uint8_t *lens, int n)
{
    constexpr int NS = (MAXSYM + 63) / 64;
    const int lane = threadIdx.x & 63;
    __syncthreads();
    int ls[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) ls[s] = lane + 64 * s < n ? (int)lens[lane + 64 * s] : 0;
    uint32_t cnt[16];
    cnt[0] = 0;
#pragma unroll
    for (int l = 1; l <= MAXLEN; l++) {
        uint32_t c = 0;
#pragma unroll
        for (int s = 0; s < NS; s++) c += (uint32_t)__popcll(__ballot(ls[s] == l));
        cnt[l] = c;
    }
    int left = 1, maxlen = 0;
    bool over = false;
#pragma unroll
    for (int l = 1; l <= MAXLEN; l++) {
        left = 2 * left - (int)cnt[l];
        over = over || left < 0;
        if (cnt[l]) maxlen = l;
    }
    if (over) return -1;
    for (int e = lane; e < (1 << PB); e += 64) h.tent[e] = 0;
    __syncthreads();
    uint32_t code = 0, off = 0;
#pragma unroll
    for (int l = 1; l <= MAXLEN; l++) {
        const uint32_t c = cnt[l];
        if (lane == 0) { h.ub[l] = (uint16_t)((code + c) << (15 - l)); h.delta[l] = (int16_t)((int)off - (int)code); }
        if (c) {
            uint32_t running = 0;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const uint64_t bl = __ballot(ls[s] == l);
                if (ls[s] == l) {
                    const uint32_t k = running + (uint32_t)__popcll(bl & (lane ? (~0ull >> (64 - lane)) : 0ull));
                    const uint16_t sym = (uint16_t)(lane + 64 * s);
                    h.sorted[off + k] = sym;
                    if (l <= PB) {
                        const uint32_t r = __brev(code + k) >> (32 - l);
                        for (uint32_t e = r; e < (1u << PB); e += 1u << l) h.tent[e] = (uint16_t)(sym | (l << 9));
                    }
                }
                running += (uint32_t)__popcll(bl);
            }
        }
        code = (code + c) << 1;
        off += c;
    }
    if (lane == 0) h.maxlen = maxlen;
    __syncthreads();
    return left > 0 ? 1 : 0;
}
// inclusive prefix sum over the 64 lanes in six DPP additions (row shifts 1, 2, 4, 8, then the row broadcasts of gfx9)
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
#ifdef SS_GZ_SCAN_SHFL
    for (int d = 1; d < 64; d <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)v, d, 64); if ((int)(threadIdx.x & 63) >= d) v += u; }
    return v;
#endif
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);      // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
}

// inclusive prefix MAXIMUM over the 64 lanes (unsigned, 0 = nothing yet), the same six DPP steps
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v)
{
    int x = (int)v;
#define SS_MAXU(a, b) (int)max((uint32_t)(a), (uint32_t)(b))
    x = SS_MAXU(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));      // row_shr:1
    x = SS_MAXU(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));      // row_shr:2
    x = SS_MAXU(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));      // row_shr:4
    x = SS_MAXU(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));      // row_shr:8
    x = SS_MAXU(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));      // row_bcast:15 into rows 1 and 3
    x = SS_MAXU(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));      // row_bcast:31 into rows 2 and 3
#undef SS_MAXU
    return (uint32_t)x;
}

__device__ bool read_dynamic(WaveState &S, SBits &b, bool &dist_usable)
{
    const int hlit = (int)sb_get(S, b, 5) + 257, hdist = (int)sb_get(S, b, 5) + 1, hclen = (int)sb_get(S, b, 4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    __syncthreads();
    if ((threadIdx.x & 63) < 19) S.lens[threadIdx.x & 63] = 0;
    __syncthreads();
    {
        // the HCLEN 3-bit lengths (at most 57 bits) in one go: lane i takes the i-th
        sb_stage(S, b);
        const int li = threadIdx.x & 63;
        const uint32_t p = b.bp + 3u * (uint32_t)li;
        const uint32_t w = p >> 5, sh = p & 31u;
        const uint32_t d0 = S.stage[w & STAGE_MASK], d1 = S.stage[(w + 1) & STAGE_MASK];
        const uint32_t v = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh) & 7u;
        if (li < hclen) S.lens[c_cl_order[li]] = (uint8_t)v;
        sb_seek(b, b.bp + 3u * (uint32_t)hclen);
    }
    if (__builtin_amdgcn_readfirstlane(huff_build<7, 19, 7>(S.clc, S.lens, 19)) != 0) return false;   // (a call's result arrives in a VGPR)
    // the code lengths themselves, as the block loop decodes its symbols: the 64 lanes decode a code-length symbol (with its
    // extra bits: at most 14 bits) at 64 bit positions, the wave follows the chain from bp and hands every item on it
    // its value ("repeat the previous length" resolved on the way) and its place; the lanes then store their runs.
    // [One symbol at a time was ~300 cycles x 300 symbols for each of the ~80 candidates a chunk's sync search tries.]
    __shared__ uint8_t s_all[320];
    const int total = hlit + hdist, lane = threadIdx.x & 63;
    int i = 0, prev = -1;
    uint32_t kraft_lit = 0;
    while (i < total) {
        sb_stage(S, b);
        const uint32_t p = b.bp + (uint32_t)lane;
        const uint32_t w = p >> 5, sh = p & 31u;
        const uint32_t d0 = S.stage[w & STAGE_MASK], d1 = S.stage[(w + 1) & STAGE_MASK];
        const uint32_t bits = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh);
        const uint32_t e = S.clc.tent[bits & 127u];
        const uint32_t l = e >> 9, sy = e & 511u;
        const uint32_t xb = sy == 16 ? 2u : sy == 17 ? 3u : sy == 18 ? 7u : 0u;
        const uint32_t x = (bits >> l) & ((1u << xb) - 1u);
        const uint32_t rep = sy < 16 ? 1u : sy == 18 ? 11u + x : 3u + x;
        const uint32_t pk = e ? ((l + xb) | (rep << 4) | ((sy < 16 ? sy : sy == 16 ? 16u : 0u) << 12)) : 0u;      // bits | run | value (16 = copy)
        // the chain from bp (an undecodable position steps out of the window), then by prefix sum and ballots: which of
        // its items still belong to the lengths, where their runs go, and what "copy" copies
        const uint32_t step = (pk & 15u) ? (pk & 15u) : 64u;
        uint64_t chain = 0;
        uint32_t pos = 0;
        do {
            chain |= 1ull << pos;
            pos += (uint32_t)__builtin_amdgcn_readlane((int)step, (int)pos);
        } while (pos < 64);
        const bool on = (chain >> lane) & 1ull;
        const uint32_t incl = wave_inclusive_sum(on ? rep : 0u), before = incl - (on ? rep : 0u);
        const uint32_t left = (uint32_t)(total - i);
        const bool used = on && before < left;
        const uint64_t usedm = __ballot(used);                                   // lane 0 is in it
        const int lu = 63 - __clzll((long long)usedm);
        const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)incl, lu);
        if (__ballot(used && (pk & 15u) == 0) || end > left) return false;      // undecodable, or a run beyond the last length
        const uint32_t vc = pk >> 12;
        const uint64_t sources = __ballot(used && vc != 16u) & ((lane == 63 ? 0ull : (~0ull << (lane + 1))) ^ ~0ull);   // at or below this lane
        const int src = sources ? 63 - __clzll((long long)sources) : -1;
        const int got = __shfl((int)vc, src < 0 ? 0 : src, 64);
        const int val = src < 0 ? prev : got;
        if (__ballot(used && val < 0)) return false;                             // "repeat the previous length" with none before
        // Kraft sums as the lengths arrive: a position that is no header -- the sync search tries ~40 per chunk -- decodes to
        // random lengths, and those over-subscribe a code within the first window or two (a symbol adds 1 700 / 32 768 on average):
        // out here, not after all ~300 lengths and the counting of huff_build
        {
            const uint32_t idx0 = (uint32_t)i + before;                          // (the literal/length code's lengths come first)
            const uint32_t n_lit = !used || idx0 >= (uint32_t)hlit ? 0u : min(rep, (uint32_t)hlit - idx0);
            const uint32_t unit = val > 0 ? (1u << (15 - val)) : 0u;
            kraft_lit += (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(n_lit * unit), 63);
            if (kraft_lit > 32768u) return false;
        }
        if (used)
            for (uint32_t r = 0; r < rep; r++) s_all[i + (int)before + (int)r] = (uint8_t)val;
        prev = __builtin_amdgcn_readlane(val, lu);
        i += (int)end;
        sb_seek(b, b.bp + (uint32_t)lu + (uint32_t)__builtin_amdgcn_readlane((int)step, lu));
        if (sb_past_end(b)) return false;
    }
    __syncthreads();
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

#ifdef SS_GZ_TIMING
// diagnostic build: cycles per section, summed over the waves (0 decode, 1 chain, 2 deliver, 3 flush, 4 header, 5 windows)
__device__ unsigned long long g_gz_t[12];
#define GZ_T(var) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const uint64_t var = clock64()
#define GZ_ACC(i, a, b) (o.t[i] += (b) - (a))
#else
#define GZ_T(var)
#define GZ_ACC(i, a, b)
#endif

// Output of a wave: symbols go to the LDS ring first; whenever 1024 new ones have gathered, the 64 lanes write them to
// global memory together (128 contiguous bytes per store instruction).  A match reads its source from the ring when it
// is at most RING_REACH symbols back (always true for what is not flushed yet) and from global memory otherwise --
// data flushed at least two flushes ago; every flush first waits for the stores of the one before.
constexpr uint32_t RING_REACH = RING - 264;
struct OutState {
    uint16_t *out;           // null: count only
    uint32_t cap, n, flushed;        // symbols: a chunk's output is far below 2^31 (its region is capped when the state is made)
#ifdef SS_GZ_TIMING
    uint64_t t[10];
#endif
};
__device__ __forceinline__ void out_flush(WaveState &S, OutState &o, uint32_t upto)
{
    const int lane = threadIdx.x & 63;
    GZ_T(tf0);
    __threadfence_block();
    for (uint32_t i = o.flushed + (uint32_t)lane; i < upto; i += 64) o.out[i] = S.ring[i & (RING - 1)];
    o.flushed = upto;
    GZ_T(tf1);
    GZ_ACC(3, tf0, tf1);
}

// a code longer than the table's PB bits (or none at all), per lane, without a loop: the 15 bits as a left-justified code
// value; canonical codes are sorted by length, so the length is PB + 1 + the number of ends ub[PB + 1 .. 14] the value has reached
template <int PB, int MAXSYM>
__device__ __forceinline__ int lane_long(const LHuff<PB, MAXSYM> &h, uint32_t v, int maxlen, int &len)
{
    (void)maxlen;
    const uint32_t c15 = __brev(v & 0x7FFFu) >> 17;                  // the first bit of the stream is the code's most significant
    uint32_t l = PB + 1;
#pragma unroll
    for (int k = PB + 1; k <= 14; k++) l += c15 >= (uint32_t)h.ub[k] ? 1u : 0u;
    const bool ok = c15 < (uint32_t)h.ub[l];                           // (beyond the last code: an incomplete code's unused values)
    len = ok ? (int)l : 0;
    const int at = (int)h.delta[l] + (int)(c15 >> (15u - l));
    return ok ? (int)h.sorted[ok ? at : 0] : -1;
}

// `len` symbols from `dist` back to position n of the wave's output (the 64 lanes together)
__device__ __forceinline__ void copy_match(WaveState &S, const OutState &o, uint32_t n, uint32_t len, uint32_t dist)
{
    const int lane = threadIdx.x & 63;
    const uint32_t n32 = n;
    const bool overlaps = dist < len;                  // (uniform: the division below is rarely reached)
    for (uint32_t base = 0; base < len; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        if (i < len) {
            const uint32_t k = overlaps ? i % dist : i;             // a match that overlaps itself repeats with period dist
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

// The symbols of a compressed block (block_body below): the 64 lanes decode SPECULATIVELY at the 64 bit positions bp .. bp + 63 -- each a
// whole item (literal, end of block, or length + extra + distance + extra: at most 48 bits, every lane has 64) --, then
// the wave follows the chain of item lengths from bp (one readlane per item on the scalar unit) and the lanes ON the
// chain deliver at the offsets a prefix sum over their output lengths gives them: the literals in one store; the
// matches whose source was flushed to global memory long ago (85 % of them in FASTQ: k-mers seen 10 KB earlier) load
// their few symbols TOGETHER -- one memory round trip per window of ~18 items instead of one per match --; matches
// that read the ring (their source may be this window's own output) follow one by one.  [A wave-uniform decoder -- one
// table lookup and ~40 instructions per symbol, 63 lanes idle, every far match a round trip -- took 47 ms per 1 M reads.]
constexpr uint32_t F_MATCH = 64u, F_EOB = 128u, F_BAD = 256u;
constexpr uint32_t FAR_MAX = 12;            // longest match of the batched kind (one 2-byte load per symbol and lane)
constexpr uint32_t WIN_MAX = RING - RING_REACH;      // 264: a window that delivers more goes item by item (ring aliasing)
// the block's first bits: BFINAL, BTYPE and the code tables (a stored block is copied here and now).
// 0 = tables in LDS, block_body follows; 10 + BFINAL = a stored block, done; < 0 = error
struct BlockCtx { uint32_t bfinal; bool dist_usable; };
__device__ int block_begin(WaveState &S, SBits &b, OutState &o, BlockCtx &cx)
{
    const int lane = threadIdx.x & 63;
    const bool store = o.out != nullptr;
    const uint32_t bfinal = sb_get(S, b, 1), btype = sb_get(S, b, 2);
    cx.bfinal = bfinal;
    cx.dist_usable = true;
    if (btype == 3) return -1;
    uint32_t n = o.n;
#define GI_RET(v) do { o.n = n; return (v); } while (0)
    if (btype == 0) {
        sb_seek(b, (b.bp + 7u) & ~7u);
        const uint32_t len = sb_get(S, b, 16), nlen = sb_get(S, b, 16);
        if (sb_past_end(b) || (len ^ 0xFFFFu) != nlen) GI_RET(-2);
        if (b.bp + 8u * len > b.nbits) GI_RET(-3);
        if (store) {
            if (n + len > o.cap) GI_RET(-9);
            const uint8_t *src = reinterpret_cast<const uint8_t *>(b.in) + (b.bp >> 3);
            uint32_t done = 0;
            while (done < len) {
                if (n - o.flushed >= 1024) out_flush(S, o, n);
                const uint32_t take = min(len - done, 1024u - (n - o.flushed));
                for (uint32_t i = (uint32_t)lane; i < take; i += 64) S.ring[(n + i) & (RING - 1)] = src[done + i];
                n += take;
                done += take;
            }
        } else {
            n += len;
        }
        sb_seek(b, b.bp + 8u * len);
        GI_RET(10 + (int)bfinal);
    }
    GZ_T(th0);
    if (btype == 1) fixed_codes(S);
    else if (!read_dynamic(S, b, cx.dist_usable)) GI_RET(-4);
    GZ_T(th1);
    GZ_ACC(4, th0, th1);
    return 0;
#undef GI_RET
}

// The items of a compressed block from the current position (the block's first item, or -- a chunk that was entered inside a
// block, or one that goes on behind an entry it ran over -- any item's first bit) to its end-of-block code.
// 0 = block done, 1 = final block done, 2 = probe satisfied, 3 = stopped exactly AT mid_target (an item's first bit inside
// this block: the next chunk's entry), 4 = mid_target lies behind the position and was not an item's first bit, < 0 = error.
__device__ int block_body(WaveState &S, SBits &b, OutState &o, bool known_window, const BlockCtx cx, uint64_t probe_symbols, uint32_t stop_bits,
                          uint32_t mid_target)      // (stop_bits, mid_target: relative positions, sb_rel; SB_NEVER = none)
{
    const int lane = threadIdx.x & 63;
    const bool store = o.out != nullptr;
    const uint32_t bfinal = cx.bfinal;
    const bool dist_usable = cx.dist_usable;
    uint32_t n = o.n;
    const int lit_maxlen = __builtin_amdgcn_readfirstlane(S.lit.maxlen), dist_maxlen = __builtin_amdgcn_readfirstlane(S.dist.maxlen);
    const uint32_t probe_end = probe_symbols >= 0x7FFFFFFFull ? SB_NEVER : n + (uint32_t)probe_symbols;
    const uint32_t end_bits = b.nbits;
    const uint64_t mine_below = lane ? (~0ull >> (64 - lane)) : 0ull;
    // the window whose far matches are still loading while the next one is decoded (software pipeline: a far match reads
    // what this wave wrote tens of KB ago -- with thousands of waves at work that is HBM or the Infinity Cache, ~2 us)
    bool p_active = false, p_fv = false;
    uint64_t p_farmask = 0, p_near = 0;
    uint32_t p_at = 0, p_mlen = 0, p_val = 0, p_fdst = 0;
    uint16_t p_fg = 0;
    int ret = 0;
#define GI_FINISH()                                                                                                          \
    do {                                                                                                                     \
        if (p_active) {                                                                                                      \
            if (p_farmask && p_fv) S.ring[p_fdst & (RING - 1)] = p_fg;                                                       \
            uint64_t mt_ = p_near;                                                                                           \
            while (mt_) {                                                                                                    \
                const int fm_ = __ffsll((long long)mt_) - 1;                                                                 \
                mt_ &= mt_ - 1;                                                                                              \
                copy_match(S, o, (uint32_t)__builtin_amdgcn_readlane((int)p_at, fm_), (uint32_t)__builtin_amdgcn_readlane((int)p_mlen, fm_), \
                           (uint32_t)__builtin_amdgcn_readlane((int)p_val, fm_));                                            \
            }                                                                                                                \
            p_active = false;                                                                                                \
        }                                                                                                                    \
    } while (0)
#define GI_OUT(v) do { ret = (v); goto out; } while (0)
    for (;;) {
        if (store && n - o.flushed >= 1024) { GI_FINISH(); out_flush(S, o, n); }
        if (n >= probe_end) GI_OUT(2);                     // sync search: the header was valid and this many symbols decoded
        if (b.bp > end_bits) GI_OUT(-5);
        if (b.bp > stop_bits) GI_OUT(-21);                 // ran past the next chunk's entry: that entry was not a block's start
        if (b.bp >= mid_target) GI_OUT(b.bp == mid_target ? 3 : 4);
        GZ_T(tw0);
        sb_stage(S, b);
        // ---- every lane: the item at bp + lane
        uint32_t info, val = 0, mlen = 0;                  // info = bits of the item | F_*; val = literal byte or match distance
        {
            const uint32_t p = b.bp + (uint32_t)lane;
            const uint32_t w = p >> 5, sh = p & 31u;
            const uint32_t d0 = S.stage[w & STAGE_MASK], d1 = S.stage[(w + 1) & STAGE_MASK], d2 = S.stage[(w + 2) & STAGE_MASK];
            uint64_t bits = (((uint64_t)d1 << 32) | d0) >> sh;
            if (sh) bits |= (uint64_t)d2 << (64 - sh);
            const uint32_t e = S.lit.tent[(uint32_t)bits & ((1u << SS_GZ_LITBITS) - 1)];
            int l = (int)(e >> 9), sym = (int)(e & 511u);
            if (e == 0) sym = lane_long(S.lit, (uint32_t)bits & 0x7FFFu, lit_maxlen, l);
            if (sym < 0 || sym > 285) info = F_BAD;
            else if (sym < 256) { info = (uint32_t)l; val = (uint32_t)sym; }
            else if (sym == 256) info = (uint32_t)l | F_EOB;
            else {
                int xb, xd;
                uint32_t base;
                len_code(sym - 257, base, xb);
                mlen = base + ((uint32_t)(bits >> l) & ((1u << xb) - 1u));
                l += xb;
                const uint32_t v = (uint32_t)(bits >> l) & 0x7FFFu;
                const uint32_t de = S.dist.tent[v & 255u];
                int dl = (int)(de >> 9), ds = (int)(de & 511u);
                if (de == 0) ds = lane_long(S.dist, v, dist_maxlen, dl);
                if (!dist_usable || ds < 0 || ds > 29) info = F_BAD;
                else {
                    dist_code(ds, base, xd);
                    val = base + ((uint32_t)(bits >> (l + dl)) & ((1u << xd) - 1u));
                    info = (uint32_t)(l + dl + xd) | F_MATCH;
                }
            }
        }
        GZ_T(tw1);
        // ---- the chain of items from bp: a stopping item (end of block, undecodable) steps out of the window
        const uint32_t step = (info & (F_EOB | F_BAD)) ? 64u : (info & 63u);
        // (three scalar instructions, a readlane and the branch per item: the bit is set in place, the last item is looked up
        //  afterwards -- the scalar unit is shared by the CU's four SIMDs and this kernel keeps it as busy as the vector units)
        uint64_t chain = 0;
        uint32_t pos = 0;
        do {
            asm volatile("s_bitset1_b64 %0, %1" : "+s"(chain) : "s"(pos));
            pos += (uint32_t)__builtin_amdgcn_readlane((int)step, (int)pos);
        } while (pos < 64);
        const uint32_t last = 63u - (uint32_t)__clzll((long long)chain);
        uint32_t stop = (uint32_t)__builtin_amdgcn_readlane((int)info, (int)last);
        stop = (stop & (F_EOB | F_BAD)) ? stop : 0u;
        if (stop) { chain &= ~(1ull << last); pos = last; }
        if (mid_target - b.bp < 64u && ((chain >> (mid_target - b.bp)) & 1ull)) {      // the next chunk begins at an item of this window
            pos = mid_target - b.bp;
            chain &= (1ull << pos) - 1ull;
            stop = 0;
        }
        if (stop & F_BAD) GI_OUT(-5);
        GZ_T(tw2);
        // ---- deliver
        const bool on = (chain >> lane) & 1ull, is_match = on && (info & F_MATCH);
        const uint64_t matches = __ballot(is_match);
        const uint32_t olen = on ? (is_match ? mlen : 1u) : 0u;
        const uint32_t incl = wave_inclusive_sum(olen);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t at = n + (incl - olen);             // where this lane's item goes
        if (known_window && __ballot(is_match && val > at)) GI_OUT(-8);
#ifdef SS_GZ_TIMING
        o.t[6] += (uint64_t)__popcll(matches); o.t[7] += total; o.t[8] += (uint64_t)__popcll(__ballot(is_match && val > RING_REACH + mlen - 1));
#endif
        if (store) {
            if (n + total > o.cap) GI_OUT(-9);
            GI_FINISH();                                   // the window before this one: its far symbols have arrived by now
            if (total <= WIN_MAX) {
                if (on && !is_match) S.ring[at & (RING - 1)] = (uint16_t)val;
                // matches from far back, few symbols: all their loads in flight together, and left in flight
#ifdef SS_GZ_NOFAR
                const bool far = false;
#else
                const bool far = is_match && mlen <= FAR_MAX && val > RING_REACH + mlen - 1;      // every symbol beyond the ring's reach
#endif
                // ONE LANE PER SYMBOL of those matches (a window has ~4 of them, ~30 symbols): a prefix sum over their lengths
                // gives every match its run of slots, the match's lane number is dropped at the run's first slot (LDS) and
                // spread over the run by a prefix maximum; slot t then fetches its match's position and distance (two
                // bpermutes) and loads ITS symbol -- one load instruction per window instead of twelve
                const uint32_t fl = far ? mlen : 0u;
                const uint32_t fincl = wave_inclusive_sum(fl);
                const bool far2 = far && fincl <= 64u;     // (what does not fit the 64 slots goes the near way)
                p_farmask = __ballot(far2);
                if (p_farmask) {
                    uint32_t *own = reinterpret_cast<uint32_t *>(S.lens);
                    __builtin_amdgcn_wave_barrier();
                    own[lane] = 0u;
                    __builtin_amdgcn_wave_barrier();
                    if (far2) own[fincl - fl] = (uint32_t)lane + 1u;
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t ow = wave_inclusive_max(own[lane]);
                    const uint32_t slots = (uint32_t)__builtin_amdgcn_readlane((int)fincl, 63 - __clzll((long long)p_farmask));
                    const int from = (int)((ow ? ow - 1u : 0u) << 2);
                    const uint32_t o_at = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)at);
                    const uint32_t o_vf = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(val | ((fincl - fl) << 16)));      // distance <= 32768 | first slot
                    const uint32_t k = (uint32_t)lane - (o_vf >> 16);
                    const int32_t sp = (int32_t)o_at - (int32_t)(o_vf & 0xFFFFu) + (int32_t)k;
                    p_fv = (uint32_t)lane < slots;
                    uint16_t g = (uint16_t)(UNRES | (uint32_t)((int32_t)WSIZE + sp));
                    if (p_fv && sp >= 0) g = o.out[sp];
                    p_fg = g;
                    p_fdst = o_at + k;
                }
                p_at = at; p_mlen = mlen; p_val = val;
                p_near = matches & ~p_farmask;
                p_active = true;
                n += total;
            } else {
                // a window that delivers a lot (long matches): item by item, flushing on the way
                uint64_t rem = chain, mt = matches;
                for (;;) {
                    const int fm = mt ? __ffsll((long long)mt) - 1 : 64;
                    const uint64_t lits = (fm == 64 ? rem : rem & ((1ull << fm) - 1));
                    if ((lits >> lane) & 1ull) S.ring[(n + (uint32_t)__popcll(lits & mine_below)) & (RING - 1)] = (uint16_t)val;
                    n += (uint32_t)__popcll(lits);
                    if (fm == 64) break;
                    const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)mlen, fm), dist = (uint32_t)__builtin_amdgcn_readlane((int)val, fm);
                    if (n - o.flushed >= 1024) out_flush(S, o, n);
                    copy_match(S, o, n, len, dist);
                    n += len;
                    mt &= mt - 1;
                    rem = fm == 63 ? 0ull : rem & ~((2ull << fm) - 1ull);
                }
            }
        } else {
            n += total;
        }
        GZ_T(tw3);
#ifdef SS_GZ_TIMING
        GZ_ACC(0, tw0, tw1); GZ_ACC(1, tw1, tw2); GZ_ACC(2, tw2, tw3); o.t[5]++;
#endif
        if (stop & F_EOB) {
            sb_seek(b, b.bp + pos + (stop & 63u));
            GI_OUT(sb_past_end(b) ? -5 : (int)bfinal);
        }
        sb_seek(b, b.bp + pos);
    }
out:
    if (store) GI_FINISH();
    o.n = n;
    return ret;
#undef GI_FINISH
#undef GI_OUT
}

// One block from the current position.  0 = block done, 1 = final block done, 2 = probe satisfied, < 0 = error.
__device__ int inflate_block(WaveState &S, SBits &b, OutState &o, bool known_window, uint64_t probe_symbols = ~0ull, uint32_t stop_bits = SB_NEVER)
{
    BlockCtx cx;
    const int r = block_begin(S, b, o, cx);
    if (r != 0) return r >= 10 ? r - 10 : r;
    return block_body(S, b, o, known_window, cx, probe_symbols, stop_bits, SB_NEVER);
}

__device__ unsigned g_sync_tries;      // candidates that passed the header check and were decoded (trace)

// ---- kernels ------------------------------------------------------------------------------------------------------
// A: entry point of every chunk (chunk 0: the start of the deflate data)
// slice_chunks != 0 (several ranks share a file, gpu_gunzip's range mode): a rank looks for entries only in the slices it
// owns (slice s belongs to rank s mod world) and in the first `look` chunks behind each of them (where its last chunk stops)
__global__ __launch_bounds__(64) void sync_kernel(const uint8_t *in, uint64_t in_n, uint64_t data_off, uint64_t chunk_bytes,
                                                  uint32_t n_chunks, uint64_t *entry /* bit position or ~0 */, uint64_t probe, int count_tries,
                                                  uint32_t slice_chunks, uint32_t rank, uint32_t world, uint32_t look, uint32_t c0)
{
    __shared__ WaveState S;
    const uint32_t c = blockIdx.x + c0;                       // (c0: the search runs piece by piece while the image arrives)
    const int lane = threadIdx.x & 63;
    if (slice_chunks) {
        const uint32_t sl = c / slice_chunks, within = c % slice_chunks;
        const bool mine = sl % world == rank || (sl > 0 && (sl - 1) % world == rank && within < look);
        if (!mine) { if (lane == 0) entry[c] = ~0ull; return; }
    }
    if (c == 0) { if (lane == 0) entry[0] = data_off * 8; return; }
    const uint64_t lo = (data_off + (uint64_t)c * chunk_bytes) * 8, hi = min(in_n * 8, lo + chunk_bytes * 8);
    uint64_t found = ~0ull;
    SBits b;
    sb_init(b, in, in_n, lo);
    // Two sieves.  The first 13 bits (type, HLIT, HDIST) are tested for every position -- a lane takes the eight positions of one
    // byte from a single 4-byte load --; the one in nine that passes is QUEUED, and the second sieve -- the code-length code's
    // Kraft sum, a loop of up to 19 steps that a wave runs as long as its slowest lane -- only sees full waves of queued
    // positions (a ninth of the loop's executions).  The queue is in position order, so the first confirmed entry is still
    // the first in the chunk.  (lo and hi are whole bytes.)
    uint32_t *queue = reinterpret_cast<uint32_t *>(S.ring);      // (1024 words, a ring; the LDS ring holds output, and the search produces none)
    constexpr uint32_t QMASK = RING * 2 / 4 - 1;
    static_assert(RING * 2 / 4 >= 1024, "the queue takes up to 63 + 512 positions");
    __shared__ uint8_t kraft2[64];                          // (8080 + 64 bytes: still 20 workgroups of one wave per CU)
    {
        const uint32_t la = (uint32_t)lane & 7u, lb = (uint32_t)lane >> 3;
        kraft2[lane] = (uint8_t)((la ? 128u >> la : 0u) + (lb ? 128u >> lb : 0u));
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t qh = 0, qn = 0;
    auto word_at = [&](uint64_t p) { uint32_t w; __builtin_memcpy(&w, in + (p >> 3) + (uint64_t)lane, 4); return w; };
    uint32_t ahead = word_at(lo);                           // the next 512 positions' bytes are loaded while these are tested
    for (uint64_t p0 = lo; p0 < hi && found == ~0ull; p0 += 512) {
        const uint32_t w = ahead;
        ahead = word_at(p0 + 512);                          // (the input is padded by 8 KB)
        const uint64_t pl = p0 + 8ull * (uint64_t)lane;
        // all eight positions of the byte at once, bit j of every term = the test at position j: BFINAL = 0 and BTYPE = 2 are
        // "bits j, j + 1 clear, bit j + 2 set"; a five-bit field (LSB first) is 30 or 31 iff its upper four bits are all set:
        // HLIT at j + 3 .. j + 7, HDIST at j + 8 .. j + 12 (bit 19 of the word at most)
        const uint32_t nw = ~w;
        const uint32_t type_ok = nw & (nw >> 1) & (w >> 2);
        const uint32_t hlit_big = (w >> 4) & (w >> 5) & (w >> 6) & (w >> 7);
        const uint32_t hdist_big = (w >> 9) & (w >> 10) & (w >> 11) & (w >> 12);
        uint32_t m8 = type_ok & ~hlit_big & ~hdist_big & 0xFFu;
        if (pl >= hi) m8 = 0;
        const uint32_t cnt = (uint32_t)__popc(m8);
        const uint32_t incl = wave_inclusive_sum(cnt);
        uint32_t at = qh + qn + incl - cnt;
        while (__ballot(m8 != 0u)) {
            if (m8) {
                const int j = __ffs((int)m8) - 1;
                m8 &= m8 - 1u;
                queue[at & QMASK] = (uint32_t)(pl - lo) + (uint32_t)j;
                at++;
            }
        }
        qn += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const bool last = p0 + 512 >= hi;
        while ((qn >= 64u || (last && qn > 0u)) && found == ~0ull) {
            const uint32_t take = min(qn, 64u);
            __builtin_amdgcn_wave_barrier();
            const uint32_t q = (uint32_t)lane < take ? queue[(qh + (uint32_t)lane) & QMASK] : 0u;
            const uint64_t pos = lo + (uint64_t)q;
            const bool ok = (uint32_t)lane < take && header_prefilter_lut(header_bytes(in, pos), pos, kraft2);
            uint64_t m = __ballot(ok);
            while (m && found == ~0ull) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint64_t cand = lo + (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)q, l);
                // confirm with the whole wave: a complete, valid header (zlib's rules) and 512 symbols that decode: a
                // position inside a block passes this with negligible probability, and if one ever does, the chunk in front
                // runs over it (inflate_kernel)
                sb_seek_abs(b, cand);
                OutState o{nullptr, 0, 0, 0};
                const int r = inflate_block(S, b, o, false, probe);
                if (count_tries && lane == 0) atomicAdd(&g_sync_tries, 1u);
                if (r == 2 || (r == 0 && o.n > 0)) found = cand;
            }
            __builtin_amdgcn_wave_barrier();
            qh = (qh + take) & QMASK;
            qn -= take;
        }
    }
    if (lane == 0) entry[c] = found;
}

// A2: entries INSIDE blocks.  A block is the unit the search can find (a sync point exists only at a block's start), and a
// wave that decodes 30-60 KB of deflate data alone is what the inflate kernel waits for.  But Huffman-coded data
// synchronises itself: decode from ANY bit with the block's tables and after a few dozen items the decoder is on the true
// item boundaries.  One wave per wanted position `from` inside the block whose header is at `hdr`: parse the header, then
// window by window (64 bits) every lane decodes the item at its bit as the inflate kernel does; all 64 offsets are
// hypotheses at first, a window maps the live ones to the offsets at which their chains enter the next window (chains
// followed by pointer doubling: six bpermutes), dead ends (undecodable, end of block) drop out.  When ONE offset is left it is the
// true chain -- the true one never dies before the block ends -- and becomes the entry: a chunk that begins there with
// `hdr`'s tables.  Verified when the chunks are inflated: the chunk in front must arrive exactly there, inside that block.
constexpr int SUB_WINDOWS = 48;
__device__ __forceinline__ uint32_t lane_item_bits(const WaveState &S, uint64_t bits, int lit_maxlen, int dist_maxlen, bool dist_usable)
{
    const uint32_t e = S.lit.tent[(uint32_t)bits & ((1u << SS_GZ_LITBITS) - 1)];
    int l = (int)(e >> 9), sym = (int)(e & 511u);
    if (e == 0) sym = lane_long(S.lit, (uint32_t)bits & 0x7FFFu, lit_maxlen, l);
    if (sym < 0 || sym > 285 || sym == 256) return 0;           // undecodable, or the end of the block: the hypothesis ends
    if (sym < 256) return (uint32_t)l;
    int xb, xd;
    uint32_t base;
    len_code(sym - 257, base, xb);
    l += xb;
    const uint32_t v = (uint32_t)(bits >> l) & 0x7FFFu;
    const uint32_t de = S.dist.tent[v & 255u];
    int dl = (int)(de >> 9), ds = (int)(de & 511u);
    if (de == 0) ds = lane_long(S.dist, v, dist_maxlen, dl);
    if (!dist_usable || ds < 0 || ds > 29) return 0;
    dist_code(ds, base, xd);
    return (uint32_t)(l + dl + xd);
}
__global__ __launch_bounds__(64) void subsync_kernel(const uint8_t *in, uint64_t in_n, const uint64_t *hdr, const uint32_t *first /* [n_blocks + 1] */,
                                                     const uint64_t *from, const uint64_t *limit, uint32_t n_blocks, uint64_t *entry)
{
    __shared__ WaveState S;
    __shared__ uint32_t flag[64];
    const uint32_t blk = blockIdx.x;                          // one wave per block: its header is parsed once for all its positions
    if (blk >= n_blocks) return;
    const int lane = threadIdx.x & 63;
    SBits b;
    sb_init(b, in, in_n, hdr[blk]);
    OutState o{nullptr, 0, 0, 0};
    BlockCtx cx;
    const bool usable = block_begin(S, b, o, cx) == 0;
    const uint64_t body = sb_bitpos(b);
    const int lit_maxlen = __builtin_amdgcn_readfirstlane(S.lit.maxlen), dist_maxlen = __builtin_amdgcn_readfirstlane(S.dist.maxlen);
    for (uint32_t i = first[blk]; i < first[blk + 1]; i++) {
        uint64_t found = ~0ull;
        const uint64_t p0 = from[i], lim = limit[i];
        if (usable && p0 >= body) {
            uint64_t live = ~0ull;
            for (int w = 0; w < SUB_WINDOWS; w++) {
                const uint64_t base = p0 + 64ull * (uint64_t)w;
                if (base + 192 > lim) break;
                sb_seek_abs(b, base);
                if (b.bp == SB_NEVER) break;                  // (out of this wave's reach: no entry here)
                sb_stage(S, b);
                const uint32_t p = b.bp + (uint32_t)lane;     // the stage is addressed by relative positions
                const uint32_t wd = p >> 5, sh = p & 31u;
                const uint32_t d0 = S.stage[wd & STAGE_MASK], d1 = S.stage[(wd + 1) & STAGE_MASK], d2 = S.stage[(wd + 2) & STAGE_MASK];
                uint64_t bits = (((uint64_t)d1 << 32) | d0) >> sh;
                if (sh) bits |= (uint64_t)d2 << (64 - sh);
                const uint32_t step = lane_item_bits(S, bits, lit_maxlen, dist_maxlen, cx.dist_usable);
                uint32_t e = step ? (uint32_t)lane + step : 255u;              // < 64: the chain goes on at that lane; 255: dead
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((e & 63u) << 2), (int)e);
                    if (e < 64u) e = t;
                }
                __builtin_amdgcn_wave_barrier();
                flag[lane] = 0u;
                __builtin_amdgcn_wave_barrier();
                if (((live >> lane) & 1ull) && e != 255u) flag[e - 64u] = 1u; // (an item has at most 48 bits: e - 64 < 48)
                __builtin_amdgcn_wave_barrier();
                live = __ballot(flag[lane] != 0u);
                if (!live) break;
                if (__popcll(live) == 1) { found = base + 64ull + (uint64_t)(__ffsll((long long)live) - 1); break; }
            }
        }
        if (lane == 0) entry[i] = found;
    }
}

// B: every listed chunk from its entry to the next entry (or the end of the stream).  An entry is a block's start
// (hdr[c] = ~0) or an item's first bit inside the block whose header is at hdr[c] (subsync_kernel).
__global__ __launch_bounds__(64) void inflate_kernel(const uint8_t *in, uint64_t in_n, const uint64_t *start, const uint64_t *stop, const uint64_t *hdr,
                                                     uint32_t n_chunks, uint16_t *sym, const uint64_t *sym_off, const uint64_t *sym_cap,
                                                     uint64_t *out_len, uint64_t *end_bit, int *status, const uint32_t *only, uint32_t max_over)
{
    __shared__ WaveState S;
    const uint32_t c = only ? only[blockIdx.x] : blockIdx.x;      // (a second pass inflates the changed chunks only)
    if (c >= n_chunks) return;
    const int lane = threadIdx.x & 63;
#ifdef SS_GZ_TIMING
    const uint64_t t_wave0 = clock64();
#endif
    SBits b;
    const uint64_t s0 = start[c];
    const bool fresh = (s0 >> 63) != 0;                     // the first chunk of a member: nothing lies in front of it
    const uint64_t first_bit = s0 & ~(1ull << 63);
    sb_init(b, in, in_n, first_bit, hdr[c]);                 // (hdr[c]: ~0, or the header of the block the chunk begins in -- in front of it)
    OutState o{sym ? sym + sym_off[c] : nullptr, (uint32_t)min(sym_cap[c], (uint64_t)0x7FFF0000u), 0, 0};
    // A block that ends BEHIND the next chunk's entry: that entry was no block's start (a position inside a block passes the
    // sync search about once in a million candidates).  The chunk simply goes on to the entry after it -- its symbol
    // region (12x its input) has room for two or three chunks of FASTQ -- and says how many entries it ran over; within
    // a block it gives up beyond the second entry ahead (what itself started at a wrong entry decodes garbage).
    // An entry INSIDE a block (hdr != ~0) is met when this chunk, in the block with that header, arrives at an item's first
    // bit exactly there; if it passes it, or is in another block, the entry was wrong and is run over like the others.
    const uint64_t stop0 = stop[c];
    uint64_t stop_at = stop0;
    const uint64_t hard = stop[min(c + max_over, n_chunks - 1)];
    uint32_t skipped = 0;
    int st = 0;
    uint64_t block_at = hdr[c];                             // the header of the block the position is in
    bool resume = block_at != ~0ull;
    for (;;) {
        BlockCtx cx;
        int r;
        if (resume) {                                        // a chunk that begins inside a block: its tables first
            sb_seek_abs(b, block_at);
            OutState none{nullptr, 0, 0, 0};
            r = block_begin(S, b, none, cx);
            if (r != 0 || first_bit < sb_bitpos(b)) { st = -22; break; }
            sb_seek_abs(b, first_bit);
            resume = false;
        } else {
            block_at = sb_bitpos(b);
            r = block_begin(S, b, o, cx);
        }
        if (r == 0) {
            for (;;) {
                const uint32_t k = c + 1 + skipped;
                const uint64_t mid = (k < n_chunks && skipped <= max_over && hdr[k] == block_at) ? (start[k] & ~(1ull << 63)) : ~0ull;
                r = block_body(S, b, o, fresh, cx, ~0ull, sb_rel(b, hard), sb_rel(b, mid));
                if (r == 4 && skipped < max_over) { stop_at = stop[c + ++skipped]; continue; }      // not an item's first bit: a wrong entry, run over
                break;
            }
            if (r == 4) { st = -21; break; }
            if (r == 3) { st = 0; stop_at = sb_bitpos(b); break; }              // exactly at the next (not skipped) entry
        } else if (r >= 10) {
            r -= 10;
        }
        if (r < 0) { st = (r == -9 && sb_bitpos(b) > stop0) ? -21 : r; break; }      // (no room for more than its own: the host merges the two)
        const uint64_t pos = sb_bitpos(b);
        // (an entry inside a block cannot lie at a block's end -- it claims a header in front of it --: wrong, run over as well)
        while (skipped < max_over && c + skipped + 1 < n_chunks && (pos > stop_at || (pos == stop_at && hdr[c + 1 + skipped] != ~0ull)))
            stop_at = stop[c + ++skipped];
        if (r == 1) { st = (stop_at == ~0ull) ? 1 : -20; break; }      // the final block ends the LAST chunk of a member only
        if (pos == stop_at) {
            const uint32_t k = c + 1 + skipped;
            st = (k < n_chunks && hdr[k] != ~0ull) ? -21 : 0;
            break;
        }
        if (pos > stop_at) { st = -21; break; }
    }
    if (st >= 0 && o.out) out_flush(S, o, o.n);
    const uint64_t n = o.n;
    if (lane == 0) { out_len[c] = n; end_bit[c] = sb_bitpos(b); status[c] = st >= 0 ? st + 16 * (int)skipped : st; }
#ifdef SS_GZ_TIMING
    if (lane == 0) {
        for (int i = 0; i < 10; i++) atomicAdd(&g_gz_t[i], (unsigned long long)o.t[i]);
        atomicAdd(&g_gz_t[10], (unsigned long long)(clock64() - t_wave0));
        atomicAdd(&g_gz_t[11], 1ull);
    }
#endif
}

// C: windows.  The 32 KB in front of chunk c + 1 are the last 32 KB of chunk c's output, in which a symbol may still
//    point into chunk c's own window, and so on down the chain.  The tails are treated as maps "window of c -> window of
//    c + 1" (an entry is a byte or an index into the previous window) and composed (below).
__global__ __launch_bounds__(256) void tails_kernel(const uint16_t *sym, const uint64_t *sym_off, const uint64_t *out_len,
                                                    uint32_t n_chunks, uint16_t *map)
{
    const uint32_t c = blockIdx.y;                            // map[c] : window of chunk c -> window of chunk c + 1
    const uint64_t L = out_len[c];
    const uint16_t *sy = sym + sym_off[c];
    const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * 8;      // eight entries (16 bytes) per lane
    uint16_t e[8];
    if (L >= WSIZE - i0) {
        __builtin_memcpy(e, sy + (L - (WSIZE - i0)), 16);          // 2-byte aligned: an unaligned 16-byte load
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t i = i0 + k;
            e[k] = L >= WSIZE - i ? sy[L - (WSIZE - i)] : (uint16_t)(UNRES | (i + (uint32_t)L));      // the chunk was shorter than
        }                                                                                           // the window: its own window shifts in
    }
    __builtin_memcpy(map + (uint64_t)c * WSIZE + i0, e, 16);
}
// Two levels instead of log2(n_chunks) doubling rounds over all maps (every chunk of a FASTQ file hands lines like
// "+\n" down from the chunk before, so the chains are as long as the file and no round finishes early: twelve passes
// over 260 MB for a 130 MB file).  Chunks are grouped, GROUP consecutive ones:
//   1  one workgroup per group composes its maps IN ORDER, the running map (window at the group's start -> window behind
//      chunk c) in LDS: a 32 K-entry gather per chunk, the next chunk's tail prefetched meanwhile;
//   2  one workgroup walks the groups: the window at the start of group g + 1 = the last map of group g applied to the
//      window at the start of g (32 KB of bytes in LDS);
//   3  every chunk's window = its composed map applied to its group's start window, all in parallel.
__global__ __launch_bounds__(1024) void group_maps_kernel(const uint16_t *tails, uint32_t n_chunks, uint32_t group, uint16_t *maps)
{
    __shared__ uint16_t cur[WSIZE];                        // 64 KB
    const uint32_t c0 = blockIdx.x * group, c1 = min(c0 + group, n_chunks);
    const uint32_t tid = threadIdx.x;
    uint4 nx[4], v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) nx[k] = *reinterpret_cast<const uint4 *>(tails + (uint64_t)c0 * WSIZE + ((uint32_t)k * 1024 + tid) * 8);
    for (uint32_t c = c0; c < c1; c++) {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = nx[k];
        if (c + 1 < c1) {
#pragma unroll
            for (int k = 0; k < 4; k++) nx[k] = *reinterpret_cast<const uint4 *>(tails + (uint64_t)(c + 1) * WSIZE + ((uint32_t)k * 1024 + tid) * 8);
        }
        if (c > c0) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((v[k].x | v[k].y | v[k].z | v[k].w) & 0x80008000u) {
                    uint16_t e[8];
                    __builtin_memcpy(e, &v[k], 16);
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (e[j] & UNRES) e[j] = cur[e[j] & (WSIZE - 1)];
                    __builtin_memcpy(&v[k], e, 16);
                }
            }
        }
        __syncthreads();                                   // all gathers from the running map are done
#pragma unroll
        for (int k = 0; k < 4; k++) {
            *reinterpret_cast<uint4 *>(&cur[((uint32_t)k * 1024 + tid) * 8]) = v[k];
            *reinterpret_cast<uint4 *>(maps + (uint64_t)c * WSIZE + ((uint32_t)k * 1024 + tid) * 8) = v[k];
        }
        __syncthreads();
    }
}
// gwin[g] = the window in front of the first chunk of group g (bytes)
__global__ __launch_bounds__(1024) void group_windows_kernel(const uint16_t *maps, uint32_t n_chunks, uint32_t group, uint32_t n_groups,
                                                             const uint8_t *w0, uint8_t *gwin)
{
    __shared__ uint8_t w[WSIZE];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < WSIZE; i += 1024) w[i] = w0 ? w0[i] : (uint8_t)0;     // the 32 KB in front of the first chunk: the end of
                                                                                  // the text so far (a file's first chunk: nothing, never read)
    __syncthreads();
    for (uint32_t g = 0; g < n_groups; g++) {
        for (uint32_t i = tid; i < WSIZE; i += 1024) gwin[(uint64_t)g * WSIZE + i] = w[i];
        if (g + 1 == n_groups) break;
        const uint16_t *m = maps + (uint64_t)(min((g + 1) * group, n_chunks) - 1) * WSIZE;      // the group's last composed map
        uint8_t nw[32];
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const uint16_t e = m[(uint32_t)k * 1024 + tid];
            nw[k] = (e & UNRES) ? w[e & (WSIZE - 1)] : (uint8_t)e;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; k++) w[(uint32_t)k * 1024 + tid] = nw[k];
        __syncthreads();
    }
}
// win[c + 1] = bytes of the composed map of chunk c over its group's start window; win[0] = the window in front of the segment
__global__ __launch_bounds__(256) void windows_kernel(const uint16_t *maps, uint32_t n_chunks, uint32_t group, const uint8_t *gwin, uint8_t *win)
{
    const uint32_t c = blockIdx.y;
    const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    uint2 o = *reinterpret_cast<const uint2 *>(gwin + i0);      // the first chunk: what lies in front of the segment (group 0's start)
    if (c) {
        const uint4 v = *reinterpret_cast<const uint4 *>(maps + (uint64_t)(c - 1) * WSIZE + i0);
        if ((v.x | v.y | v.z | v.w) & 0x80008000u) {
            const uint8_t *gw = gwin + (uint64_t)((c - 1) / group) * WSIZE;
            uint16_t e[8];
            uint8_t by[8];
            __builtin_memcpy(e, &v, 16);
#pragma unroll
            for (int j = 0; j < 8; j++) by[j] = (e[j] & UNRES) ? gw[e[j] & (WSIZE - 1)] : (uint8_t)e[j];
            __builtin_memcpy(&o, by, 8);
        } else {
            o.x = (v.x & 0xFFu) | ((v.x >> 8) & 0xFF00u) | ((v.y & 0xFFu) << 16) | ((v.y << 8) & 0xFF000000u);
            o.y = (v.z & 0xFFu) | ((v.z >> 8) & 0xFF00u) | ((v.w & 0xFFu) << 16) | ((v.w << 8) & 0xFF000000u);
        }
    }
    *reinterpret_cast<uint2 *>(win + (uint64_t)c * WSIZE + i0) = o;
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

// the last 32 KB of the text so far = the window in front of the next segment's first chunk
__global__ __launch_bounds__(256) void lastwin_kernel(const uint8_t *text, uint64_t total, uint8_t *prev)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < WSIZE) prev[i] = total >= WSIZE - i ? text[total - (WSIZE - i)] : (uint8_t)0;
}

// CRC-32 (zlib's polynomial) of the text's segments [seg_at[s], seg_at[s] + seg_len[s]), one lane per segment, byte-wise
// table in LDS (the segments of a member start at its first byte: any alignment)
__global__ __launch_bounds__(64) void crc_kernel(const uint8_t *text, const uint64_t *seg_at, const uint32_t *seg_len, uint64_t n_seg,
                                                 const uint32_t *table, uint32_t *crc)
{
    __shared__ uint32_t tab[256];
    for (int i = threadIdx.x; i < 256; i += 64) tab[i] = table[i];
    __syncthreads();
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    uint64_t i = seg_at[s];
    const uint64_t e = i + seg_len[s];
    uint32_t k = 0xFFFFFFFFu;
    for (; i < e && (i & 3); i++) k = tab[(k ^ text[i]) & 0xFFu] ^ (k >> 8);
    for (; i + 4 <= e; i += 4) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(text + i);
        k = tab[(k ^ w) & 0xFFu] ^ (k >> 8);
        k = tab[(k ^ (w >> 8)) & 0xFFu] ^ (k >> 8);
        k = tab[(k ^ (w >> 16)) & 0xFFu] ^ (k >> 8);
        k = tab[(k ^ (w >> 24)) & 0xFFu] ^ (k >> 8);
    }
    for (; i < e; i++) k = tab[(k ^ text[i]) & 0xFFu] ^ (k >> 8);
    crc[s] = k ^ 0xFFFFFFFFu;
}

// The same for the members of a file, segments of 2^lg bytes from every member's first byte: segment s belongs to the member
// m with first[m] <= s < first[m + 1] (binary search: a bgzip file has thousands) and begins (s - first[m]) << lg bytes into it.
// Nothing but the member table travels to the device (the list of segments was 12 bytes per 4 KB of text, and three
// pageable copies in front of the kernel); the byte table is made here.
struct CrcMember { uint64_t at, len, first; };
__global__ __launch_bounds__(64) void crc_members_kernel(const uint8_t *text, const CrcMember *mem, uint32_t n_mem, uint64_t n_seg, int lg, uint32_t *crc)
{
    __shared__ uint32_t tab[256];
    for (int i = threadIdx.x; i < 256; i += 64) {
        uint32_t k = (uint32_t)i;
        for (int j = 0; j < 8; j++) k = (k & 1u) ? 0xEDB88320u ^ (k >> 1) : k >> 1;
        tab[i] = k;
    }
    __syncthreads();
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    uint32_t lo = 0, hi = n_mem - 1;                         // the last member whose first segment is <= s
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (mem[mid].first <= s) lo = mid; else hi = mid - 1;
    }
    const uint64_t a0 = (s - mem[lo].first) << lg;
    uint64_t i = mem[lo].at + a0;
    const uint64_t e = i + min((uint64_t)1 << lg, mem[lo].len - a0);
    uint32_t k = 0xFFFFFFFFu;
    for (; i < e && (i & 3); i++) k = tab[(k ^ text[i]) & 0xFFu] ^ (k >> 8);
    for (; i + 4 <= e; i += 4) {
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
// newlines of text[0, n): 16 bytes per lane and step (SWAR zero-byte test on w ^ 0x0A0A0A0A), the ragged ends byte by byte
__global__ __launch_bounds__(256) void count_nl_kernel(const uint8_t *text, uint64_t n, unsigned long long *out)
{
    unsigned long long c = 0;
    const uint64_t head = std::min<uint64_t>(n, (16 - ((uintptr_t)text & 15)) & 15), body = (n - head) / 16;
    const uint4 *t4 = reinterpret_cast<const uint4 *>(text + head);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < body; i += (uint64_t)gridDim.x * 256) {
        const uint4 v = t4[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t x = w[d] ^ 0x0A0A0A0Au;
            c += (unsigned)__popc(~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u);
        }
    }
    if (blockIdx.x == 0) {
        for (uint64_t i = threadIdx.x; i < head; i += 256) c += text[i] == '\n' ? 1u : 0u;
        for (uint64_t i = head + body * 16 + threadIdx.x; i < n; i += 256) c += text[i] == '\n' ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
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

// bgzip (BGZF): every member carries its size in an extra field ("BC", 2 bytes: BSIZE = size - 1), so the members -- at most
// 64 KB of text each, nothing in front of any -- are found without decoding.  -> (first data byte, trailer byte) of every
// member, or nothing when the file is not BGZF throughout
struct Bgzf { uint64_t data, trailer; };
std::vector<Bgzf> bgzf_members(const uint8_t *p, uint64_t n)
{
    std::vector<Bgzf> out;
    uint64_t pos = 0;
    while (pos < n) {
        if (pos + 18 > n || p[pos] != 0x1f || p[pos + 1] != 0x8b || p[pos + 2] != 8 || p[pos + 3] != 4) return {};
        const uint64_t xlen = (uint64_t)p[pos + 10] | (uint64_t)p[pos + 11] << 8;
        if (xlen < 6 || pos + 12 + xlen > n) return {};
        uint64_t bsize = 0;
        for (uint64_t q = pos + 12; q + 4 <= pos + 12 + xlen;) {
            const uint64_t slen = (uint64_t)p[q + 2] | (uint64_t)p[q + 3] << 8;
            if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= pos + 12 + xlen) bsize = ((uint64_t)p[q + 4] | (uint64_t)p[q + 5] << 8) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 2 + 8 || pos + bsize > n) return {};
        out.push_back(Bgzf{pos + 12 + xlen, pos + bsize - 8});
        pos += bsize;
    }
    return out;
}

// Scratch of one call, kept for the next one: the per-segment symbol streams, window maps and chunk arrays.  (Allocating
// and freeing tens of GB per file is what this replaces: on this platform a large hipMalloc that follows a large hipFree
// can take 1.6 s -- scripts/archive/dev/t_bigalloc.py -- and a 1 GB .gz needed 28 GB of symbols in one piece.)  At most two are
// kept (the two mates of a pair are inflated concurrently), ~4.5 GB each, until the process ends.
struct Arena {
    uint64_t cap_chunks = 0, sym_elems = 0;
    uint16_t *sym = nullptr, *map[2] = {nullptr, nullptr};
    uint64_t *meta = nullptr;
    int *status = nullptr;
    uint8_t *win = nullptr, *gwin = nullptr, *prev = nullptr;
    uint32_t *todo = nullptr;
    uint8_t *text = nullptr;              // the text of the call that holds the arena (lent to the caller until gpu_gunzip_done)
    uint64_t text_cap = 0;
    void *late_free[4] = {nullptr, nullptr, nullptr, nullptr};      // the call's stage and lists: given back when the text is (the
                                                                     // frees took ~1 ms beside another thread's allocations, in front of the caller's work)
};
std::mutex g_arena_mu;
std::vector<Arena *> g_arena_free;
void arena_destroy(Arena *a)
{
    if (!a) return;
    void *q[] = {a->sym, a->map[0], a->map[1], a->meta, a->status, a->win, a->gwin, a->prev, a->todo, a->text};
    for (void *x : q) if (x) hipFree(x);
    for (void *x : a->late_free) if (x) hipFreeAsync(x, nullptr);
    delete a;
}
Arena *arena_get(uint64_t cap_chunks, uint64_t sym_elems)
{
    Arena *a = nullptr;
    {
        std::lock_guard<std::mutex> g(g_arena_mu);
        if (!g_arena_free.empty()) { a = g_arena_free.back(); g_arena_free.pop_back(); }
    }
    if (a && a->cap_chunks >= cap_chunks && a->sym_elems >= sym_elems) return a;
    uint8_t *text = nullptr;
    uint64_t text_cap = 0;
    if (a) { text = a->text; text_cap = a->text_cap; a->text = nullptr; }      // (the text buffer moves to the new arena)
    arena_destroy(a);
    a = new (std::nothrow) Arena();
    if (a) { a->text = text; a->text_cap = text_cap; } else if (text) hipFree(text);
    if (!a) return nullptr;
    a->cap_chunks = cap_chunks;
    a->sym_elems = sym_elems;
    const uint64_t n_groups = cap_chunks / 8 + 2;
    const bool ok = hipMalloc((void **)&a->sym, sym_elems * 2) == hipSuccess && hipMalloc((void **)&a->map[0], cap_chunks * WSIZE * 2) == hipSuccess &&
                    hipMalloc((void **)&a->map[1], cap_chunks * WSIZE * 2) == hipSuccess && hipMalloc((void **)&a->meta, cap_chunks * 8 * 8) == hipSuccess &&
                    hipMalloc((void **)&a->status, cap_chunks * 4) == hipSuccess && hipMalloc((void **)&a->win, cap_chunks * WSIZE) == hipSuccess &&
                    hipMalloc((void **)&a->gwin, n_groups * WSIZE) == hipSuccess && hipMalloc((void **)&a->prev, WSIZE) == hipSuccess &&
                    hipMalloc((void **)&a->todo, cap_chunks * 4) == hipSuccess;
    if (!ok) { arena_destroy(a); return nullptr; }
    return a;
}
// a waiting arena that needs no allocation for this call (then nothing has to ask the driver how much memory is left: 1-2 ms)
Arena *arena_take_if_fits(uint64_t cap_chunks, uint64_t sym_elems, uint64_t text_cap)
{
    std::lock_guard<std::mutex> g(g_arena_mu);
    for (size_t i = g_arena_free.size(); i-- > 0;) {
        Arena *a = g_arena_free[i];
        if (a->cap_chunks >= cap_chunks && a->sym_elems >= sym_elems && a->text_cap >= text_cap) {
            g_arena_free.erase(g_arena_free.begin() + (long)i);
            return a;
        }
    }
    return nullptr;
}
void arena_put(Arena *a)
{
    if (!a) return;
    if (a->text_cap > (6ull << 30)) { ss::big_put(a->text, a->text_cap); a->text = nullptr; a->text_cap = 0; }      // (not kept here: the text of a very large file
                                                                                                                      //  becomes a slab of the read set, ss::big_take)
    {
        std::lock_guard<std::mutex> g(g_arena_mu);
        if (g_arena_free.size() < 2) { g_arena_free.push_back(a); return; }
    }
    arena_destroy(a);
}

// A large file image on its way to the device: copied out of a mapping of the page cache the runtime faults the pages in
// one by one (8 GB/s); eight threads that pread() blocks of a few MB into pinned buffers of their own and ship them on streams
// of their own do ~25 GB/s.  The pinned buffers (128 MB a set, at most two sets) are kept like the arenas.
constexpr uint64_t PIN_BYTES = 16ull << 20;
constexpr int PIN_N = 8;      // threads = buffers of a set (128 MB a set); each with a stream and two events of its own, made
                              // with the set (a stream costs 2-3 ms to make: as much as the upload of 30 MB)
struct PinSet
{
    uint8_t *b[PIN_N] = {};
    hipStream_t s[PIN_N] = {};
    hipEvent_t ev[PIN_N][2] = {};
};
std::vector<PinSet *> g_pin_free;
void pin_destroy(PinSet *p)
{
    if (!p) return;
    for (int i = 0; i < PIN_N; i++) {
        if (p->s[i]) { hipStreamSynchronize(p->s[i]); hipStreamDestroy(p->s[i]); }
        for (int q = 0; q < 2; q++) if (p->ev[i][q]) hipEventDestroy(p->ev[i][q]);
        if (p->b[i]) hipHostFree(p->b[i]);
    }
    delete p;
}
PinSet *pin_get()
{
    {
        std::lock_guard<std::mutex> g(g_arena_mu);
        if (!g_pin_free.empty()) { PinSet *p = g_pin_free.back(); g_pin_free.pop_back(); return p; }
    }
    PinSet *p = new (std::nothrow) PinSet();
    if (!p) return nullptr;
    int device = 0;
    hipGetDevice(&device);
    std::atomic<int> bad(0);
    std::vector<std::thread> th;                              // (~10 ms a buffer, ~3-13 ms a stream: all at the same time)
    for (int i = 0; i < PIN_N; i++)
        th.emplace_back([p, i, device, &bad] {
            if (hipSetDevice(device) != hipSuccess || hipHostMalloc((void **)&p->b[i], PIN_BYTES, hipHostMallocDefault) != hipSuccess ||
                hipStreamCreateWithFlags(&p->s[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&p->ev[i][0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&p->ev[i][1], hipEventDisableTiming) != hipSuccess)
                bad = 1;
        });
    for (auto &t : th) t.join();
    if (bad) { pin_destroy(p); return nullptr; }
    return p;
}
bool pin_waiting()
{
    std::lock_guard<std::mutex> g(g_arena_mu);
    return !g_pin_free.empty();
}
void pin_put(PinSet *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_arena_mu);
        if (g_pin_free.size() < 2) { g_pin_free.push_back(p); return; }
    }
    pin_destroy(p);
}
// `ready(bytes)`: called (serialised, from the upload threads) whenever the PREFIX of the image that has arrived on the device
// has grown -- the caller starts work on it while the rest is still on its way.
bool upload_file(int fd, uint64_t n, uint8_t *d_in, const std::function<void(uint64_t)> &ready = nullptr)
{
    static const bool trace = getenv("SS_INGEST_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() * 1e3; };
    // one file at a time on the link: the files of a call would otherwise arrive together, late, and their searches start together;
    // in turn the first one is searched while the second one travels
    static std::mutex link;
    constexpr bool turns = true;
    std::unique_lock<std::mutex> my_turn(link, std::defer_lock);
    if (turns) my_turn.lock();
    PinSet *pins = pin_get();
    if (!pins) return false;
    const double t_pins = since();
    std::atomic<int> streams_up(0);
    double t_streams = 0, t_first = 0;
    int device = 0;
    hipGetDevice(&device);
    std::atomic<uint64_t> next(0);
    std::atomic<int> failed(0);
    // every thread's pinned buffer is used in two halves: the copy of one block travels while the next is read.  Blocks
    // of 1/16 of the file (2 MB .. half a buffer), handed out in order, so that the prefix grows steadily
    const uint64_t half = PIN_BYTES / 2;
    constexpr uint64_t div = 16;
    const uint64_t blk = std::min<uint64_t>(half, std::max<uint64_t>(2ull << 20, ((n / div) + (1ull << 20) - 1) & ~((1ull << 20) - 1)));
    const uint64_t n_blocks = (n + blk - 1) / blk;
    std::vector<uint8_t> done((size_t)n_blocks, 0);
    std::mutex mu;
    uint64_t prefix = 0;                                     // blocks [0, prefix) are on the device
    auto finished = [&](uint64_t b) {
        std::lock_guard<std::mutex> g(mu);
        done[(size_t)b] = 1;
        const uint64_t before = prefix;
        while (prefix < n_blocks && done[(size_t)prefix]) prefix++;
        if (trace && before == 0 && prefix) t_first = since();
        if (prefix != before && ready) ready(std::min(n, prefix * blk));
    };
    std::vector<std::thread> pool;
    const int n_threads = PIN_N;
    for (int t = 0; t < n_threads; t++)
        pool.emplace_back([&, t] {
            hipStream_t s2 = pins->s[t];
            hipEvent_t *ev = pins->ev[t];
            if (hipSetDevice(device) != hipSuccess) failed = 1;
            if (trace && streams_up.fetch_add(1) + 1 == n_threads) t_streams = since();
            int64_t in_flight[2] = {-1, -1};                  // the block whose copy was last issued from each half
            int h = 0;
            for (uint64_t b; !failed && (b = next.fetch_add(1)) < n_blocks; h ^= 1) {
                if (in_flight[h] >= 0) {                      // this half's earlier copy must have left it
                    if (hipEventSynchronize(ev[h]) != hipSuccess) { failed = 1; break; }
                    finished((uint64_t)in_flight[h]);
                    in_flight[h] = -1;
                }
                if (in_flight[h ^ 1] >= 0 && hipEventQuery(ev[h ^ 1]) == hipSuccess) {      // (the other half's: reported as soon as it is seen,
                    finished((uint64_t)in_flight[h ^ 1]);                                     //  the caller works on the prefix that has arrived)
                    in_flight[h ^ 1] = -1;
                }
                const uint64_t a = b * blk, len = std::min<uint64_t>(blk, n - a);
                uint8_t *buf = pins->b[t] + (uint64_t)h * half;
                uint64_t got = 0;
                while (got < len) {
                    const ssize_t r = pread(fd, buf + got, len - got, (off_t)(a + got));
                    if (r <= 0) break;
                    got += (uint64_t)r;
                }
                if (got != len || hipMemcpyAsync(d_in + a, buf, len, hipMemcpyHostToDevice, s2) != hipSuccess ||
                    hipEventRecord(ev[h], s2) != hipSuccess) { failed = 1; break; }
                in_flight[h] = (int64_t)b;
            }
            for (int q = 0; q < 2; q++) {
                const int hh = (h + q) & 1;                   // the older copy first
                if (in_flight[hh] >= 0 && !failed) {
                    if (hipEventSynchronize(ev[hh]) != hipSuccess) failed = 1;
                    else finished((uint64_t)in_flight[hh]);
                }
            }
            hipStreamSynchronize(s2);                       // (a failure may leave a copy in flight: the buffers go back to the pool)
        });
    for (auto &th : pool) th.join();
    pin_put(pins);
    if (trace)
        fprintf(stderr, "[ginflate] upload: %llu blocks of %.1f MB; buffers %.2f ms, threads up %.2f, first block %.2f, all %.2f\n", (unsigned long long)n_blocks,
                blk / 1048576.0, t_pins, t_streams, t_first, since());
    return !failed;
}

}  // namespace

namespace ss {

// (the same way in for other whole files: ss_l2_import's cluster images)
bool upload_file_to_device(int fd, uint64_t n, uint8_t *d_dst) { return upload_file(fd, n, d_dst); }

// The file image `in` (host) inflated on the device.  true: *text_dev holds *len bytes, every member verified against its
// trailer; the buffer belongs to the scratch arena of the call and is lent until gpu_gunzip_done(*lease).  false: not
// handled here (the caller inflates on the host).
static std::atomic<uint64_t> g_handled{0}, g_declined{0}, g_range_files{0}, g_range_pieces{0};
// ss_test_hook: nothing in the environment can switch these on
std::atomic<long long> g_hook_entry{0}, g_hook_decline{0}, g_hook_skip_chain{0};

void gpu_gunzip_done(void *lease)
{
    Arena *a = static_cast<Arena *>(lease);
    if (a)
        for (void *&q : a->late_free) {
            if (q) hipFreeAsync(q, nullptr);                 // (last used on the call's stream, which was synchronised before the text was handed out)
            q = nullptr;
        }
    arena_put(a);
}

// The streams of the calls, kept (making one and destroying it was 0.6 ms of every call), and the stream-ordered allocator
// told to keep what a call frees (1 GB) instead of handing it back to the driver at the next synchronisation.
static std::vector<hipStream_t> g_stream_free;
hipStream_t call_stream_get()
{
    static std::once_flag once;
    std::call_once(once, [] { ss::pool_keep_at_least(1024ull << 20); });
    {
        std::lock_guard<std::mutex> g(g_arena_mu);
        if (!g_stream_free.empty()) { hipStream_t s = g_stream_free.back(); g_stream_free.pop_back(); return s; }
    }
    hipStream_t s = nullptr;
    return hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess ? s : nullptr;
}
void call_stream_put(hipStream_t s)                          // (synchronised by the caller)
{
    if (!s) return;
    {
        std::lock_guard<std::mutex> g(g_arena_mu);
        if (g_stream_free.size() < 4) { g_stream_free.push_back(s); return; }
    }
    hipStreamDestroy(s);
}

// ---- several ranks share ONE gzip member (ss_gz_set_range) -------------------------------------------------------------------
// The deflate data is cut into slices of `slice_chunks` search chunks; slice s belongs to rank s mod world.  A rank looks
// for block starts in its slices only, inflates their chunks to symbols -- the expensive part, all ranks at once -- and
// then needs what lies in front of each slice: the last 32 KB of the text before it.  That comes from the owner of the
// slice before, down a CHAIN of small messages (the caller's callback: point-to-point send / receive between ranks):
//     window     the 32 KB in front of the next slice
//     end_bit    where this slice's last chunk ended: the next slice's first chunk must start exactly there
//     nl         newlines so far: the next rank knows which of the four FASTQ lines its text starts in
//     carry      the bytes behind the last complete record (its rest is in the next slice)
//     crc, len   CRC-32 and length of the text so far (the owner of the last slice checks them against the trailer)
//     status     < 0: somebody declined -- passed on to the end of the chain, everybody declines
// A rank that fails at any point still takes part in the chain for all its slices (receives, passes the bad news on), so
// nobody waits for a message that never comes.  Several members (lanes joined with cat) are followed inside the slices as in
// the whole-file path -- a member that ends in a slice is checked against its trailer by the slice's owner, crc / len
// always describe the member that is open at the cut.  bgzip: a slice's chunks are the members that begin in its byte range
// (no search, no window needed); the chain carries the newline count and the straddling record all the same.
constexpr uint32_t CARRY_MAX = 65536;
struct ChainMsg {
    int32_t status;
    uint32_t crc;
    uint64_t len, nl, end_bit;
    uint32_t carry_len, pad;
    uint8_t window[WSIZE];
    uint8_t carry[CARRY_MAX];
};
// the order in which the files of a load use the chain (ss_common.h: gz_range_ticket)
static std::mutex g_turn_mu;
static std::condition_variable g_turn_cv;
static uint64_t g_turn_issued = 0, g_turn_serving = 0;
uint64_t gz_range_ticket()
{
    std::lock_guard<std::mutex> g(g_turn_mu);
    return g_turn_issued++;
}
static void gz_range_wait(uint64_t ticket)                   // (returns at once for the holder of the turn and for tickets already passed)
{
    std::unique_lock<std::mutex> g(g_turn_mu);
    g_turn_cv.wait(g, [&] { return g_turn_serving >= ticket; });
}
void gz_range_pass(uint64_t ticket)
{
    std::unique_lock<std::mutex> g(g_turn_mu);
    g_turn_cv.wait(g, [&] { return g_turn_serving >= ticket; });
    if (g_turn_serving == ticket) { g_turn_serving++; g_turn_cv.notify_all(); }
}

struct RangeRun {
    uint32_t rank = 0, world = 1, slice_chunks = 0, n_slices = 0;
    ss_gz_chain_fn fn = nullptr;
    void *user = nullptr;
    std::vector<uint32_t> mine;           // my slices, ascending
    size_t duty = 0;                      // mine[duty]: the first slice whose part in the chain is not finished
    bool received = false;                // ... and whether its message has come in
    bool failed = false;                  // a callback failed: this rank declines, status -1 still goes down the chain
    bool broken = false;                  // a SEND failed: no more chain traffic
    bool inject_decline = false;          // test hook
    ChainMsg msg;                         // the last message received / the one being sent
    std::vector<GzPiece> *pieces = nullptr;
    // A failed callback (the caller's point-to-point call raised, or its bounded wait ran out: dist._gz_chain) does not end
    // this rank's part: nothing more is RECEIVED (the outcome is known: declined), but status -1 still goes down the chain
    // for every slice of this rank, so that the ranks behind it learn it at once instead of each waiting for its own
    // deadline; only a failed SEND ends the traffic (the peer is gone).
    uint64_t ticket = 0;                  // this file's place in the order in which the files of a load use the chain
    bool recv_for(uint32_t s)
    {
        if (s == 0) return !failed;
        if (failed) return false;
        gz_range_wait(ticket);
        if (fn(&msg, sizeof(ChainMsg), (int)s, 0, user) != 0) { failed = true; msg.status = -1; msg.carry_len = 0; return false; }
        received = true;
        return true;
    }
    bool send_from(uint32_t s)
    {
        if (s + 1 >= n_slices) return !failed;
        if (broken) return false;
        if (failed) { msg.status = -1; msg.carry_len = 0; }
        gz_range_wait(ticket);
        if (fn(&msg, sizeof(ChainMsg), (int)s, 1, user) != 0) { broken = true; failed = true; return false; }
        return !failed;
    }
    // whatever is left of this rank's part in the chain, as a bystander: receive, pass on that it went wrong
    void abort_chain()
    {
        for (; duty < mine.size() && !broken; duty++) {
            const uint32_t s = mine[duty];
            if (s > 0 && !received) recv_for(s);
            msg.status = -1;
            msg.carry_len = 0;
            send_from(s);
            received = false;
        }
    }
};
static std::mutex g_range_mu;            // (guards g_range)
static struct { int rank = 0, world = 1; uint64_t slice_bytes = 0; ss_gz_chain_fn fn = nullptr; void *user = nullptr; } g_range;

bool gz_range_active() { return g_range.world > 1 && g_range.fn != nullptr; }

static bool gpu_gunzip_impl(const uint8_t *in, uint64_t in_n, char **text_dev, uint64_t *len, void **lease, int fd, RangeRun *rr);

bool gpu_gunzip(const uint8_t *in, uint64_t in_n, char **text_dev, uint64_t *len, void **lease, int fd)
{
    return gpu_gunzip_impl(in, in_n, text_dev, len, lease, fd, nullptr);
}

// This rank's slices of the file: *text_dev holds their texts back to back, `pieces` says where each lies, what came down
// the chain in front of it (the bytes of a record that began in the slice before) and where its last complete record
// ends.  false: declined (the chain has been served all the same).
bool gpu_gunzip_range(const uint8_t *in, uint64_t in_n, char **text_dev, void **lease, int fd, std::vector<GzPiece> *pieces, uint64_t ticket, bool decline)
{
    RangeRun rr;
    {
        std::lock_guard<std::mutex> one(g_range_mu);
        if (!gz_range_active()) return false;
        rr.rank = (uint32_t)g_range.rank; rr.world = (uint32_t)g_range.world; rr.fn = g_range.fn; rr.user = g_range.user;
    }
    if (g_hook_skip_chain.load()) return false;               // test hook: a rank that leaves WITHOUT serving the chain (the peers' bounded wait)
    rr.ticket = ticket;
    rr.pieces = pieces;
    rr.inject_decline = decline;
    uint64_t n = 0;
    const bool ok = gpu_gunzip_impl(in, in_n, text_dev, &n, lease, fd, &rr);
    gz_range_pass(ticket);                                    // (this file's chain traffic is over, whatever came of it)
    return ok;
}

static bool gpu_gunzip_impl(const uint8_t *in, uint64_t in_n, char **text_dev, uint64_t *len, void **lease, int fd, RangeRun *rr)
{
    static const bool trace = getenv("SS_INGEST_TRACE") != nullptr;
    // own stream: the two mates of a paired sample are inflated by two host threads, and the legacy default stream would
    // serialise them
    hipStream_t st = call_stream_get();
    const bool have_stream = st != nullptr;                  // (checked once the slices are known: a rank that fails here still serves the chain)
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        hipStreamSynchronize(st);
        fprintf(stderr, "[ginflate] %-18s at %.4f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    uint8_t *d_in = nullptr, *d_text = nullptr;
    uint64_t *d_entry = nullptr;
    uint32_t *d_crc = nullptr, *d_tab = nullptr;
    Arena *A = nullptr;
    std::future<Arena *> ahead;                               // (a large file: the arena allocated while the image travels)
    auto cleanup = [&](bool keep_text) {
        if (ahead.valid()) arena_put(ahead.get());            // (a call that leaves before it took the arena over)
        auto now = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
        const double c0 = now();
        void *scratch[] = {d_in, d_entry, d_crc, d_tab};
        for (int q = 0; q < 4; q++) {
            if (!scratch[q]) continue;
            if (keep_text && A) A->late_free[q] = scratch[q];      // (freed by gpu_gunzip_done)
            else hipFreeAsync(scratch[q], st);
        }
        const double c1 = now();
        hipStreamSynchronize(st);
        const double c2 = now();
        call_stream_put(st);
        if (trace) fprintf(stderr, "[ginflate] cleanup: free %.4f, sync %.4f, stream back %.4f s\n", c1 - c0, c2 - c1, now() - c2);
        if (!keep_text) { arena_put(A); A = nullptr; }            // (else the caller holds it, with the text, until gpu_gunzip_done)
    };
    auto no = [&](const char *why, long long a = 0) {
        if (trace) fprintf(stderr, "[ginflate] not handled: %s (%lld)\n", why, a);
        if (rr) rr->abort_chain();                            // (the other ranks' chain goes on through this one)
        g_declined++;
        cleanup(false);
        return false;
    };
    auto h2d = [&](void *dst, const void *src, uint64_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st) == hipSuccess; };
    auto d2h = [&](void *dst, const void *src, uint64_t bytes) {
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    };
#define GI(call) do { if ((call) != hipSuccess) return no(#call); } while (0)
#define GB(call) do { if (!(call)) return no(#call); } while (0)
    // (a rank that only serves the chain may hold no more than the first 70 KB of the file: ss_fastq_dev.hip)
    const uint64_t data_off = gzip_header_len(in, rr && rr->inject_decline ? std::min<uint64_t>(in_n, 70u << 10) : in_n);
    if (!data_off) return no("header");
    // SS_GZ_SPLIT_KB: blocks are entered every so many KB of deflate data (subsync_kernel; 0 = at their starts only)
    uint64_t split_bytes = 12 << 10;
    if (const char *e = getenv("SS_GZ_SPLIT_KB")) split_bytes = (uint64_t)std::max<long long>(0, atoll(e)) << 10;
    uint64_t chunk_bytes = split_bytes ? 16 << 10 : 32 << 10, ratio = 12, seg_bytes = 128ull << 20;      // (split: every block's start should be found)
    if (const char *e = getenv("SS_GZ_CHUNK")) chunk_bytes = std::max<uint64_t>(4096, (uint64_t)atoll(e));
    if (const char *e = getenv("SS_GZ_SEG_KB")) seg_bytes = std::max<uint64_t>(64, (uint64_t)atoll(e)) << 10;      // (tests: many segments)
    const uint64_t data_n = in_n - 8 - data_off;
    seg_bytes = std::max(seg_bytes, 8 * chunk_bytes);
    constexpr size_t SEG_CHUNKS = 16384;                      // chunks of a segment at most
    const uint64_t n_chunks0_ = std::max<uint64_t>(1, (data_n + chunk_bytes - 1) / chunk_bytes);
    if (n_chunks0_ > 0x7FFFFFF0ull) return no("size");
    const uint32_t n_chunks0 = (uint32_t)n_chunks0_;
    constexpr uint32_t LOOK = 64;                             // search chunks behind a slice in which its last chunk's stop is looked for
    if (rr) {
        // the slices, before anything can fail: every rank derives the same ones from the file's size
        uint64_t slice_bytes = g_range.slice_bytes ? g_range.slice_bytes : std::min<uint64_t>(128ull << 20, std::max<uint64_t>(4ull << 20, data_n / (2ull * rr->world)));
        if (const char *e = getenv("SS_GZ_SLICE_KB")) slice_bytes = std::max<uint64_t>(64, (uint64_t)atoll(e)) << 10;      // (tests: many slices)
        rr->slice_chunks = (uint32_t)std::max<uint64_t>(2 * LOOK, slice_bytes / chunk_bytes);
        rr->n_slices = (n_chunks0 + rr->slice_chunks - 1) / rr->slice_chunks;
        rr->slice_chunks = (n_chunks0 + rr->n_slices - 1) / rr->n_slices;      // equal slices (no sliver at the end)
        rr->n_slices = (n_chunks0 + rr->slice_chunks - 1) / rr->slice_chunks;
        for (uint32_t sl = rr->rank; sl < rr->n_slices; sl += rr->world) rr->mine.push_back(sl);
        if (rr->n_slices < 2) return no("one slice");         // nothing to share out: the whole-file path
        if (rr->inject_decline) return no("declined on request (test hook, or the file could not be mapped)");
        if (trace) fprintf(stderr, "[ginflate] range mode: rank %u of %u, %u slices of %u chunks, %zu mine\n", rr->rank, rr->world, rr->n_slices,
                           rr->slice_chunks, rr->mine.size());
    }

    if (!have_stream) return no("hipStreamCreateWithFlags");
    auto text_guess = [&] {
        const uint8_t *t8 = in + in_n - 8;
        uint64_t guess = (uint64_t)t8[4] | (uint64_t)t8[5] << 8 | (uint64_t)t8[6] << 16 | (uint64_t)t8[7] << 24;      // ISIZE of the last member
        while (guess < in_n) guess += 1ull << 32;
        return (guess <= 16 * in_n ? guess : 3 * in_n) + 64;
    };
    // A LARGE file (round 5): its scratch arena and text buffer are allocated on a thread of their own WHILE the image travels --
    // sized by what a segment can need at most (SEG_CHUNKS chunks, seg_bytes of data) instead of what this file's largest one
    // does, which is known only after the search.  A fresh process is handed new device memory at ~25 GB/s: for a 3.4 GB file
    // (4 GB of symbols + 7.7 GB of text) that was 0.14-0.38 s of waiting between the search and the first segment.
    const bool big = fd >= 0 && !rr && in_n >= (256ull << 20);
    if (big) {
        int device = 0;
        hipGetDevice(&device);
        const uint64_t ub_chunks = SEG_CHUNKS + 80;
        const uint64_t ub_need = (seg_bytes + chunk_bytes + SEG_CHUNKS) * ratio + SEG_CHUNKS * 4096;
        const uint64_t ub_sym = ub_need + ub_need / 4 + 64 * (4096 + 64 * ratio), ub_text = text_guess();
        ahead = std::async(std::launch::async, [=]() -> Arena * {
            if (hipSetDevice(device) != hipSuccess) return nullptr;
            Arena *a = arena_take_if_fits(ub_chunks, ub_sym, ub_text);
            if (!a) {
                size_t mem_free = 0, mem_total = 0;
                if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) return nullptr;
                if (2 * ub_text + ub_sym * 2 + ub_chunks * WSIZE * 5 + (256ull << 20) + in_n > mem_free / 2) return nullptr;      // (decided below, with the real sizes)
                a = arena_get(ub_chunks, ub_sym);
            }
            if (a && a->text_cap < ub_text) {
                if (a->text) hipFree(a->text);
                a->text = nullptr;
                a->text_cap = 0;
                uint64_t got = ub_text;
                if (ss::big_malloc((void **)&a->text, ub_text, &got) == hipSuccess) a->text_cap = got;
            }
            return a;
        });
    }
    GI(hipMallocAsync((void **)&d_in, in_n + 8192, st));                 // the stage is filled 1 KB at a time, up to 2 KB ahead
    constexpr bool no_pread = false;
    bool uploaded = false;
    // (pinned buffers cost ~40 ms to make: a file of less than 256 MB takes that way only when a set is there already --
    //  ss_gz_warm_up, or an earlier call -- and then arrives in 8 ms instead of 12-30)
    const uint64_t pread_from = pin_waiting() ? 32ull << 20 : 256ull << 20;
    const std::vector<Bgzf> bgzf = bgzf_members(in, in_n);            // a bgzip file: its members ARE the chunks, no search
    uint64_t probe = 512;
    uint32_t c_searched = 0;                                   // search chunks [0, c_searched) have been launched
    auto search_to = [&](uint32_t c_hi) {
        if (c_hi > c_searched)
            hipLaunchKernelGGL(sync_kernel, dim3(c_hi - c_searched), dim3(64), 0, st, d_in, in_n - 8, data_off, chunk_bytes, n_chunks0, d_entry, probe,
                               trace ? 1 : 0, rr ? rr->slice_chunks : 0u, rr ? rr->rank : 0u, rr ? rr->world : 1u, LOOK, c_searched);
        c_searched = std::max(c_searched, c_hi);
    };
    if (fd >= 0 && in_n >= pread_from && !no_pread && !rr) {    // (`fd`: the same file; smaller ones are there before the buffers are)
        GI(hipMemsetAsync(d_in + in_n, 0, 8192, st));
        GI(hipMallocAsync((void **)&d_entry, (uint64_t)n_chunks0 * 8, st));
        GI(hipStreamSynchronize(st));                         // the allocations are stream-ordered
        lap("stage allocated");
        // the sync search runs on the prefix of the image that has arrived (a candidate's probe reads a few KB beyond its chunk)
        constexpr bool pipelined = true;
        const uint64_t margin = 64 << 10;
        uploaded = upload_file(fd, in_n, d_in, [&](uint64_t ready) {
            if (!pipelined || !bgzf.empty()) return;
            // (runs on an upload thread, which has set the device; calls are serialised by upload_file)
            const uint64_t usable = ready >= in_n ? in_n : (ready > margin + data_off ? ready - margin - data_off : 0);
            // (round 4, pieces > 1: the search starts on the prefix that has arrived, in that many pieces.  Measured (26 loads of each
            // setting interleaved in one process, a pair of 66 MB files): 1 piece -- upload, then search -- 26.0 ms, 2 pieces 26.3,
            // 4 pieces 27.0, 8 pieces 28.4: the search is bound by the chip's throughput (4.5 ms for the pair), a piece takes as
            // long as its slowest chunk, and the pieces of a stream run one after another.  So: off.
            constexpr uint32_t pieces = 1;
            const uint32_t c_hi = ready >= in_n ? n_chunks0 : (uint32_t)std::min<uint64_t>(n_chunks0, usable / chunk_bytes);
            if (c_hi == n_chunks0 || c_hi >= c_searched + (n_chunks0 + pieces - 1) / pieces) {
                if (trace) fprintf(stderr, "[ginflate] search to chunk %u of %u at %.4f s\n", c_hi, n_chunks0, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
                search_to(c_hi);
            }
        });
        if (!uploaded) c_searched = 0;                         // (the image is copied again below: search everything)
    }
    if (!uploaded && rr) {
        // range mode: only this rank's slices travel (+ the search chunks behind each in which its last chunk stops, + the
        // header and the trailer); the rest of the image is never read
        const uint64_t margin = 64 << 10;
        auto part = [&](uint64_t lo, uint64_t hi) { hi = std::min(hi, in_n); return lo >= hi || h2d(d_in + lo, in + lo, hi - lo); };
        GB(part(0, data_off + margin));
        GB(part(in_n > margin ? in_n - margin : 0, in_n));
        for (uint32_t sl : rr->mine) {
            const uint64_t lo = data_off + (uint64_t)sl * rr->slice_chunks * chunk_bytes, hi = lo + ((uint64_t)rr->slice_chunks + LOOK + 8) * chunk_bytes;
            GB(part(lo > margin ? lo - margin : 0, hi + margin));
        }
        uploaded = true;
    }
    if (!uploaded) GB(h2d(d_in, in, in_n));
    GI(hipMemsetAsync(d_in + in_n, 0, 8192, st));
    if (!d_entry) GI(hipMallocAsync((void **)&d_entry, (uint64_t)n_chunks0 * 8, st));
    lap("input on device");
    std::vector<uint64_t> entry(n_chunks0);
    if (bgzf.empty()) {
        search_to(n_chunks0);
        GB(d2h(entry.data(), d_entry, (uint64_t)n_chunks0 * 8));
    }
    if (trace && bgzf.empty()) {
        unsigned tries = 0;
        hipMemcpyFromSymbol(&tries, HIP_SYMBOL(g_sync_tries), 4);
        fprintf(stderr, "[ginflate] %u chunks, %u candidate blocks decoded\n", n_chunks0, tries);
        tries = 0;
        hipMemcpyToSymbol(HIP_SYMBOL(g_sync_tries), &tries, 4);
    }
    lap("sync");
    if (g_hook_entry.load() > 0) {                           // test hook (ss_test_hook): a wrong entry (a position inside a block) in chunk <n>
        const uint64_t c = (uint64_t)g_hook_entry.load();
        if (c > 0 && c < n_chunks0) entry[c] = (data_off + c * chunk_bytes) * 8 + 12345 % (chunk_bytes * 8);
    }
    // The file's chunks: one per entry (a chunk of the search without one belongs to its predecessor).  `fresh`: the first
    // chunk of a gzip member (nothing in front of it); `last`: it ends with the member's final block, the trailer follows
    // at `trailer`.
    // `hdr`: ~0, or -- an entry INSIDE a block (subsync_kernel) -- where that block's header is.
    struct Chunk { uint64_t start; bool fresh, last; uint64_t trailer; uint64_t hdr = ~0ull; };
    std::vector<Chunk> G;
    struct Seg { size_t gi, gj, n_ph; uint32_t slice; uint64_t hdr_bit; };
    std::vector<Seg> segs;                                   // range mode: one segment per slice of this rank, its look-ahead entries behind it
    const bool is_bgzf = !bgzf.empty();
    if (rr && is_bgzf) {
        // bgzip in range mode: a slice's chunks are the members whose deflate data begins in its byte range -- complete in
        // themselves (no search, no look-ahead, nothing unknown in front of any); the chain still carries the newline count,
        // the record that straddles the cut and the position the slice must begin at (hdr_bit: where its first member's header starts)
        size_t mi = 0;
        for (uint32_t sl : rr->mine) {
            const uint64_t lo = sl == 0 ? 0 : data_off + (uint64_t)sl * rr->slice_chunks * chunk_bytes;
            const uint64_t hi = sl + 1 == rr->n_slices ? ~0ull : data_off + (uint64_t)(sl + 1) * rr->slice_chunks * chunk_bytes;
            while (mi < bgzf.size() && bgzf[mi].data < lo) mi++;
            Seg sg{G.size(), 0, 0, sl, mi ? (bgzf[mi - 1].trailer + 8) * 8 : 0};
            for (; mi < bgzf.size() && bgzf[mi].data < hi; mi++) G.push_back(Chunk{bgzf[mi].data * 8, true, true, bgzf[mi].trailer});
            sg.gj = G.size();
            if (sg.gj == sg.gi) return no("slice without a member", sl);
            segs.push_back(sg);
        }
        if (trace) fprintf(stderr, "[ginflate] bgzip: %zu members, %zu in this rank's slices\n", bgzf.size(), G.size());
    } else if (rr) {
        for (uint32_t sl : rr->mine) {
            const uint32_t c_lo = sl * rr->slice_chunks, c_hi = std::min<uint32_t>(n_chunks0, c_lo + rr->slice_chunks);
            Seg sg{G.size(), 0, 0, sl, 0};
            for (uint32_t c = c_lo; c < c_hi; c++)
                if (entry[c] != ~0ull) G.push_back(Chunk{entry[c], c == 0, false, 0});
            sg.gj = G.size();
            if (sg.gj == sg.gi) return no("slice without a block start", sl);
            if (sl + 1 < rr->n_slices) {
                for (uint32_t c = c_hi; c < std::min<uint32_t>(n_chunks0, c_hi + LOOK) && sg.n_ph < 2; c++)
                    if (entry[c] != ~0ull) { G.push_back(Chunk{entry[c], false, false, 0}); sg.n_ph++; }
                if (!sg.n_ph) return no("no block start behind the slice", sl);
            } else {
                G[sg.gj - 1].last = true;
                G[sg.gj - 1].trailer = in_n - 8;
            }
            segs.push_back(sg);
        }
    } else if (bgzf.empty()) {
        for (uint32_t c = 0; c < n_chunks0; c++)
            if (entry[c] != ~0ull) G.push_back(Chunk{entry[c], c == 0, false, 0});
        G.back().last = true;
        G.back().trailer = in_n - 8;
    } else {
        for (const Bgzf &m : bgzf) G.push_back(Chunk{m.data * 8, true, true, m.trailer});
        if (trace) fprintf(stderr, "[ginflate] bgzip: %zu members\n", bgzf.size());
    }
    entry.clear();
    entry.shrink_to_fit();
    uint32_t n_sub = 0;
    if (split_bytes && !is_bgzf && !G.empty()) {
        // ---- entries inside the blocks: a block [its start, the next entry) of more than 2 x split_bytes is entered at
        //      equidistant places as well (at most eight pieces; range mode: the look-ahead entries stay whole)
        std::vector<uint64_t> b_hdr, s_from, s_lim;
        std::vector<uint32_t> b_first, s_of;                   // positions of a block: [b_first[b], b_first[b + 1]); s_of: which chunk of G
        auto want = [&](size_t i, uint64_t endb) {
            const uint64_t len = endb - G[i].start, pieces = std::min<uint64_t>(8, len / (split_bytes * 8));
            if (pieces < 2) return;
            b_hdr.push_back(G[i].start);
            b_first.push_back((uint32_t)s_of.size());
            for (uint64_t j = 1; j < pieces; j++) { s_from.push_back(G[i].start + len * j / pieces); s_lim.push_back(endb); s_of.push_back((uint32_t)i); }
        };
        if (rr) {
            for (const Seg &sg : segs)
                for (size_t i = sg.gi; i < sg.gj; i++) {
                    if (i + 1 < sg.gj + sg.n_ph) want(i, G[i + 1].start);
                    else if (G[i].last) want(i, G[i].trailer * 8);
                }
        } else {
            for (size_t i = 0; i < G.size(); i++) want(i, i + 1 < G.size() ? G[i + 1].start : (in_n - 8) * 8);
        }
        const size_t ns = s_of.size(), nb = b_hdr.size();
        b_first.push_back((uint32_t)ns);
        if (ns) {
            uint64_t *d_sub = nullptr;                         // block headers | from | limit | entries | first (u32)
            GI(hipMallocAsync((void **)&d_sub, nb * 8 + ns * 8 * 3 + (nb + 1) * 4, st));
            uint64_t *d_bh = d_sub, *d_from = d_sub + nb, *d_lim = d_from + ns, *d_ent = d_lim + ns;
            uint32_t *d_first = reinterpret_cast<uint32_t *>(d_ent + ns);
            std::vector<uint64_t> got(ns);
            const bool ok = h2d(d_bh, b_hdr.data(), nb * 8) && h2d(d_from, s_from.data(), ns * 8) && h2d(d_lim, s_lim.data(), ns * 8) &&
                            h2d(d_first, b_first.data(), (nb + 1) * 4);
            if (ok) hipLaunchKernelGGL(subsync_kernel, dim3((unsigned)nb), dim3(64), 0, st, d_in, in_n - 8, d_bh, d_first, d_from, d_lim, (uint32_t)nb, d_ent);
            const bool ok2 = ok && d2h(got.data(), d_ent, ns * 8);
            hipFreeAsync(d_sub, st);
            if (!ok2) return no("sub-entries");
            std::vector<Chunk> G2;
            std::vector<size_t> at(G.size() + 1);
            size_t k = 0;
            for (size_t i = 0; i < G.size(); i++) {
                at[i] = G2.size();
                G2.push_back(G[i]);
                uint64_t prev = G[i].start;
                for (; k < ns && s_of[k] == i; k++) {
                    if (got[k] == ~0ull || got[k] <= prev || got[k] + 256 >= s_lim[k]) continue;      // (none found, or two searches met in one place)
                    Chunk sub{got[k], false, G[i].last, G[i].trailer, G[i].start};
                    G2.back().last = false;                    // the member's final block ends the LAST piece
                    G2.back().trailer = 0;
                    G2.push_back(sub);
                    prev = got[k];
                    n_sub++;
                }
            }
            at[G.size()] = G2.size();
            for (Seg &sg : segs) { const size_t ph = sg.n_ph; sg.gi = at[sg.gi]; sg.gj = at[sg.gj]; sg.n_ph = ph; }
            G.swap(G2);
        }
        if (trace) fprintf(stderr, "[ginflate] %zu entries inside blocks wanted, %u found: %zu chunks\n", ns, n_sub, G.size());
        lap("sub-entries");
    }

    // What a wave decodes alone is the stretch from its entry to the next.  Entries are dynamic blocks' starts (and places inside
    // them): zlib, pigz and libdeflate begin one every 16-300 KB, but a stream of stored or fixed-Huffman blocks only (level 0,
    // Z_FIXED, some hardware compressors) has none, and ONE wave would then decode the whole file -- minutes of a kernel that looks
    // like a hang.  Such a file goes to the host inflaters, which read it at memory speed.
    if (!rr && !is_bgzf) {
        constexpr uint64_t LONELY_BYTES = 8ull << 20;
        uint64_t longest = 0;
        for (size_t i = 0; i < G.size(); i++)
            longest = std::max(longest, ((i + 1 < G.size() ? G[i + 1].start : (in_n - 8) * 8) - G[i].start) / 8);
        if (longest > LONELY_BYTES) return no("a stretch of deflate data without a dynamic block's start (MB)", (long long)(longest >> 20));
    }
    // The chunks are inflated SEGMENT by segment (seg_bytes of deflate data, 128 MB): the scratch stays a few GB whatever
    // the file's size, and the text of one segment is complete -- bytes -- before the next one starts, so the 32 KB in
    // front of a segment's first chunk are simply the end of the text so far.
    auto segment_end = [&](size_t gi) {
        size_t gj = gi + 1;
        while (gj < G.size() && (G[gj].start - G[gi].start) / 8 < seg_bytes && gj - gi < SEG_CHUNKS) gj++;
        return gj;
    };
    uint64_t need_sym = 0, need_chunks = 0;                   // the largest segment decides the scratch
    for (size_t gi = 0, si = 0; gi < G.size(); si++) {
        const size_t gj = rr ? segs[si].gj : segment_end(gi);
        const bool more = rr ? segs[si].n_ph > 0 : gj < G.size();
        const uint64_t bytes = ((more ? G[gj].start : G[gj - 1].last ? G[gj - 1].trailer * 8 : (in_n - 8) * 8) - G[gi].start) / 8;
        need_sym = std::max<uint64_t>(need_sym, (bytes + (gj - gi)) * ratio + (gj - gi) * 4096);
        need_chunks = std::max<uint64_t>(need_chunks, gj - gi);
        gi = rr ? (si + 1 < segs.size() ? segs[si + 1].gi : G.size()) : gj;
    }
    const uint64_t cap_chunks = need_chunks + 80;
    const uint64_t sym_elems = need_sym + need_sym / 4 + 64 * (4096 + 64 * ratio);      // (+ what run-over and further members add)
    if (sym_elems * 2 > (24ull << 30)) return no("segment", (long long)(sym_elems >> 20));      // (GBs without a single block start)
    uint64_t text_cap;
    if (ahead.valid()) {
        A = ahead.get();
        if (A && (A->cap_chunks < cap_chunks || A->sym_elems < sym_elems)) { arena_put(A); A = nullptr; }      // (cannot happen: the bounds are bounds)
    }
    {
        text_cap = text_guess();
        if (rr) text_cap = text_cap / rr->n_slices * rr->mine.size() * 13 / 10 + (1ull << 20);      // this rank's share (it grows when short)
        if (!A) A = arena_take_if_fits(cap_chunks, sym_elems, text_cap);
        if (!A) {
            size_t mem_free = 0, mem_total = 0;
            GI(hipMemGetInfo(&mem_free, &mem_total));
            const uint64_t need = 2 * text_cap + sym_elems * 2 + cap_chunks * WSIZE * 5 + (256ull << 20);
            if (need > mem_free / 2) return no("device memory", (long long)(need >> 20));
        }
    }
    if (!A) A = arena_get(cap_chunks, sym_elems);
    if (!A) return no("scratch");
    if (A->text_cap < text_cap) {
        if (A->text) hipFree(A->text);
        A->text = nullptr;
        A->text_cap = 0;
        GI(ss::big_malloc((void **)&A->text, text_cap, &text_cap));
        A->text_cap = text_cap;
    } else {
        text_cap = A->text_cap;
    }
    d_text = A->text;
    lap("buffers");
    uint64_t *d_start = A->meta, *d_stop = A->meta + cap_chunks, *d_off = A->meta + 2ull * cap_chunks, *d_cap = A->meta + 3ull * cap_chunks,
             *d_len = A->meta + 4ull * cap_chunks, *d_end = A->meta + 5ull * cap_chunks, *d_toff = A->meta + 6ull * cap_chunks, *d_hdr = A->meta + 7ull * cap_chunks;
    const uint32_t max_over = getenv("SS_GZ_NO_RUNOVER") ? 0u : n_sub ? 4u : 2u;      // (test hook: wrong entries are then handled by the host only)
    struct Member { uint64_t at, len; uint32_t crc, isize; bool open; uint32_t crc0; uint64_t len0; };      // crc0, len0: range mode -- the member's part in the slices before
    std::vector<Member> members;
    uint64_t total = 0, last_end_bit = 0;
    bool have_prev = false, ended = false;
    uint32_t n_segments = 0;

    // CRC-32 of text[at, at + n) for a list of (at, n) (range mode: every member's part in one slice, ONE launch for all of them --
    // a bgzip slice holds thousands of members; the members of the whole-file path are done together below)
    auto pieces_crc = [&](const std::vector<std::pair<uint64_t, uint64_t>> &items, std::vector<uint32_t> &out) -> bool {
        constexpr int LG = 12;
        const uint64_t sg = 1ull << LG;
        std::vector<uint64_t> at_v;
        std::vector<uint32_t> ln_v, tabv(256);
        for (const auto &it : items)
            for (uint64_t a0 = 0; a0 < it.second; a0 += sg) { at_v.push_back(it.first + a0); ln_v.push_back((uint32_t)std::min<uint64_t>(sg, it.second - a0)); }
        const uint64_t ns = at_v.size();
        out.assign(items.size(), (uint32_t)crc32(0L, Z_NULL, 0));
        if (!ns) return true;
        std::vector<uint32_t> got(ns);
        for (uint32_t i = 0; i < 256; i++) { uint32_t kx = i; for (int j = 0; j < 8; j++) kx = (kx & 1u) ? 0xEDB88320u ^ (kx >> 1) : kx >> 1; tabv[i] = kx; }
        uint32_t *pt = nullptr, *pc = nullptr;
        if (hipMallocAsync((void **)&pt, 1024 + ns * 12, st) != hipSuccess) return false;
        if (hipMallocAsync((void **)&pc, ns * 4, st) != hipSuccess) { hipFreeAsync(pt, st); return false; }
        uint64_t *p_at = reinterpret_cast<uint64_t *>(pt + 256);
        uint32_t *p_ln = reinterpret_cast<uint32_t *>(p_at + ns);
        bool ok = h2d(pt, tabv.data(), 1024) && h2d(p_at, at_v.data(), ns * 8) && h2d(p_ln, ln_v.data(), ns * 4);
        if (ok) {
            hipLaunchKernelGGL(crc_kernel, dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, st, d_text, p_at, p_ln, ns, pt, pc);
            ok = d2h(got.data(), pc, ns * 4);
        }
        hipFreeAsync(pt, st);
        hipFreeAsync(pc, st);
        if (!ok) return false;
        uint32_t op[32];
        crc_zero_operator(op, LG);
        uint64_t si = 0;
        for (size_t k = 0; k < items.size(); k++) {
            uint32_t crc = out[k];
            for (uint64_t a0 = 0; a0 < items[k].second; a0 += sg, si++)
                crc = ln_v[si] == sg ? gf2_times(op, crc) ^ got[si] : (uint32_t)crc32_combine(crc, got[si], (z_off_t)ln_v[si]);
            out[k] = crc;
        }
        return true;
    };
    size_t seg_i = 0;
    for (size_t gi = 0; gi < G.size();) {
        // ---- the segment's chunks [gi, gj) and up to two look-ahead entries behind them (what a chunk may run over)
        const size_t gj = rr ? segs[seg_i].gj : segment_end(gi);
        const size_t ph_end = rr ? gj + segs[seg_i].n_ph : std::min(G.size(), gj + 2);
        std::vector<Chunk> ch(G.begin() + (long)gi, G.begin() + (long)gj), ph(G.begin() + (long)gj, G.begin() + (long)ph_end);
        uint32_t nc = (uint32_t)ch.size();
        std::vector<uint64_t> start, stop, off, cap, hdrs;
        uint64_t sym_total = 0;
        auto bit_behind = [&](uint32_t c) -> uint64_t {      // where chunk c's input ends at the latest
            if (ch[c].last) return ch[c].trailer * 8;
            if (c + 1 < nc) return ch[c + 1].start;
            if (!ph.empty()) return ph[0].start;
            return (in_n - 8) * 8;
        };
        auto lay_out = [&] {
            start.clear(); stop.clear(); off.clear(); cap.clear(); hdrs.clear();
            sym_total = 0;
            for (uint32_t c = 0; c < nc; c++) {
                hdrs.push_back(ch[c].hdr);
                start.push_back(ch[c].start | (ch[c].fresh ? 1ull << 63 : 0ull));
                stop.push_back(ch[c].last ? ~0ull : bit_behind(c));
                const uint64_t cbits = bit_behind(c) - ch[c].start;
                const uint64_t cp = (cbits / 8 + 1) * ratio + 4096;
                off.push_back(sym_total);
                cap.push_back(cp);
                sym_total += cp;
            }
            for (size_t k = 0; k < ph.size(); k++) {              // look-ahead: only their stops are read
                start.push_back(ph[k].start);
                hdrs.push_back(ph[k].hdr);
                // (range mode: what follows the look-ahead entries in G is another slice: a chunk that runs over both of them is
                //  stopped a few search chunks further on and declined)
                stop.push_back(ph[k].last ? ~0ull : (k + 1 < ph.size() ? ph[k + 1].start
                                                     : rr ? ph[k].start + 4 * chunk_bytes * 8 : (gj + k + 1 < G.size() ? G[gj + k + 1].start : ~0ull)));
                off.push_back(0);
                cap.push_back(0);
            }
        };
        lay_out();
        if (sym_total > A->sym_elems) return no("segment", (long long)gi);
        std::vector<int> status(nc, 0);
        std::vector<uint64_t> out_len(nc, 0), end_bit(nc, 0);
        std::vector<uint32_t> todo;                                // empty = all chunks
        uint32_t consumed_ph = 0;
        // Two things show only when the chunks have been inflated:
        //  * An entry is a position where a valid dynamic header parses and 512 symbols decode -- a position INSIDE a
        //    block passes that about once in a million candidates (every bit string decodes under a complete code).  The
        //    chunk in front of it ends a block BEHIND it and goes on to the entry after it (inflate_kernel); the wrong
        //    entry's chunk is dropped.  (No room left in its symbol region, or more than two in a row: -21, the two
        //    chunks are merged here and inflated again.)
        //  * A file of several members (lanes joined with `cat a.gz b.gz`): the chunk that meets a final block before its
        //    stop (-20) ends a member if a trailer and a gzip header follow; the next member's first block becomes a chunk
        //    and the segment is inflated again.
        for (int attempt = 0;; attempt++) {
            const uint32_t n_all = nc + (uint32_t)ph.size();
            GB(h2d(d_start, start.data(), (uint64_t)n_all * 8));
            GB(h2d(d_stop, stop.data(), (uint64_t)n_all * 8));
            GB(h2d(d_off, off.data(), (uint64_t)n_all * 8));
            GB(h2d(d_cap, cap.data(), (uint64_t)n_all * 8));
            GB(h2d(d_hdr, hdrs.data(), (uint64_t)n_all * 8));
            if (!todo.empty()) {                                   // what the other chunks produced stays as it is
                GB(h2d(A->status, status.data(), (uint64_t)nc * 4));
                GB(h2d(d_len, out_len.data(), (uint64_t)nc * 8));
                GB(h2d(d_end, end_bit.data(), (uint64_t)nc * 8));
                GB(h2d(A->todo, todo.data(), todo.size() * 4));
            }
            const uint32_t n_run = todo.empty() ? nc : (uint32_t)todo.size();
            hipLaunchKernelGGL(inflate_kernel, dim3(n_run), dim3(64), 0, st, d_in, in_n - 8, d_start, d_stop, d_hdr, n_all, A->sym, d_off, d_cap, d_len, d_end, A->status,
                               todo.empty() ? (const uint32_t *)nullptr : A->todo, max_over);
            GB(d2h(status.data(), A->status, (uint64_t)nc * 4));
            GB(d2h(out_len.data(), d_len, (uint64_t)nc * 8));
            GB(d2h(end_bit.data(), d_end, (uint64_t)nc * 8));
#ifdef SS_GZ_TIMING
            {
                unsigned long long t[12], z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                hipMemcpyFromSymbol(t, HIP_SYMBOL(g_gz_t), sizeof t);
                hipMemcpyToSymbol(HIP_SYMBOL(g_gz_t), z, sizeof z);
                fprintf(stderr, "[ginflate] per wave (cycles): total %.0f = decode %.0f + chain %.0f + deliver %.0f (flush %.0f inside) + header %.0f; windows %.0f; waves %llu\n",
                        (double)t[10] / t[11], (double)t[0] / t[11], (double)t[1] / t[11], (double)t[2] / t[11], (double)t[3] / t[11], (double)t[4] / t[11], (double)t[5] / t[11], t[11]);
                fprintf(stderr, "[ginflate] matches %llu (symbols %llu), beyond the ring %llu, into the unknown window %llu; windows %llu\n", t[6], t[7], t[8], t[9], t[5]);
            }
#endif
            // the new chunk list, with what the unchanged chunks produced; `again`: positions in it that must be inflated (again)
            std::vector<Chunk> nxt;
            std::vector<uint64_t> n_off, n_cap, n_len, n_end;
            std::vector<int> n_status;
            std::vector<uint32_t> again;
            std::vector<char> drop(nc, 0);
            uint32_t n_drop = 0, n_members = 0, n_over = 0, over_ph = 0;
            bool relayout = false;                                 // new chunks need room of their own: lay everything out anew
            auto keep = [&](uint32_t c, const Chunk &a) {
                nxt.push_back(a); n_off.push_back(off[c]); n_cap.push_back(cap[c]); n_len.push_back(out_len[c]); n_end.push_back(end_bit[c]);
                n_status.push_back(status[c]);
            };
            for (uint32_t c = 0; c < nc; c++) {
                if (drop[c]) continue;                                     // its own outcome means nothing
                const uint64_t e = (end_bit[c] + 7) / 8;
                if (status[c] >= 0) {
                    // done; `over` entries behind it were positions inside its blocks: their chunks go, nothing is inflated again
                    const uint32_t over = (uint32_t)status[c] >> 4;
                    const int stc = status[c] & 15;
                    if (c + over < n_all) {
                        const Chunk &eff = c + over < nc ? ch[c + over] : ph[c + over - nc];
                        if (stc == (eff.last ? 1 : 0) && (!eff.last || e == eff.trailer)) {
                            Chunk a = ch[c];
                            a.last = eff.last;
                            a.trailer = eff.trailer;
                            keep(c, a);
                            n_status.back() = stc;
                            for (uint32_t k = 1; k <= over; k++) {
                                if (c + k < nc) drop[c + k] = 1;
                                else over_ph = std::max(over_ph, c + k - nc + 1);
                            }
                            n_over += over;
                            continue;
                        }
                    }
                    status[c] = stc == 1 ? 1 : -21;                        // (falls through: a member's end, or not explainable)
                }
                if (status[c] == -21 && c + 1 < nc) {                      // ran past the next entry: the two chunks become one
                    // (... and with them the entries inside blocks that follow: found with the tables of a block they are not in --
                    //  a chunk of the search that held two block starts --, what they decoded means nothing)
                    uint32_t k = c + 1;
                    while (k + 1 < nc && ch[k + 1].hdr != ~0ull) k++;
                    Chunk a = ch[c];
                    a.last = ch[k].last;
                    a.trailer = ch[k].trailer;
                    keep(c, a);
                    for (uint32_t d = c + 1; d <= k; d++) {
                        drop[d] = 1;
                        n_drop++;
                        if (off[d] == off[c] + n_cap.back()) n_cap.back() += cap[d];      // their symbol regions are neighbours
                        else relayout = true;
                    }
                    again.push_back((uint32_t)nxt.size() - 1);
                    continue;
                }
                if (status[c] == -20 || status[c] == 1) {      // a final block before the next entry / before the file's end (its own, or run over to it)
                    const uint64_t hdr = e + 8 + 18 <= in_n ? gzip_header_len(in + e + 8, in_n - (e + 8)) : 0;
                    if (!hdr && ch[c].hdr != ~0ull && !nxt.empty()) {         // (an entry inside a block that decoded garbage: see below)
                        Chunk &pv = nxt.back();
                        pv.last = ch[c].last;
                        pv.trailer = ch[c].trailer;
                        if (n_off.size() == nxt.size() && off[c] == n_off.back() + n_cap.back()) n_cap.back() += cap[c];
                        else relayout = true;
                        if (again.empty() || again.back() != (uint32_t)nxt.size() - 1) again.push_back((uint32_t)nxt.size() - 1);
                        n_drop++;
                        continue;
                    }
                    if (!hdr) return no("chunk status", status[c] * 1000000ll + c);
                    Chunk a = ch[c];
                    a.last = true;
                    a.trailer = e;
                    keep(c, a);
                    const uint64_t d = (e + 8 + hdr) * 8;                  // the next member's first block
                    bool was_last = ch[c].last;                            // (the new chunk ends the file if what it replaces did)
                    uint64_t was_trailer = ch[c].trailer;
                    // "entries" within the block, trailer and header -- and entries that claim to lie inside a block of the member that ends here
                    for (uint32_t k = c + 1; k < nc && (ch[k].start < d || (ch[k].hdr != ~0ull && ch[k].hdr < d)); k++) {
                        drop[k] = 1;
                        n_drop++;
                        if (ch[k].last) { was_last = true; was_trailer = ch[k].trailer; }
                    }
                    if (c + 1 >= nc || drop[nc - 1])                       // (the look-ahead entries too)
                        for (size_t k = 0; k < ph.size() && ph[k].start < d; k++) over_ph = std::max<uint32_t>(over_ph, (uint32_t)k + 1);
                    nxt.push_back(Chunk{d, true, was_last, was_trailer});
                    n_members++;
                    relayout = true;
                    continue;
                }
                if (ch[c].hdr != ~0ull && !nxt.empty()) {
                    // an entry inside a block whose chunk cannot be explained (the chunk in front did not arrive there: what it decoded
                    // means nothing): no entry -- the chunk in front takes its bytes and is inflated again
                    Chunk &pv = nxt.back();
                    pv.last = ch[c].last;
                    pv.trailer = ch[c].trailer;
                    if (n_off.size() == nxt.size() && off[c] == n_off.back() + n_cap.back()) n_cap.back() += cap[c];
                    else relayout = true;
                    if (again.empty() || again.back() != (uint32_t)nxt.size() - 1) again.push_back((uint32_t)nxt.size() - 1);
                    n_drop++;
                    continue;
                }
                return no("chunk status", status[c] * 1000000ll + c);
            }
            if (over_ph) {                                             // look-ahead entries that were run over are no chunks any more
                over_ph = (uint32_t)std::min<size_t>(over_ph, ph.size());
                consumed_ph += over_ph;
                ph.erase(ph.begin(), ph.begin() + over_ph);
                // (the arrays' look-ahead part is only read by chunks that run over again: rebuilt below when something is inflated again)
            }
            if (trace && n_over) fprintf(stderr, "[ginflate] %u entries were inside a block: run over\n", n_over);
            if (!n_drop && !n_members) {
                if (n_over) {                                          // the shorter chunk list, everything else as it is
                    ch.swap(nxt);
                    nc = (uint32_t)ch.size();
                    off.swap(n_off); cap.swap(n_cap); out_len.swap(n_len); end_bit.swap(n_end); status.swap(n_status);
                }
                break;
            }
            if (trace) fprintf(stderr, "[ginflate] %u entries were inside a block, %u further members found: %s inflated again\n", n_drop, n_members,
                               relayout ? "the segment" : "their chunks");
            if (attempt >= 6 || nxt.size() + ph.size() > cap_chunks - 4) return no("chunk list", (long long)nxt.size());
            ch.swap(nxt);
            nc = (uint32_t)ch.size();
            if (relayout) {
                lay_out();
                if (sym_total > A->sym_elems) return no("symbol budget");
                todo.clear();
                status.assign(nc, 0);
                out_len.assign(nc, 0); end_bit.assign(nc, 0);
            } else {
                off.swap(n_off); cap.swap(n_cap); out_len.swap(n_len); end_bit.swap(n_end); status.swap(n_status);
                const std::vector<uint64_t> o2 = off, c2 = cap;
                lay_out();                                             // (starts, stops, look-ahead) ...
                off = o2; cap = c2;                                    // ... the symbol regions stay where they are
                off.resize(nc + ph.size(), 0); cap.resize(nc + ph.size(), 0);
                todo.swap(again);
            }
        }
        // ---- range mode: what lies in front of this slice comes down the chain now (the symbols are ready: the ranks did that
        //      part at the same time); the slice must begin exactly where the one before ended
        const uint32_t my_slice = rr ? segs[seg_i].slice : 0;
        uint64_t nl_before = 0, len_before = 0;
        uint32_t crc_before = (uint32_t)crc32(0L, Z_NULL, 0);
        std::vector<uint8_t> carry_in;
        if (rr) {
            if (consumed_ph || ch.size() == 0) return no("range: entries dropped at the slice edge", my_slice);
            if (my_slice > 0) {
                if (!rr->recv_for(my_slice)) return no("chain receive", my_slice);
                if (rr->msg.status < 0) return no("chain: a rank before this one declined", my_slice);
                if (rr->msg.end_bit != (is_bgzf ? segs[seg_i].hdr_bit : ch[0].start) || rr->msg.carry_len > CARRY_MAX)
                    return no("range: the slice before ends elsewhere", my_slice);
                GB(h2d(A->prev, rr->msg.window, WSIZE));
                have_prev = true;
                nl_before = rr->msg.nl; len_before = rr->msg.len; crc_before = rr->msg.crc;
                carry_in.assign(rr->msg.carry, rr->msg.carry + rr->msg.carry_len);
            } else {
                have_prev = false;
            }
            // the member that is open at the cut goes on in this slice (several members -- lanes joined with cat -- are followed
            // as in the whole-file path: a member that ends inside the slice is checked against its trailer here, the next one
            // starts with nothing in front of it; CRC-32 and length of the open member travel down the chain)
            if (!ch[0].fresh) members.push_back(Member{total, 0, 0, 0, true, crc_before, len_before});
        }
        const size_t m_first = members.empty() ? 0 : members.size() - ((rr && !ch[0].fresh) ? 1 : 0);      // members of this segment: [m_first, ...)
        // ---- the segment's text
        std::vector<uint64_t> text_off(nc, 0);
        for (uint32_t c = 0; c < nc; c++) {
            if (ch[c].fresh) members.push_back(Member{total, 0, 0, 0, true, (uint32_t)crc32(0L, Z_NULL, 0), 0});
            if (members.empty() || !members.back().open) return no("member start");
            text_off[c] = total;
            total += out_len[c];
            members.back().len += out_len[c];
            if (ch[c].last) {
                const uint8_t *t8 = in + ch[c].trailer;
                Member &m = members.back();
                m.crc = (uint32_t)t8[0] | (uint32_t)t8[1] << 8 | (uint32_t)t8[2] << 16 | (uint32_t)t8[3] << 24;
                m.isize = (uint32_t)t8[4] | (uint32_t)t8[5] << 8 | (uint32_t)t8[6] << 16 | (uint32_t)t8[7] << 24;
                m.open = false;
                if ((uint32_t)(m.len + m.len0) != m.isize) return no("isize", (long long)members.size());      // (range mode: + its part in the slices before)
                ended = ch[c].trailer == in_n - 8;
            }
        }
        last_end_bit = end_bit[nc - 1];
        if (total + 64 > text_cap) {                                   // (several members: the last ISIZE said little)
            const uint64_t ncap = std::max(total + 64, text_cap + text_cap / 2);
            uint8_t *nt = nullptr;
            GI(hipMalloc((void **)&nt, ncap));
            const uint64_t have = total - (total - text_off[0]);
            const bool ok = hipMemcpyAsync(nt, d_text, have, hipMemcpyDeviceToDevice, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            hipFree(d_text);
            d_text = nt;
            text_cap = ncap;
            A->text = nt;
            A->text_cap = ncap;
            if (!ok) return no("text copy");
        }
        GB(h2d(d_off, off.data(), (uint64_t)nc * 8));
        GB(h2d(d_len, out_len.data(), (uint64_t)nc * 8));
        GB(h2d(d_toff, text_off.data(), (uint64_t)nc * 8));
        {
            uint32_t group = 1;
            while ((uint64_t)group * group < nc) group++;               // ~sqrt: as many groups as chunks in a group
            group = std::max<uint32_t>(group, 8);
            const uint32_t n_groups = (nc + group - 1) / group;
            hipLaunchKernelGGL(tails_kernel, dim3(16, nc), dim3(256), 0, st, A->sym, d_off, d_len, nc, A->map[0]);
            hipLaunchKernelGGL(group_maps_kernel, dim3(n_groups), dim3(1024), 0, st, A->map[0], nc, group, A->map[1]);
            hipLaunchKernelGGL(group_windows_kernel, dim3(1), dim3(1024), 0, st, A->map[1], nc, group, n_groups,
                               have_prev ? (const uint8_t *)A->prev : (const uint8_t *)nullptr, A->gwin);
            hipLaunchKernelGGL(windows_kernel, dim3(16, nc), dim3(256), 0, st, A->map[1], nc, group, (const uint8_t *)A->gwin, A->win);
            hipLaunchKernelGGL(bytes_kernel, dim3(64, nc), dim3(256), 0, st, A->sym, d_off, d_len, d_toff, A->win, d_text);
            hipLaunchKernelGGL(lastwin_kernel, dim3(WSIZE / 256), dim3(256), 0, st, d_text, total, A->prev);
            GI(hipGetLastError());
            GI(hipStreamSynchronize(st));                          // (the host arrays of this segment go out of scope)
        }
        have_prev = true;
        n_segments++;
        if (rr) {
            // ---- this slice's piece: CRC, newlines, the bytes behind its last complete record; then the chain goes on
            const uint64_t p_at = text_off[0], p_len = total - text_off[0];
            if (p_len < WSIZE && my_slice + 1 < rr->n_slices) return no("range: a slice with less text than a window", my_slice);      // (the window it hands on would reach into another piece)
            // CRC-32 of every member's part in this slice: a member that ended here against its trailer, the open one goes on
            uint32_t crc_now = (uint32_t)crc32(0L, Z_NULL, 0);
            uint64_t len_now = 0;
            std::vector<std::pair<uint64_t, uint64_t>> m_items;
            for (size_t mi = m_first; mi < members.size(); mi++) m_items.emplace_back(members[mi].at, members[mi].len);
            std::vector<uint32_t> m_crc;
            if (!pieces_crc(m_items, m_crc)) return no("piece crc", my_slice);
            for (size_t mi = m_first; mi < members.size(); mi++) {
                const Member &m = members[mi];
                const uint32_t c = m.len ? (uint32_t)crc32_combine(m.crc0, m_crc[mi - m_first], (z_off_t)m.len) : m.crc0;
                if (!m.open) { if (c != m.crc) return no("crc (range mode)", (long long)mi); }
                else if (mi + 1 == members.size()) { crc_now = c; len_now = m.len0 + m.len; }
                else return no("member left open", (long long)mi);
            }
            unsigned long long p_nl = 0;
            GI(hipMemsetAsync(d_entry, 0, 8, st));                   // (the entries are on the host by now: a free device word)
            hipLaunchKernelGGL(count_nl_kernel, dim3(1024), dim3(256), 0, st, d_text + p_at, p_len, reinterpret_cast<unsigned long long *>(d_entry));
            GB(d2h(&p_nl, d_entry, 8));
            const bool final_slice = my_slice + 1 == rr->n_slices;
            uint64_t keep = p_len;                                   // bytes of the piece up to the end of its last complete record
            if (!final_slice) {
                // records end at newlines whose number in the file is a multiple of four: the last such newline of this piece is
                // among its last four; the tail of the piece is looked at on the host
                const uint64_t drop = (nl_before + p_nl) % 4;        // newlines behind the last complete record
                if (p_nl < drop + 1) return no("range: a slice without a complete record", my_slice);
                const uint64_t tail = std::min<uint64_t>(p_len, (uint64_t)CARRY_MAX + 1);
                std::vector<uint8_t> tb(tail);
                GB(d2h(tb.data(), d_text + p_at + (p_len - tail), tail));
                uint64_t seen = 0, q = tail;
                while (q > 0) {                                      // q - 1: the newline that ends the last complete record
                    if (tb[q - 1] == '\n') { if (seen == drop) break; seen++; }
                    q--;
                }
                if (q == 0) return no("range: a record longer than the chain carries", my_slice);
                keep = p_len - tail + q;
                rr->msg.status = 0;
                rr->msg.crc = crc_now;
                rr->msg.len = len_now;
                rr->msg.nl = nl_before + p_nl;
                rr->msg.end_bit = is_bgzf ? (ch[nc - 1].trailer + 8) * 8 : end_bit[nc - 1];      // (bgzip: the next member's header)
                rr->msg.carry_len = (uint32_t)(p_len - keep);
                memcpy(rr->msg.carry, tb.data() + q, p_len - keep);
                GB(d2h(rr->msg.window, A->prev, WSIZE));
                if (!rr->send_from(my_slice)) return no("chain send", my_slice);
            } else if (members.empty() || members.back().open) {
                return no("the last member is not closed (range mode)");     // (every closed member was checked against its trailer above)
            }
            rr->pieces->push_back(GzPiece{p_at, p_len, keep, carry_in});
            rr->duty++;
            rr->received = false;
            seg_i++;
            gi = seg_i < segs.size() ? segs[seg_i].gi : G.size();
        } else {
            gi = gj + consumed_ph;
        }
    }
    lap("inflate + windows + bytes");
    if (trace) fprintf(stderr, "[ginflate] %u segments, %zu members\n", n_segments, members.size());
    if (rr) {
        // (every piece was checked as it was made; the owner of the last slice has compared CRC-32 and length with the trailer
        //  and seen the stream end where the trailer begins)
        if (!rr->mine.empty() && rr->mine.back() + 1 == rr->n_slices && (!ended || (last_end_bit + 7) / 8 != in_n - 8))
            return no("stream end (range mode)", (long long)((last_end_bit + 7) / 8));
        cleanup(true);
        g_handled++;
        g_range_files++;
        g_range_pieces += rr->pieces->size();
        *text_dev = (char *)d_text;
        *len = total;
        *lease = A;
        return true;
    }
    // the stream must end where the last trailer begins (after padding to a byte)
    if (!ended || members.empty() || members.back().open || (last_end_bit + 7) / 8 != in_n - 8) return no("stream end", (long long)((last_end_bit + 7) / 8));
    // CRC-32 by segments of 4 KB from every member's first byte, combined on the host with ONE precomputed operator
    constexpr int SEG_LOG2 = 12;
    const uint64_t seg = 1ull << SEG_LOG2;
    std::vector<CrcMember> cm;
    uint64_t nseg = 0;
    for (const Member &m : members) {
        if (!m.len) continue;                                  // (an empty member has no segment: its CRC is that of nothing)
        cm.push_back(CrcMember{m.at, m.len, nseg});
        nseg += (m.len + seg - 1) >> SEG_LOG2;
    }
    GI(hipMallocAsync((void **)&d_tab, std::max<size_t>(16, cm.size() * sizeof(CrcMember)), st));      // (the member table)
    GI(hipMallocAsync((void **)&d_crc, std::max<uint64_t>(1, nseg) * 4, st));
    std::vector<uint32_t> crcs(std::max<uint64_t>(1, nseg));
    if (nseg) {
        GB(h2d(d_tab, cm.data(), cm.size() * sizeof(CrcMember)));
        hipLaunchKernelGGL(crc_members_kernel, dim3((unsigned)((nseg + 63) / 64)), dim3(64), 0, st, d_text, reinterpret_cast<const CrcMember *>(d_tab), (uint32_t)cm.size(), nseg,
                           SEG_LOG2, d_crc);
        GB(d2h(crcs.data(), d_crc, nseg * 4));
    }
    bool crc_ok = true;
    {
        uint32_t op[32];
        crc_zero_operator(op, SEG_LOG2);
        // the operator as four byte-indexed tables: one application = four lookups (there are 250 segments per MB)
        std::vector<uint32_t> opt(4 * 256);
        for (int byte = 0; byte < 4; byte++)
            for (uint32_t v = 0; v < 256; v++) opt[(size_t)byte * 256 + v] = gf2_times(op, v << (8 * byte));
        uint64_t si = 0;
        for (const Member &m : members) {
            uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
            for (uint64_t a0 = 0; a0 < m.len; a0 += seg, si++) {
                const uint64_t l = std::min<uint64_t>(seg, m.len - a0);
                if (l == seg) crc = opt[crc & 0xFF] ^ opt[256 + ((crc >> 8) & 0xFF)] ^ opt[512 + ((crc >> 16) & 0xFF)] ^ opt[768 + (crc >> 24)] ^ crcs[si];
                else crc = (uint32_t)crc32_combine(crc, crcs[si], (z_off_t)l);
            }
            crc_ok = crc_ok && crc == m.crc;
        }
    }
    lap("crc");
#undef GI
#undef GB
#ifdef SS_GZ_DEBUG_SKIPCRC           // diagnostic builds only (scripts/dev): hand out the text although it is wrong
    if (!crc_ok) fprintf(stderr, "[ginflate] CRC MISMATCH (debug build: text returned)\n");
#else
    if (!crc_ok) return no("crc");
#endif
    cleanup(true);
    g_handled++;
    *text_dev = (char *)d_text;
    *len = total;
    *lease = A;
    return true;
}

}  // namespace ss

// the scratch arenas kept between calls (up to two, ~6 GB each, plus a text buffer) go back to the device
extern "C" int ss_gz_gpu_release(void)
{
    std::vector<Arena *> all;
    {
        std::lock_guard<std::mutex> g(g_arena_mu);
        all.swap(g_arena_free);
    }
    for (Arena *a : all) arena_destroy(a);
    std::vector<PinSet *> pins;
    {
        std::lock_guard<std::mutex> g(g_arena_mu);
        pins.swap(g_pin_free);
    }
    for (PinSet *p : pins) pin_destroy(p);
    std::vector<hipStream_t> streams;
    {
        std::lock_guard<std::mutex> g(g_arena_mu);
        streams.swap(ss::g_stream_free);
    }
    for (hipStream_t q : streams) hipStreamDestroy(q);
    ss::reorder_release();
    ss::big_release();
    return SS_OK;
}

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
    void *lease = nullptr;
    if (!ss::gpu_gunzip(buf.data(), buf.size(), &d, &n, &lease, -1)) return SS_ERANGE;
    char *h = (char *)malloc(std::max<uint64_t>(n, 1));
    if (!h) { ss::gpu_gunzip_done(lease); return SS_ENOMEM; }
    const hipError_t e = n ? hipMemcpy(h, d, n, hipMemcpyDeviceToHost) : hipSuccess;
    ss::gpu_gunzip_done(lease);
    if (e != hipSuccess) { free(h); return SS_EHIP; }
    *text = h;
    *len = n;
    return SS_OK;
}

// A deflated member of a ZIP archive (a .npy array of scipy.sparse.save_npz: all_strains_re.npz, Recls_withR_new.py:110-112)
// inflated on the device and LEFT there: raw deflate data at [off, off + comp_n) of the file, CRC-32 and length of the content
// as the archive's directory states them.  The member is dressed as a gzip member (10-byte header in front, CRC-32 + ISIZE
// behind: the inflater checks both) and goes through gpu_gunzip.  *d_data is lent until ss_npz_member_done(*lease).
// SS_ERANGE: not handled here (no dynamic blocks to enter, too small, an absurd ratio): the caller reads it on the host.
namespace {
struct NpzLease { void *gz = nullptr; void *d_buf = nullptr; };
}
extern "C" int ss_npz_member_dev(const char *path, uint64_t off, uint64_t comp_n, uint32_t crc, uint64_t usize, int method, void **d_data,
                                 uint64_t *n, void **lease)
{
    if (!path || !d_data || !n || !lease || (method != 0 && method != 8) || (method == 8 && comp_n < 64) || (method == 0 && comp_n != usize)) return SS_EINVAL;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return SS_EIO;
    struct stat st;
    if (fstat(fd, &st) != 0 || (uint64_t)st.st_size < off + comp_n) { close(fd); return SS_EIO; }
    NpzLease *L = new (std::nothrow) NpzLease();
    if (!L) { close(fd); return SS_ENOMEM; }
    if (method == 0) {
        // STORED (scipy.sparse.save_npz(..., compressed=False)): the bytes go to the device as they are, through pinned pieces;
        // every piece's CRC-32 is taken on its way (zipfile's single thread spends 7 ms per million non-zeros on exactly this)
        // and the pieces' CRCs are combined in order
        constexpr uint64_t PIECE = 16ull << 20;
        constexpr int T = 8;
        const uint64_t pieces = (usize + PIECE - 1) / PIECE;
        int device = 0;
        hipGetDevice(&device);
        if (hipMalloc(&L->d_buf, std::max<uint64_t>(usize, 64)) != hipSuccess) { (void)hipGetLastError(); delete L; close(fd); return SS_ENOMEM; }
        std::vector<uint32_t> pcrc((size_t)pieces, 0);
        std::atomic<bool> good(true);
        std::atomic<uint64_t> next(0);
        std::vector<std::thread> pool;
        for (int t = 0; t < T && (uint64_t)t < pieces; t++)
            pool.emplace_back([&, t] {
                hipSetDevice(device);
                char *pin = nullptr;
                if (hipHostMalloc((void **)&pin, PIECE, hipHostMallocDefault) != hipSuccess) { good = false; return; }
                hipStream_t sx = ss::ingest_stream((unsigned)t);
                for (uint64_t c; good && (c = next.fetch_add(1)) < pieces;) {
                    const uint64_t o = c * PIECE, m = std::min<uint64_t>(PIECE, usize - o);
                    uint64_t got = 0;
                    while (got < m) {
                        const ssize_t r = pread(fd, pin + got, m - got, (off_t)(off + o + got));
                        if (r <= 0) break;
                        got += (uint64_t)r;
                    }
                    if (got != m) { good = false; break; }
                    pcrc[(size_t)c] = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)pin, (uInt)m);
                    if (hipMemcpyAsync((char *)L->d_buf + o, pin, m, hipMemcpyHostToDevice, sx) != hipSuccess || hipStreamSynchronize(sx) != hipSuccess) good = false;
                }
                hipHostFree(pin);
            });
        for (auto &th : pool) th.join();
        close(fd);
        uLong all = crc32(0L, Z_NULL, 0);
        for (uint64_t c = 0; c < pieces; c++) all = crc32_combine(all, pcrc[(size_t)c], (z_off_t)std::min<uint64_t>(PIECE, usize - c * PIECE));
        if (!good || (uint32_t)all != crc) { hipFree(L->d_buf); delete L; return good ? SS_EIO : SS_EHIP; }
        *d_data = L->d_buf; *n = usize; *lease = L;
        return SS_OK;
    }
    const uint64_t in_n = 10 + comp_n + 8;
    std::unique_ptr<uint8_t[]> img(new (std::nothrow) uint8_t[in_n]);
    if (!img) { close(fd); delete L; return SS_ENOMEM; }
    static const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
    memcpy(img.get(), hdr, 10);
    {   // the compressed bytes, by a few threads (the page cache hands a single pread ~3 GB/s)
        const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(8, comp_n >> 24));
        std::atomic<bool> ok(true);
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < T; w++)
            pool.emplace_back([&, w] {
                uint64_t a = comp_n * w / T;
                const uint64_t e = comp_n * (w + 1) / T;
                while (a < e) {
                    const ssize_t r = pread(fd, img.get() + 10 + a, (size_t)std::min<uint64_t>(e - a, 1u << 30), (off_t)(off + a));
                    if (r <= 0) { ok = false; return; }
                    a += (uint64_t)r;
                }
            });
        for (auto &th : pool) th.join();
        close(fd);
        if (!ok) { delete L; return SS_EIO; }
    }
    uint8_t *t8 = img.get() + 10 + comp_n;
    for (int q = 0; q < 4; q++) { t8[q] = (uint8_t)(crc >> (8 * q)); t8[4 + q] = (uint8_t)(usize >> (8 * q)); }
    char *d = nullptr;
    uint64_t len = 0;
    if (!ss::gpu_gunzip(img.get(), in_n, &d, &len, &L->gz, -1)) { delete L; return SS_ERANGE; }
    if (len != usize) { ss::gpu_gunzip_done(L->gz); delete L; return SS_EIO; }
    *d_data = d; *n = len; *lease = L;
    return SS_OK;
}
extern "C" int ss_npz_member_done(void *lease)
{
    NpzLease *L = static_cast<NpzLease *>(lease);
    if (!L) return SS_OK;
    if (L->gz) ss::gpu_gunzip_done(L->gz);
    if (L->d_buf) hipFree(L->d_buf);
    delete L;
    return SS_OK;
}

// CRC-32 (zlib's) of `prefix` followed by n copies of one byte, from the prefix's CRC: O(log n) crc32_combine steps.  What
// the `data.npy` member of a binary matrix must have -- a few header bytes, then nnz ones -- so that 630 MB of ones need not
// be inflated to know that they are ones (the archive's directory carries the content's CRC-32 and length).
extern "C" int ss_crc32_repeat(uint32_t prefix_crc, int byte, uint64_t n, uint32_t *out)
{
    if (!out || byte < 0 || byte > 255) return SS_EINVAL;
    const unsigned char b = (unsigned char)byte;
    uLong pw = crc32(crc32(0L, Z_NULL, 0), &b, 1);       // CRC of 2^i copies
    uLong acc = prefix_crc;
    uint64_t len = 1;
    for (uint64_t m = n; m; m >>= 1, len <<= 1) {
        if (m & 1) acc = crc32_combine(acc, pw, (z_off_t)len);
        if (m >> 1) pw = crc32_combine(pw, pw, (z_off_t)len);
    }
    *out = (uint32_t)acc;
    return SS_OK;
}

// the pinned upload buffers of `n_files` concurrent .gz inputs (at most two sets are kept), made ahead of time
extern "C" int ss_gz_warm_up(int n_files)
{
    std::vector<PinSet *> got;
    for (int i = 0; i < std::min(n_files, 2); i++) {
        PinSet *p = pin_get();
        if (!p) break;
        got.push_back(p);
    }
    const bool ok = (int)got.size() == std::min(n_files, 2);
    for (PinSet *p : got) pin_put(p);
    return ok ? SS_OK : SS_ENOMEM;
}

extern "C" int ss_gz_set_range(int rank, int world, uint64_t slice_bytes, ss_gz_chain_fn chain, void *user)
{
    if (world < 1 || rank < 0 || rank >= world) return SS_EINVAL;
    std::lock_guard<std::mutex> one(ss::g_range_mu);
    ss::g_range.rank = rank; ss::g_range.world = world; ss::g_range.slice_bytes = slice_bytes; ss::g_range.fn = chain; ss::g_range.user = user;
    return SS_OK;
}

// Test hooks, switched by an explicit call only (never by the environment): 1 = plant a wrong entry in search chunk `value`
// (0 = off), 2 = this process declines .gz inputs on the device (range mode: it still serves the chain), 3 = this rank leaves
// range mode without serving the chain (what a crashed peer looks like: the others' bounded wait must end it), 4 = which scan
// kernel of the page index a table goes through (ss_mini.hip launch_scan_mini: 1 = k = 31 through the per-position kernel, 2 = every
// k through it, 3 = every k through scan_mini_kernel; 0 = the product's choice): the kernels held to each other.
extern "C" int ss_test_hook(int which, long long value)
{
    switch (which) {
    case 1: ss::g_hook_entry = value; return SS_OK;
    case 2: ss::g_hook_decline = value; return SS_OK;
    case 3: ss::g_hook_skip_chain = value; return SS_OK;
    case 4: ss::g_hook_generic_k = value; return SS_OK;
    default: return SS_EINVAL;
    }
}

extern "C" int ss_gz_range_counters(uint64_t *files, uint64_t *pieces)
{
    if (files) *files = ss::g_range_files.load();
    if (pieces) *pieces = ss::g_range_pieces.load();
    return SS_OK;
}
