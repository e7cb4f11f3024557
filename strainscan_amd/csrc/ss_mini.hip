// ss_mini.hip -- minimizer-bucketed k-mer index and its scan kernel (k = 31).
//
// Why: with the flat open-address table (ss_scan.hip) every one of the 120 k-mers of a 150-bp
// read costs a random 64-byte HBM sector although ~95 % of them are not in the database
// (profiles/r01a: 213 GB fetched per 22 GB of algorithmic bytes).  Consecutive k-mers of a read
// share their minimizer (the 15-mer with the smallest ordering key inside the k-mer) for ~9 positions
// on average, so:
//   * database k-mers are stored grouped by minimizer ("buckets"), contiguous in HBM;
//   * a one-probe Bloom filter (<= 4 MB: L2 resident) and behind it a small exact directory
//     minimizer -> bucket (tens of MB) answer "no database k-mer has this minimizer" for most
//     read windows;
//   * a lane probes the directory once per run of equal minimizers (~13 per read instead of
//     120) and touches a bucket only when the minimizer exists.
// Counting semantics are unchanged and exact: a k-mer is looked up in the bucket of ITS OWN
// minimizer, which is a function of the k-mer alone, on the database side and on the read side.
//
// Layout in HBM
//   d_mkeys[n_slots] u64  buckets back to back.  A bucket = 1 header word + its k-mers sorted by
//                         (offset of the minimizer inside the k-mer, rest):
//                           header: bits 0..16 = which offsets occur, bit 17 = some offset occurs more
//                                   than once ("multi"), bits 32..63 = number of k-mers
//                           k-mer : the 62-bit key
//                         With one k-mer per offset (the normal case: a bucket is one super-k-mer)
//                         the k-mer with offset o sits at header + 1 + popcount(mask & ((1 << o) - 1)):
//                         a hit costs two dependent loads, a miss inside an existing bucket one.
//   d_counts[n_slots] u32 occurrences per slot (same index; header slots unused)
//   d_dir[n_dir][2]   u64 cuckoo directory of 16-byte buckets (two entries each, two hash functions).
//                         keyed by the minimizer itself (the 30-bit m-mer).
//                         entry = 14-bit fingerprint (never 0x3FFF) << 50 | multi << 49 |
//                                 offset mask (17 bits) << 32 | moved flag << 31 | bucket start (31 bits)
//                         i.e. the bucket's header travels with the directory entry: a found run goes
//                         straight to its candidate k-mers (dir -> candidate: two dependent round trips
//                         per tile instead of three).  A fingerprint false positive (2^-14) only costs a
//                         candidate compare that fails: k-mers are compared in full, and a k-mer lives in
//                         exactly one bucket, so every fingerprint match is simply tried.  A key lives
//                         in its FIRST bucket unless that was full when it arrived; "moved" on a first
//                         bucket's entry 0 means a key of that bucket lives in its second bucket, which is
//                         then read too (a few % of the lookups; the second bucket is never the first
//                         one again).  EMPTY = ~0.
//   d_bloom[2^b/32]   u32 bit dir_mix(minimizer) >> (32 - b) set for every bucket
#include "ss_common.h"
#include "ss_scan_dev.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <thread>
#include <vector>

namespace ss {

constexpr int MINI_M = 15;                       // minimizer length (30 bits)
constexpr uint32_t M30 = 0x3FFFFFFFu;
constexpr uint32_t HDR_MULTI = 1u << 17;

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

// the two directory buckets of a minimizer (the 30-bit m-mer): 32-bit mixes reduced to [0, n_dir)
// by the high half of a 32x32 product (any table size, no power-of-two rounding).  Once per RUN.
__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
__host__ __device__ __forceinline__ uint32_t dir_mix(uint32_t mini)
{
    uint32_t h = mini * 0x9E3779B1u;
    return h ^ (h >> 15);
}
__host__ __device__ __forceinline__ uint32_t dir_bucket1(uint32_t mini, uint32_t n_dir)
{
    return mulhi32(dir_mix(mini), n_dir);
}
// never the first bucket again (an entry reachable through both would be counted twice): n_dir >= 16
__host__ __device__ __forceinline__ uint32_t dir_bucket2(uint32_t mini, uint32_t n_dir)
{
    const uint32_t b = dir_bucket1(mini, n_dir) + 1u + mulhi32((mini ^ 0x5bd1e995u) * 0x85EBCA6Bu, n_dir - 1u);
    return b >= n_dir ? b - n_dir : b;
}
// 14-bit fingerprint: the LOW bits of the mix (the bucket index uses its high bits), never 0x3FFF so that
// an empty entry (all ones) matches nothing
__host__ __device__ __forceinline__ uint32_t dir_fp_of_mix(uint32_t h) { const uint32_t f = h & 0x3FFFu; return f < 0x3FFEu ? f : 0x3FFEu; }
__host__ __device__ __forceinline__ uint32_t dir_fp(uint32_t mini) { return dir_fp_of_mix(dir_mix(mini)); }
constexpr uint64_t DIR_MOVED = 1ull << 31;        // flag in entry 0 of a first bucket
constexpr uint32_t START_MASK = 0x7FFFFFFFu;

// minimizer (the m-mer itself) of a k-mer and its LEFTMOST offset inside the k-mer
static inline uint32_t mini_of_key(uint64_t key, int k, uint32_t *offset)
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

namespace {

using namespace ss::dev;

struct Ent {
    uint32_t mini;
    uint32_t row;
    uint64_t key;
    uint32_t off;   // offset of the minimizer inside the k-mer (sort key inside the bucket)
    uint32_t part;  // partition of the build sort
};

// ---------------------------------------------------------------------------------------------
// scan kernel, minimizer layout: dense SIMD for the arithmetic, compacted LDS work queues for
// the memory probes.
//
// Measured on MI355X (profiles/r01b..f): what bounds this kernel is not HBM bytes but the
// instructions a wave issues per tile (VALU 82 % busy at 8 waves/SIMD) and the latency chains of
// its three dependent memory probes.  So per tile of 62 x 16 = 992 start positions, ONE wave:
//   phase 0  coalesced 16-byte loads of the bases, 2-bit encode, codes + invalid flags -> LDS
//   phase 1a every lane keys the 16 m-mers that start in its bases (one multiply-add each) and tags
//            them with their index: packed = (key & ~31) | index, kept in registers
//   phase 1b minimizer of the lane's 16 k-mers = suffix minima over the own packed words, prefix
//            minima over the next lane's (read through a DPP wave shift; one v_min_u32 per step
//            decides key AND leftmost position); runs of equal minimizers, merged across lane
//            boundaries, are pushed to LDS queue q1 through a DPP wave prefix sum
//            (one 32-bit entry per run: ~94 per tile instead of 992 positions)
//   phase 2a lanes pull runs from q1: the minimizer m-mer is re-read from the codes, one mix, one
//            probe of the Bloom filter (L2).  The ~10 % that pass are compacted into q1b
//   phase 2b ONE 16-byte directory load per surviving run; runs whose minimizer exists go to q2
//   phase 3  16 lanes per found run, one per position: candidate slot from the run's offset mask
//            -> 64-bit compare -> atomicAdd
// ---------------------------------------------------------------------------------------------
// threads per workgroup of this kernel = one wave
#ifndef SS_NT
#define SS_NT 64
#endif
constexpr int MT = SS_NT;
static_assert(MT == 64, "one wave per workgroup: phase 2 compacts with ballots and keeps its counters in scalar registers");
#ifndef SS_Q1CAP
#define SS_Q1CAP 160
#endif
constexpr int Q1CAP = SS_Q1CAP;    // runs (q1), Bloom survivors (q1b) and found runs (q2) per tile held in LDS (random
                                   // reads: mean 94 runs, max ~110); overflow is handled inline.  With 160 the
                                   // workgroup needs 4.9 KB: 32 one-wave workgroups per CU.  Test builds shrink it.
static_assert(Q1CAP <= 4096, "q2 keeps a q1b index in 12 bits");
// A tile is (MT - 2) x 16 start positions: all MT lanes load 16 bases and key the 16 m-mers that
// START in them; lanes 0..MT-3 own 16 k-mers each, whose 17-m-mer windows end in the NEXT lane's
// m-mers, whose last bases lie in the lane after that.  The last two lanes only feed their
// neighbours: 3 % of the lanes idle is cheaper than a separate halo load + encode, which costs a
// full wave instruction stream for one lane.
constexpr int MLANES = MT - 2;
constexpr int MTILE = MLANES * PPT;

struct QShared {
    uint32_t code[MT + 2];
    uint16_t inv[MT + 2];
    uint32_t q1[Q1CAP + 64];           // run:   len << 12 | tile position of its first k-mer (+64 dump slots)
    alignas(16) uint8_t ib[MT * PPT];  // per tile position: index (0..31, counted from the lane's first m-mer) of the minimizer
    uint64_t q1b[Q1CAP];               // run that passed the Bloom filter: mix << 32 | minimizer offset in the first k-mer << 17 | q1 entry
    uint64_t q2[Q1CAP];                // found: bucket start << 32 | multi << 31 | aligned offset mask << 12 | q1b index
    uint32_t cnt[4];                   // [1] = found runs
#ifdef SS_LDS_PAD
    uint32_t pad[SS_LDS_PAD / 4];      // occupancy experiments only
#endif
};

// the 31-mer starting at tile position pos as two 32-bit halves (funnel shifts; no 64-bit shifts)
__device__ __forceinline__ void kmer_at(const QShared &S, uint32_t pos, uint32_t &lo, uint32_t &hi)
{
    const uint32_t w = pos >> 4, sh = 2 * (pos & 15);
    const uint32_t a = S.code[w], b = S.code[w + 1], c = S.code[w + 2];
    lo = __builtin_amdgcn_alignbit(b, a, sh);
    hi = __builtin_amdgcn_alignbit(c, b, sh) & 0x3FFFFFFFu;
}

// the 15-mer starting at tile position p (one funnel shift over two code words)
__device__ __forceinline__ uint32_t mmer_at(const QShared &S, uint32_t p)
{
    const uint32_t w = p >> 4;
    return __builtin_amdgcn_alignbit(S.code[w + 1], S.code[w], 2 * (p & 15)) & ss::M30;
}

// A found run carries the bucket's offset mask shifted so that the offset of the run's FIRST k-mer
// sits at bit 16: the k-mer q positions further has offset bit 16 - q (the minimizer stands still
// while the k-mer start moves right).  Slot of the database k-mer that k-mer q would be, 0 = none.
__device__ __forceinline__ uint32_t cand_slot(uint32_t bstart, uint32_t amask, uint32_t q)
{
    const uint32_t o = 16u - q;
    return ((amask >> o) & 1u) ? bstart + 1u + (uint32_t)__popc(amask & ((1u << o) - 1u)) : 0u;
}
__device__ __forceinline__ uint32_t aligned_mask(uint32_t hdr, uint32_t o0) { return ((hdr & 0x1FFFFu) << (16u - o0)) & 0x1FFFFu; }

// compare the candidate (already loaded) with k-mer `pos`; count; fall back to a bucket scan when
// several database k-mers share a minimizer offset (repeated / colliding minimizer)
__device__ __forceinline__ void settle_item(const QShared &S, uint32_t pos, uint32_t bstart, uint32_t multi, uint32_t cpos,
                                            uint64_t cand, const uint64_t *__restrict__ mkeys,
                                            uint32_t *__restrict__ counts)
{
    uint32_t klo, khi;
    kmer_at(S, pos, klo, khi);
    if ((uint32_t)cand == klo && (uint32_t)(cand >> 32) == khi) {
        atomicAdd(&counts[cpos], 1u);
    } else if (multi) {
        const uint64_t km = ((uint64_t)khi << 32) | klo;
        const uint32_t cnt = (uint32_t)(mkeys[bstart] >> 32);
        for (uint32_t q = 0; q < cnt; q++)
            if (mkeys[bstart + 1 + q] == km) { atomicAdd(&counts[bstart + 1 + q], 1u); break; }
    }
}

// inclusive prefix sum over the 64 lanes of a wave in the VALU (DPP row shifts + row broadcasts):
// no LDS round trips (ds_bpermute) on the critical path of every tile
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// v_mad_u32_u24 with the addend as a scalar constant (one SGPR per VOP3 on gfx9)
__device__ __forceinline__ uint32_t mad24s(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
// a_i = min(a_i, p_i of the NEXT lane), eight at a time: the DPP wave shift rides in the v_min itself (the
// compiler keeps a separate v_mov_dpp per word).  s_nop: a DPP read needs two wait states after the VALU
// write of its source, which the hazard recognizer cannot see inside inline assembly.  Lane 63 reads 0.
#define SS_MIN_DPP(i, j) "v_min_u32_dpp %" #i ", %" #j ", %" #i " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void min_next_lane8(uint32_t &a0, uint32_t &a1, uint32_t &a2, uint32_t &a3, uint32_t &a4, uint32_t &a5,
                                               uint32_t &a6, uint32_t &a7, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3,
                                               uint32_t p4, uint32_t p5, uint32_t p6, uint32_t p7)
{
    asm("s_nop 1\n\t" SS_MIN_DPP(0, 8) SS_MIN_DPP(1, 9) SS_MIN_DPP(2, 10) SS_MIN_DPP(3, 11) SS_MIN_DPP(4, 12) SS_MIN_DPP(5, 13)
        SS_MIN_DPP(6, 14) SS_MIN_DPP(7, 15)
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7));
}
// m = 2 * m + (a != b): builds a bit mask from comparisons, two instructions per bit
__device__ __forceinline__ uint32_t shift_in_ne(uint32_t m, uint32_t a, uint32_t b)
{
    asm("v_cmp_ne_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(a), "v"(b) : "vcc");
    return m;
}

#ifdef SS_TIMING
// debug build: cycles a wave spends between the phase markers, accumulated in registers and flushed once per block
__device__ unsigned long long ss_timing[32];
#define SS_T(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); t_acc[(i) & 7] += (uint32_t)(now_ - t_prev); t_prev = now_; } while (0)
#else
#define SS_T(i) asm volatile("; SSMARK " #i)
#endif
// debug builds -DSS_STOP_AFTER=n end every tile after phase n (1 = m-mer keys, 2 = runs queued, 3 = directory):
// dynamic instruction counts and times per phase (scripts/gpu_stop.sh); results are then of course wrong
#ifdef SS_STOP_AFTER
#define SS_STOP(n) if (SS_STOP_AFTER == (n)) { __syncthreads(); continue; }
#else
#define SS_STOP(n)
#endif

// SGPRs decide the residency of this kernel: a SIMD admits floor(800 / (ceil(sgprs / 16) * 16 + 16)) waves
// (MI355X_MICROARCH.md, residency), i.e. 8 waves at <= 80, 7 at <= 96, 6 beyond; VGPRs (58) and LDS (4.9 KB per
// one-wave workgroup = 32 per CU) allow 8.
#ifndef SS_NUM_SGPR
#define SS_NUM_SGPR 80
#endif
template <bool ALIGNED, int WAVES_PER_SIMD>
__global__ __launch_bounds__(MT, WAVES_PER_SIMD) __attribute__((amdgpu_num_sgpr(SS_NUM_SGPR))) void scan_mini_kernel(
    const uint8_t *__restrict__ bases, uint64_t n, uint64_t n_tiles, const uint64_t *__restrict__ mkeys,
    const uint64_t *__restrict__ dir, uint32_t n_dir, uint32_t *__restrict__ counts,
    const uint32_t *__restrict__ bloom, uint32_t bloom_shift)
{
    constexpr int K = 31;                            // 17 m-mers of length 15 per k-mer
    static_assert(K - ss::MINI_M + 1 == PPT + 1, "a k-mer window = own m-mers j..15 + neighbour m-mers 0..j");
    __shared__ QShared S;
    const int t = threadIdx.x;
    const ulonglong2 *dir2 = reinterpret_cast<const ulonglong2 *>(dir);
    const uint32_t vc1 = ss::MMK_C1;

    // the 16 bases of this lane are fetched one tile AHEAD: the HBM
    // round trip of the stream overlaps the previous tile's phases
    uint32_t wn[4];
    if (blockIdx.x < n_tiles) {
        const uint64_t b0 = (uint64_t)blockIdx.x * MTILE;
        load16<ALIGNED>(bases, b0 + (uint64_t)t * 16, n, wn);
    }
#ifdef SS_TIMING
    unsigned long long t_prev = __builtin_readcyclecounter();
    uint32_t t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // ---- phase 0: bases -> 2-bit codes in LDS ------------------------------------------------
        {
            uint32_t code, inv;
            encode16(wn, code, inv);
            S.code[t] = code;
            S.inv[t] = (uint16_t)inv;
            if (t == 1) S.cnt[1] = 0;
#ifdef SS_LDS_PAD
        if (n == 1) S.pad[t] = 1;
#endif
            const uint64_t nt = tile + gridDim.x;
            if (nt < n_tiles) {
                const uint64_t nb = nt * (uint64_t)MTILE;
                load16<ALIGNED>(bases, nb + (uint64_t)t * 16, n, wn);
            }
        }
        __syncthreads();
        SS_T(0);

        // ---- phase 1a: key the 16 m-mers that start in this lane's 16 bases -----------------------
        uint32_t hm[PPT];
        {
            const uint32_t c0 = S.code[t], c1 = S.code[t + 1];
            // low 24 bits of the m-mers 0..15 (bits 2i.. of c1:c0): 15 funnel shifts, then one multiply-add
            // per m-mer; the "+ i" of the packed word rides in the additive constant
            hm[0] = mad24s(c0, vc1, ss::MMK_C0);
#pragma unroll
            for (int i = 1; i < PPT; i++) hm[i] = mad24s(__builtin_amdgcn_alignbit(c1, c0, 2 * i), vc1, ss::MMK_C0 + (uint32_t)i);
        }
        SS_T(1);
        SS_STOP(1)

        // ---- phase 1b: minimizer of the lane's 16 k-mers, runs ------------------------------------
        // k-mer j is live iff none of the bases j..j+30 is invalid: flags of the positions 0..30 in lo
        // (smeared downwards through a bit reversal: shift-left-or is one instruction), 31..46 in hi
        // (smeared upwards; k-mer j sees hi bits 0..j-1)
        uint32_t live = 0;
        if (t < MLANES) {
            uint32_t lo;
            __builtin_memcpy(&lo, &S.inv[t], 4);
            uint32_t hi = __builtin_amdgcn_alignbit(S.inv[t + 2], lo, 31);
            lo = __builtin_bitreverse32(lo & 0x7FFFFFFFu);
            lo |= lo << 1; lo |= lo << 2; lo |= lo << 4; lo |= lo << 8; lo |= lo << 16;
            hi |= hi << 1; hi |= hi << 2; hi |= hi << 4; hi |= hi << 8;
            live = ~(__builtin_bitreverse32(lo) | (hi << 1)) & 0xFFFFu;
        }
        // k-mer j covers m-mers j..j+16 = own m-mers j..15 (index j..15) and the next lane's m-mers 0..j
        // (index 16..16+j): suffix minima over the own packed words, prefix minima over the neighbour's;
        // equal keys resolve to the smaller index = the leftmost m-mer (the database side uses the same
        // rule).  Every lane computes the prefix minima of ITS keys as its left neighbour counts them
        // (index + 16); the neighbour reads them with a DPP wave shift inside the v_min: the keys never
        // go through LDS.  All lanes execute this (a DPP source lane must be enabled).
        uint32_t need = 0;
        uint32_t mh[PPT];
        {
            uint32_t pf[PPT];
            pf[0] = hm[0] + 16u;
#pragma unroll
            for (int j = 1; j < PPT; j++) pf[j] = min(pf[j - 1], hm[j] + 16u);
#pragma unroll
            for (int i = PPT - 2; i >= 0; i--) hm[i] = min(hm[i], hm[i + 1]);         // suffix minima, in place
            min_next_lane8(hm[0], hm[1], hm[2], hm[3], hm[4], hm[5], hm[6], hm[7], pf[0], pf[1], pf[2], pf[3], pf[4], pf[5], pf[6], pf[7]);
            min_next_lane8(hm[8], hm[9], hm[10], hm[11], hm[12], hm[13], hm[14], hm[15], pf[8], pf[9], pf[10], pf[11], pf[12], pf[13], pf[14], pf[15]);
#pragma unroll
            for (int j = 0; j < PPT; j++) mh[j] = hm[j];
        }
        if (live) {
            // run starts as a bit mask: position j starts a run if it is live and (j == 0, or j-1 is
            // not live, or the minimizer changed)
            uint32_t chg = 0u;
#pragma unroll
            for (int j = PPT - 1; j >= 1; j--) chg = shift_in_ne(chg, mh[j], mh[j - 1]);   // bit j-1 <- (mh[j] != mh[j-1])
            chg = (chg << 1) | 1u;
            need = live & (chg | (~live << 1));
            // the minimizer's index (low byte of the packed word) of all 16 positions: one 16-byte store;
            // phase 2 reads the byte of a run's first position (the walk below no longer gathers it)
            uint32_t pk[4];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const uint32_t lo = __builtin_amdgcn_perm(mh[4 * w + 1], mh[4 * w], 0x0c0c0400u);
                const uint32_t hi = __builtin_amdgcn_perm(mh[4 * w + 3], mh[4 * w + 2], 0x0c0c0400u);
                pk[w] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            }
            reinterpret_cast<uint4 *>(S.ib)[t] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
        SS_T(6);
        // A run that reaches the end of a lane goes on in the next lane when that lane's first k-mer has
        // the same minimizer (same packed word: the left lane counts the right lane's m-mer i as 16 + i).
        // The left lane then owns the whole run (at most 16 k-mers: phase 3 gives a run 16 lanes; a longer
        // one stays cut) and the right lane drops its first run: a third fewer runs per tile.
        const uint32_t stop = need | (~live & 0xFFFFu) | 0x10000u;
        uint32_t ext12 = 0;                          // k-mers my last run takes over from the next lane, << 12
        {
            const uint32_t first = (uint32_t)__ffs(stop >> 1);                                   // length of the run starting at 0
            const uint32_t ownlast = (live >> 15) ? (uint32_t)__clz(need) - 15u : 0u;            // 16 - start of my last run
            const uint32_t l_mh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mh[PPT - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1
            const uint32_t l_own = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ownlast, 0x138, 0xf, 0xf, false);
            const bool cont = (live & 1u) && l_own && l_mh == mh[0] + 16u && l_own + first <= 16u;
            const uint32_t give = cont ? first << 12 : 0u;
            ext12 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0x130, 0xf, 0xf, false);                         // wave_shl:1
            if (cont) need &= ~1u;
        }
        // wave prefix sum of the run counts; then every lane walks ITS runs (about two, at most a handful:
        // the loop is as long as the busiest lane of the wave) instead of testing all 16 positions
        uint32_t ovf = 0;   // runs that did not fit q1 (processed inline below)
        uint32_t n1;
        {
            const uint32_t mine = (uint32_t)__popc(need);
            const uint32_t incl = wave_inclusive_sum(mine);
            n1 = min((uint32_t)__builtin_amdgcn_readlane((int)incl, 63), (uint32_t)Q1CAP);
            const uint32_t mybase = incl - mine;
            if (need) {
                const uint32_t dummy = (uint32_t)Q1CAP + (uint32_t)t;   // where entries beyond the capacity go
                const uint32_t stop1 = stop >> 1, tbase = (uint32_t)(t * PPT);
                uint32_t idx = mybase;
                for (uint32_t nd = need; nd; idx++) {
                    const uint32_t j = (uint32_t)__ffs(nd) - 1u;
                    nd &= nd - 1u;
                    const uint32_t len = (uint32_t)__ffs(stop1 >> j);                       // until the next run / gap
                    S.q1[min(idx, dummy)] = (len << 12) + (tbase + j + (nd ? 0u : ext12));
                }
                if (mybase + mine > (uint32_t)Q1CAP) {                          // rare: which of my runs did not fit
                    uint32_t r = mybase;
                    for (uint32_t nd = need; nd; nd &= nd - 1u, r++)
                        if (r >= (uint32_t)Q1CAP) ovf |= nd & (0u - nd);
                }
            }
        }
        __syncthreads();
        SS_T(2);
        SS_STOP(2)

        // every directory entry whose fingerprint matches becomes a found run in q2 (phase 3 gives it
        // 16 lanes, one per position).  meta = minimizer offset in the first k-mer << 17 | len << 12 | position
        auto push_found = [&](uint64_t de, uint32_t meta, uint32_t ib, bool queued) {
            const uint32_t bstart = (uint32_t)de & ss::START_MASK, hdr = (uint32_t)(de >> 32) & 0x3FFFFu;
            const uint32_t amask = aligned_mask(hdr, (meta >> 17) & 31u), multi = (hdr >> 17) & 1u;
            uint32_t i2 = Q1CAP;
            if (queued) i2 = atomicAdd(&S.cnt[1], 1u);
            if (i2 < Q1CAP) {
                S.q2[i2] = ((uint64_t)bstart << 32) | (multi << 31) | (amask << 12) | ib;
            } else {
                // the queue is full (only with floods of fingerprint collisions), or phase 3 is already
                // over (runs that overflowed q1): settle this run here, so that no k-mer is ever dropped
                const uint32_t rpos = meta & 0xFFFu, len = (meta >> 12) & 31u;
                for (uint32_t q = 0; q < len; q++) {
                    const uint32_t cpos = cand_slot(bstart, amask, q);
                    if (cpos) settle_item(S, rpos + q, bstart, multi, cpos, mkeys[cpos], mkeys, counts);
                }
            }
        };
        // the directory entries (first bucket b1, second bucket behind the "moved" flag) that match the run
        auto lookup_found = [&](const ulonglong2 &b1, uint32_t h, uint32_t meta, uint32_t ib, bool queued) {
            const uint32_t fp = ss::dir_fp_of_mix(h);
            // matching entries as a bit mask + selects (an indexed local array would live in scratch memory);
            // 32-bit compares on the upper halves: a fingerprint is never 0x3FFF, so empty entries match nothing
            ulonglong2 b2 = make_ulonglong2(ss::EMPTY_KEY, ss::EMPTY_KEY);
            if (((uint32_t)b1.x & (uint32_t)ss::DIR_MOVED) && (uint32_t)(b1.x >> 32) != 0xFFFFFFFFu)   // a key of this bucket moved
                b2 = dir2[ss::dir_bucket2(mmer_at(S, (meta & 0xFFFu) + ((meta >> 17) & 31u)), n_dir)];
            uint32_t hits = (uint32_t)(((uint32_t)(b1.x >> 32) >> 18) == fp) | (uint32_t)(((uint32_t)(b1.y >> 32) >> 18) == fp) << 1 |
                            (uint32_t)(((uint32_t)(b2.x >> 32) >> 18) == fp) << 2 | (uint32_t)(((uint32_t)(b2.y >> 32) >> 18) == fp) << 3;
            while (hits) {
                const uint32_t d = (uint32_t)__ffs(hits) - 1u;
                hits &= hits - 1u;
                push_found(d == 0 ? b1.x : d == 1 ? b1.y : d == 2 ? b2.x : b2.y, meta, ib, queued);
            }
        };
        // minimizer offset (in the run's first k-mer) from the index byte of the run's first position
        auto run_meta = [&](uint32_t e) {
            const uint32_t rpos = e & 0xFFFu;
            return e | ((((uint32_t)S.ib[rpos] & 31u) - (rpos & 15u)) << 17);
        };

        SS_T(7);
        // ---- phase 2a: Bloom filter, all lanes busy: one probe of an L2-resident bit array kills most of
        // the ~90 % of the runs whose minimizer is not in the database before they cost a random HBM
        // sector each.  Survivors are compacted into q1b (ballot + lane count: one wave per workgroup)
        uint32_t ns = 0;
        {
#ifndef SS_RPL
#define SS_RPL 2
#endif
            constexpr int RPL = SS_RPL;
            for (uint32_t r0 = 0; r0 < n1; r0 += RPL * MT) {
                uint32_t meta[RPL], hs[RPL], bw[RPL];
                bool ok[RPL];
#pragma unroll
                for (int u = 0; u < RPL; u++) {
                    const uint32_t r = r0 + u * MT + t;
                    ok[u] = r < n1;
                    meta[u] = run_meta(S.q1[ok[u] ? r : 0u]);
                    hs[u] = ss::dir_mix(mmer_at(S, (meta[u] & 0xFFFu) + (meta[u] >> 17)));   // one mix per run: Bloom bit, bucket, fingerprint
                }
                if (bloom) {
#pragma unroll
                    for (int u = 0; u < RPL; u++) bw[u] = bloom[hs[u] >> (bloom_shift + 5)];
#pragma unroll
                    for (int u = 0; u < RPL; u++) ok[u] = ok[u] && ((bw[u] >> ((hs[u] >> bloom_shift) & 31u)) & 1u);
                }
#pragma unroll
                for (int u = 0; u < RPL; u++) {
                    const uint64_t pass = __ballot(ok[u]);
                    if (ok[u]) S.q1b[ns + __builtin_amdgcn_mbcnt_hi((uint32_t)(pass >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pass, 0u))] =
                        ((uint64_t)hs[u] << 32) | meta[u];
                    ns += (uint32_t)__popcll(pass);
                }
            }
        }
        __syncthreads();
        // ---- phase 2b: one 16-byte directory load per surviving run (about a tenth of the runs) ----
        for (uint32_t i0 = 0; i0 < ns; i0 += MT) {
            const uint32_t i = i0 + t;
            if (i < ns) {
                const uint64_t sv = S.q1b[i];
                const uint32_t h = (uint32_t)(sv >> 32);
                lookup_found(dir2[ss::mulhi32(h, n_dir)], h, (uint32_t)sv, i, true);
            }
        }
        __syncthreads();
        SS_T(3);
        SS_STOP(3)

        // ---- phase 3: the k-mers whose minimizer exists in the database -------------------------
        // 16 lanes per found run (one per position of the run), four runs per lane per round: all
        // candidate loads in flight before any compare
        {
#ifndef SS_U3
#define SS_U3 4
#endif
            constexpr int U3 = SS_U3;
            const uint32_t n2 = min(S.cnt[1], (uint32_t)Q1CAP);
            for (uint32_t g0 = 0; g0 < n2 * 16u; g0 += U3 * MT) {
                uint32_t pos[U3], bst[U3], mul[U3], cps[U3];
                uint64_t cnd[U3];
#pragma unroll
                for (int u = 0; u < U3; u++) {
                    const uint32_t g = g0 + u * MT + t, q = g & 15u;
                    const bool v = (g >> 4) < n2;
                    const uint64_t r = S.q2[v ? (g >> 4) : 0u];
                    const uint32_t run = (uint32_t)S.q1b[(uint32_t)r & 0xFFFu];
                    pos[u] = (run & 0xFFFu) + q;
                    bst[u] = (uint32_t)(r >> 32);
                    mul[u] = (uint32_t)r >> 31;
                    cps[u] = (v && q < ((run >> 12) & 31u)) ? cand_slot(bst[u], ((uint32_t)r >> 12) & 0x1FFFFu, q) : 0u;
                    cnd[u] = mkeys[cps[u]];
                }
#pragma unroll
                for (int u = 0; u < U3; u++)
                    if (cps[u]) settle_item(S, pos[u], bst[u], mul[u], cps[u], cnd[u], mkeys, counts);
            }
        }
        // ---- overflow: runs that did not fit q1 (pathological inputs only) are done in place ------
        if (ovf) {
            for (int j = 0; j < PPT; j++) {
                if (!((ovf >> j) & 1u)) continue;
                const uint32_t len12 = ((uint32_t)__ffs(stop >> (j + 1)) << 12) + ((need >> (j + 1)) ? 0u : ext12);
                const uint32_t meta = run_meta(len12 + (uint32_t)(t * PPT + j));
                const uint32_t x = mmer_at(S, (meta & 0xFFFu) + (meta >> 17));
                lookup_found(dir2[ss::dir_bucket1(x, n_dir)], ss::dir_mix(x), meta, 0u, false);
            }
        }
        SS_T(4);
        __syncthreads();   // queues and codes are rewritten by the next tile
        SS_T(5);
    }
#ifdef SS_TIMING
    if (t == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&ss_timing[i], (unsigned long long)t_acc[i]);
#endif
}

void parallel_for(unsigned nthreads, uint64_t n, const std::function<void(uint64_t, uint64_t, unsigned)> &fn)
{
    if (nthreads <= 1 || n < 65536) { fn(0, n, 0); return; }
    std::vector<std::thread> pool;
    const uint64_t per = (n + nthreads - 1) / nthreads;
    for (unsigned w = 0; w < nthreads; w++) {
        const uint64_t lo = std::min<uint64_t>(n, per * w), hi = std::min<uint64_t>(n, lo + per);
        if (lo >= hi) break;
        pool.emplace_back(fn, lo, hi, w);
    }
    for (auto &th : pool) th.join();
}

}  // namespace

namespace ss {

// Host build of the minimizer index.  Fills db->d_mkeys / d_dir / d_counts / d_slot_of_row /
// d_row_valid and n_distinct; returns SS_EKEY for an un-owned k-mer when upper_keys == 0.
int build_mini(ss_db *db, const uint64_t *keys, const uint8_t *flags, uint64_t n_rows, int upper_keys)
{
    const int k = db->k;
    static const bool trace = getenv("SS_BUILD_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[build] %-28s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    constexpr int PB = 8, NP = 1 << PB;
    unsigned nthreads = std::min<unsigned>(ss::host_cpus(), 32u);
    // 1. entries of valid rows with their minimizer
    std::vector<uint64_t> pos(n_rows + 1, 0);
    for (uint64_t i = 0; i < n_rows; i++) pos[i + 1] = pos[i] + ((flags[i] & SS_ROW_VALID) ? 1 : 0);
    const uint64_t nv = pos[n_rows];
    std::vector<Ent> ents(nv), sorted(nv);
    parallel_for(nthreads, n_rows, [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t i = lo; i < hi; i++)
            if (flags[i] & SS_ROW_VALID) {
                uint32_t o;
                const uint32_t mx = mini_of_key(keys[i], k, &o);
                ents[pos[i]] = Ent{mx, (uint32_t)i, keys[i], o, dir_mix(mx) >> (32 - PB)};
            }
    });
    lap("1 minimizers");
    // 2. counting partition on 8 mixed bits of the minimizer, then per-partition sort
    std::vector<uint64_t> pcount(NP + 1, 0);
    for (uint64_t i = 0; i < nv; i++) pcount[ents[i].part + 1]++;
    for (int p = 0; p < NP; p++) pcount[p + 1] += pcount[p];
    {
        std::vector<uint64_t> cur(pcount.begin(), pcount.end() - 1);
        for (uint64_t i = 0; i < nv; i++) sorted[cur[ents[i].part]++] = ents[i];
    }
    ents.clear();
    ents.shrink_to_fit();
    {
        std::atomic<int> next(0);
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < nthreads; w++)
            pool.emplace_back([&] {
                for (int p; (p = next.fetch_add(1)) < NP;)
                    std::sort(sorted.begin() + pcount[p], sorted.begin() + pcount[p + 1], [](const Ent &a, const Ent &b) {
                        if (a.mini != b.mini) return a.mini < b.mini;
                        if (a.off != b.off) return a.off < b.off;
                        if (a.key != b.key) return a.key < b.key;
                        return a.row < b.row;
                    });
            });
        for (auto &th : pool) th.join();
    }
    lap("2 partition + sort");
    // 3. distinct k-mers, buckets (header + entries), row bookkeeping (dict overwrite: the last
    //    allowed row owns the count).  A minimizer lives in one partition, so the partitions are independent:
    //    count their slots and buckets, prefix-sum, fill in parallel (same order as a serial walk: the image
    //    does not depend on the thread count)
    struct Bkt { uint32_t first, second, hdr; };         // minimizer (m-mer), header slot, offset mask | multi
    std::vector<uint64_t> p_slots(NP + 1, 0), p_bkts(NP + 1, 0);
    auto for_partitions = [&](const std::function<void(int)> &fn) {
        std::atomic<int> next(0);
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < nthreads; w++)
            pool.emplace_back([&] { for (int p; (p = next.fetch_add(1)) < NP;) fn(p); });
        for (auto &th : pool) th.join();
    };
    for_partitions([&](int p) {
        uint64_t nb = 0, nd = 0;
        for (uint64_t i = pcount[p]; i < pcount[p + 1];) {
            uint64_t e = i;
            while (e < pcount[p + 1] && sorted[e].mini == sorted[i].mini) e++;
            nb++;
            for (uint64_t a2 = i; a2 < e;) {
                uint64_t b2 = a2;
                while (b2 < e && sorted[b2].key == sorted[a2].key) b2++;
                nd++;
                a2 = b2;
            }
            i = e;
        }
        p_slots[p + 1] = nd + nb;
        p_bkts[p + 1] = nb;
    });
    for (int p = 0; p < NP; p++) { p_slots[p + 1] += p_slots[p]; p_bkts[p + 1] += p_bkts[p]; }
    if (p_slots[NP] >= 0x7FFFFFF0ull) return SS_ERANGE;
    std::vector<uint64_t> mkeys(p_slots[NP]);
    std::vector<Bkt> buckets(p_bkts[NP]);
    std::vector<uint32_t> slot_of_row(std::max<uint64_t>(1, n_rows), SS_NO_SLOT);
    std::vector<uint8_t> row_valid(std::max<uint64_t>(1, n_rows), 0);
    std::atomic<uint64_t> orphans_a(0);
    for_partitions([&](int p) {
        uint64_t ms = p_slots[p], bi = p_bkts[p], orph = 0;
        for (uint64_t i = pcount[p]; i < pcount[p + 1];) {
            // one bucket = all entries with this minimizer
            uint64_t e = i;
            while (e < pcount[p + 1] && sorted[e].mini == sorted[i].mini) e++;
            const uint32_t hslot = (uint32_t)ms++;
            uint32_t mask = 0, multi = 0, cnt = 0;
            for (uint64_t a2 = i; a2 < e;) {
                uint64_t b2 = a2;
                int64_t owner = -1;
                while (b2 < e && sorted[b2].key == sorted[a2].key) {
                    const uint32_t r = sorted[b2].row;
                    if (upper_keys == 1 || !(flags[r] & SS_ROW_LOWER)) owner = r;   // rows ascend within equal k-mers
                    b2++;
                }
                const uint32_t o = sorted[a2].off;
                if ((mask >> o) & 1u) multi = HDR_MULTI;
                mask |= 1u << o;
                const uint32_t slot = (uint32_t)ms;
                mkeys[ms++] = sorted[a2].key;
                for (uint64_t q = a2; q < b2; q++) slot_of_row[sorted[q].row] = slot;
                if (owner >= 0) row_valid[owner] = 1;
                else orph++;
                cnt++;
                a2 = b2;
            }
            mkeys[hslot] = ((uint64_t)cnt << 32) | multi | mask;
            buckets[bi++] = Bkt{sorted[i].mini, hslot, multi | mask};
            i = e;
        }
        orphans_a += orph;
    });
    const uint64_t orphans = orphans_a.load(), n_distinct = p_slots[NP] - p_bkts[NP];
    sorted.clear();
    sorted.shrink_to_fit();
    if (orphans && upper_keys == 0) return SS_EKEY;
    if (mkeys.size() >= 0xFFFFFFF0ull) return SS_ERANGE;
    db->n_distinct = n_distinct;
    db->n_slots = std::max<uint64_t>(1, mkeys.size());
    lap("3 buckets");
    // cuckoo directory of 2-entry buckets, ~0.67 keys per bucket (1/3 load).  Keys prefer their first
    // bucket; a key that ends up in its second bucket sets DIR_MOVED on its first one.
    if (mkeys.size() >= 0x7FFFFFF0ull) return SS_ERANGE;
    uint64_t n_dir = std::max<uint64_t>(16, buckets.size() + buckets.size() / 2);
    std::vector<uint64_t> dir;
    std::vector<uint32_t> dir_h;      // minimizer of each occupied slot (needed to re-place evicted keys)
    std::vector<uint8_t> moved;
    for (;; n_dir += n_dir / 4) {
        if (n_dir >= 0xFFFFFFF0ull) return SS_ERANGE;
        dir.assign(2 * n_dir, EMPTY_KEY);
        dir_h.assign(2 * n_dir, 0);
        moved.assign(n_dir, 0);
        bool ok = true;
        uint64_t rng = 0x9E3779B97F4A7C15ull;
        auto place = [&](uint64_t cur, uint32_t h) -> bool {
            for (int kicks = 0; kicks < 1000; kicks++) {
                const uint32_t b1 = dir_bucket1(h, (uint32_t)n_dir), b2 = dir_bucket2(h, (uint32_t)n_dir);
                for (uint64_t sl : {2 * (uint64_t)b1, 2 * (uint64_t)b1 + 1})
                    if (dir[sl] == EMPTY_KEY) { dir[sl] = cur; dir_h[sl] = h; return true; }
                moved[b1] = 1;                                     // from now on look in b2 as well
                for (uint64_t sl : {2 * (uint64_t)b2, 2 * (uint64_t)b2 + 1})
                    if (dir[sl] == EMPTY_KEY) { dir[sl] = cur; dir_h[sl] = h; return true; }
                rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
                const uint64_t victim = 2 * (uint64_t)((rng & 2) ? b1 : b2) + (rng & 1);
                std::swap(cur, dir[victim]);                       // evict; the victim is re-placed next round
                std::swap(h, dir_h[victim]);
            }
            return false;
        };
        for (const auto &b : buckets) {
            const uint64_t e = ((uint64_t)dir_fp(b.first) << 50) | ((uint64_t)b.hdr << 32) | b.second;
            if (!place(e, b.first)) { ok = false; break; }
        }
        if (ok) break;
    }
    for (uint64_t b = 0; b < n_dir; b++)
        if (moved[b]) {
            if (dir[2 * b] == EMPTY_KEY) std::swap(dir[2 * b], dir[2 * b + 1]);   // keep the flag carrier in entry 0
            if (dir[2 * b] != EMPTY_KEY) dir[2 * b] |= DIR_MOVED;
            // a flagged bucket cannot be empty: keys only move on from FULL buckets and evictions swap
        }
    const uint32_t dirbits = 0;
    db->n_dir = (uint32_t)n_dir;
    db->dirbits = dirbits;
    db->n_buckets = buckets.size();
    db->capacity = db->n_slots;
    lap("cuckoo directory");
    // 4. upload
    const uint64_t nr = std::max<uint64_t>(1, n_rows);
    SS_HIP(hipMalloc((void **)&db->d_mkeys, db->n_slots * sizeof(uint64_t)));
    SS_HIP(hipMalloc((void **)&db->d_dir, dir.size() * sizeof(uint64_t)));
    SS_HIP(hipMalloc((void **)&db->d_counts, db->n_slots * sizeof(uint32_t)));
    SS_HIP(hipMalloc((void **)&db->d_slot_of_row, nr * sizeof(uint32_t)));
    SS_HIP(hipMalloc((void **)&db->d_row_valid, nr));
    db->device_bytes = db->n_slots * 12 + dir.size() * 8 + nr * 5;
    if (!mkeys.empty())
        SS_HIP(hipMemcpy(db->d_mkeys, mkeys.data(), mkeys.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    else
        SS_HIP(hipMemset(db->d_mkeys, 0xFF, sizeof(uint64_t)));
    SS_HIP(hipMemcpy(db->d_dir, dir.data(), dir.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    SS_HIP(hipMemset(db->d_counts, 0, db->n_slots * sizeof(uint32_t)));
    {
        // about 8 bits per minimizer, at most 2^25 bits = 4 MB (the L2 of one XCD): measured on the
        // 25 M-row table 2^23: 4.71 ms, 2^25: 4.60 ms, 2^27: 5.29 ms, none: 6.0 ms.  SS_BLOOM_BITS=0 disables.
        int bits = 10;
        while (bits < 25 && (1ull << bits) < 8 * (uint64_t)buckets.size()) bits++;
        const char *bb = getenv("SS_BLOOM_BITS");
        if (bb) bits = atoi(bb);
        if (bits >= 10 && bits <= 30) {
            std::vector<uint32_t> bloom((size_t)1 << (bits - 5), 0);
            for (const auto &b : buckets) {
                const uint32_t h = dir_mix(b.first) >> (32 - bits);
                bloom[h >> 5] |= 1u << (h & 31u);
            }
            SS_HIP(hipMalloc((void **)&db->d_bloom, bloom.size() * 4));
            SS_HIP(hipMemcpy(db->d_bloom, bloom.data(), bloom.size() * 4, hipMemcpyHostToDevice));
            db->bloom_bits = (uint32_t)bits;
            db->device_bytes += bloom.size() * 4;
        }
    }
    SS_HIP(hipMemcpy(db->d_slot_of_row, slot_of_row.data(), nr * sizeof(uint32_t), hipMemcpyHostToDevice));
    SS_HIP(hipMemcpy(db->d_row_valid, row_valid.data(), nr, hipMemcpyHostToDevice));
    lap("4 bloom + upload");
    return SS_OK;
}

template <int LB>
static void launch_lb(bool aligned, unsigned blocks, hipStream_t stream, const uint8_t *bases, uint64_t n,
                      uint64_t n_tiles, ss_db *db)
{
    if (aligned)
        hipLaunchKernelGGL((scan_mini_kernel<true, LB>), dim3(blocks), dim3(MT), 0, stream, bases, n, n_tiles,
                           db->d_mkeys, db->d_dir, db->n_dir, db->d_counts, db->d_bloom, 32u - db->bloom_bits);
    else
        hipLaunchKernelGGL((scan_mini_kernel<false, LB>), dim3(blocks), dim3(MT), 0, stream, bases, n, n_tiles,
                           db->d_mkeys, db->d_dir, db->n_dir, db->d_counts, db->d_bloom, 32u - db->bloom_bits);
}

#ifdef SS_TIMING
extern "C" int ss_debug_timing(unsigned long long *out32, int reset)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out32, HIP_SYMBOL(ss_timing), 256);
    if (reset) { unsigned long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(ss_timing), z, 256); }
    return 0;
}
#endif

int launch_scan_mini(ss_db *db, const void *bases_dev, uint64_t n, hipStream_t stream, unsigned blocks,
                     uint64_t n_tiles)
{
    const bool aligned = (((uintptr_t)bases_dev) & 15) == 0;
    static int lb = -1, bpc = 0;
    if (lb < 0) {   // tuning knobs for A/B measurements: register budget and blocks per CU
        const char *e = getenv("SS_MINI_LB");
        lb = e ? atoi(e) : 4;
        const char *g = getenv("SS_MINI_BLOCKS_PER_CU");
        bpc = g ? atoi(g) : 0;
    }
    n_tiles = (n + MTILE - 1) / MTILE;                      // this kernel's tile is 62 x 16 positions
    // grid-stride over tiles with MANY more blocks than fit the chip: short blocks start at scattered times, so
    // the waves sharing a SIMD stop marching through their ALU and memory phases in step.  Measured with 8 waves
    // per SIMD resident (20 M reads = 3.04 M tiles; blocks = x * 1024): x = 8 (one round of resident blocks)
    // 4.03 ms, 32: 3.76, 128: 3.59, 512: 3.55, 2048 (1.5 tiles per block): 3.50, 4096 (one tile each): 3.51
    blocks = (unsigned)std::min<uint64_t>(n_tiles, (uint64_t)(bpc > 0 ? bpc : 2048) * 256 * (256 / MT));
    const uint8_t *b = (const uint8_t *)bases_dev;
    switch (lb) {
    case 3: launch_lb<3>(aligned, blocks, stream, b, n, n_tiles, db); break;
    case 5: launch_lb<5>(aligned, blocks, stream, b, n, n_tiles, db); break;
    case 6: launch_lb<6>(aligned, blocks, stream, b, n, n_tiles, db); break;
    case 7: launch_lb<7>(aligned, blocks, stream, b, n, n_tiles, db); break;
    default: launch_lb<4>(aligned, blocks, stream, b, n, n_tiles, db); break;
    }
    SS_HIP(hipGetLastError());
    return SS_OK;
}

}  // namespace ss

// ---------------------------------------------------------------------------------------------
// Index image on disk: the built minimizer index (device arrays) dumped verbatim, so that a
// database is indexed once, not at every run (device image cache).
// ---------------------------------------------------------------------------------------------
namespace {
struct ImageHeader {
    char magic[8];          // "SSIDX06\0"
    int32_t k, layout;
    uint64_t n_rows, n_distinct, n_slots, n_buckets;
    uint32_t n_dir, bloom_bits;
};

bool write_dev(FILE *f, const void *d, uint64_t bytes)
{
    std::vector<char> buf(std::min<uint64_t>(bytes, 64ull << 20));
    for (uint64_t off = 0; off < bytes; off += buf.size()) {
        const uint64_t n = std::min<uint64_t>(buf.size(), bytes - off);
        if (hipMemcpy(buf.data(), (const char *)d + off, n, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (fwrite(buf.data(), 1, n, f) != n) return false;
    }
    return true;
}

// A file range straight to device memory: four threads pread() 16 MB pieces into pinned buffers and copy them on
// the shared ingest streams (one pageable 64 MB bounce buffer moved the 0.54 GB image of an E. coli database in
// 0.09 s: more than reading the sample).
struct PinnedReaders {
    static constexpr int T = 4;
    static constexpr uint64_t PIECE = 16ull << 20;
    char *buf[T] = {nullptr, nullptr, nullptr, nullptr};
    bool ok = true;
    PinnedReaders()
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; t++) pool.emplace_back([this, t] { if (hipHostMalloc((void **)&buf[t], PIECE, hipHostMallocDefault) != hipSuccess) buf[t] = nullptr; });
        for (auto &th : pool) th.join();
        for (int t = 0; t < T; t++) ok = ok && buf[t] && ss::ingest_stream((unsigned)t);
    }
    ~PinnedReaders() { for (int t = 0; t < T; t++) if (buf[t]) hipHostFree(buf[t]); }
    bool read(int fd, uint64_t file_off, void *d, uint64_t bytes)
    {
        if (!ok) return false;
        int device = 0;
        hipGetDevice(&device);
        std::atomic<bool> good(true);
        std::atomic<uint64_t> next(0);
        const uint64_t pieces = (bytes + PIECE - 1) / PIECE;
        std::vector<std::thread> pool;
        for (int t = 0; t < T && (uint64_t)t < pieces; t++)
            pool.emplace_back([&, t] {
                hipSetDevice(device);
                hipStream_t st = ss::ingest_stream((unsigned)t);
                for (uint64_t c; good && (c = next.fetch_add(1)) < pieces;) {
                    const uint64_t off = c * PIECE, n = std::min<uint64_t>(PIECE, bytes - off);
                    uint64_t got = 0;
                    while (got < n) {
                        const ssize_t r = pread(fd, buf[t] + got, n - got, (off_t)(file_off + off + got));
                        if (r <= 0) break;
                        got += (uint64_t)r;
                    }
                    if (got != n || hipMemcpyAsync((char *)d + off, buf[t], n, hipMemcpyHostToDevice, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess)
                        good = false;
                }
            });
        for (auto &th : pool) th.join();
        return good;
    }
};
}  // namespace

extern "C" {

int ss_db_export(const ss_db *db, const char *path)
{
    if (!db || !path) return SS_EINVAL;
    if (db->layout != 1) return SS_ERANGE;          // only the minimizer layout has a build worth caching
    FILE *f = fopen(path, "wb");
    if (!f) return SS_EIO;
    ImageHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "SSIDX06", 8);
    h.k = db->k; h.layout = db->layout;
    h.n_rows = db->n_rows; h.n_distinct = db->n_distinct; h.n_slots = db->n_slots; h.n_buckets = db->n_buckets;
    h.n_dir = db->n_dir;
    h.bloom_bits = db->d_bloom ? db->bloom_bits : 0;
    const uint64_t nr = std::max<uint64_t>(1, db->n_rows);
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && write_dev(f, db->d_mkeys, db->n_slots * 8) &&
              write_dev(f, db->d_dir, (uint64_t)db->n_dir * 16) && write_dev(f, db->d_slot_of_row, nr * 4) &&
              write_dev(f, db->d_row_valid, nr) &&
              (!h.bloom_bits || write_dev(f, db->d_bloom, (1ull << h.bloom_bits) / 8));
    ok = (fclose(f) == 0) && ok;
    if (!ok) { remove(path); return SS_EIO; }
    return SS_OK;
}

int ss_db_import(const char *path, ss_db **out)
{
    if (!path || !out) return SS_EINVAL;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return SS_EIO;
    ImageHeader h;
    struct stat st;
    if (fstat(fd, &st) != 0 || pread(fd, &h, sizeof(h), 0) != (ssize_t)sizeof(h) || memcmp(h.magic, "SSIDX06", 8) != 0 ||
        h.layout != 1 || h.k != 31 || h.n_slots == 0 || h.n_dir == 0 || (h.bloom_bits && (h.bloom_bits < 10 || h.bloom_bits > 30))) {
        close(fd);
        return SS_EINVAL;
    }
    const uint64_t nr = std::max<uint64_t>(1, h.n_rows);
    const uint64_t sizes[5] = {h.n_slots * 8, (uint64_t)h.n_dir * 16, nr * 4, nr, h.bloom_bits ? (1ull << h.bloom_bits) / 8 : 0};
    uint64_t offs[6] = {sizeof(h), 0, 0, 0, 0, 0};
    for (int i = 0; i < 5; i++) offs[i + 1] = offs[i] + sizes[i];
    if ((uint64_t)st.st_size != offs[5]) { close(fd); return SS_EIO; }     // the file must be exactly the image
    ss_db *db = new (std::nothrow) ss_db();
    if (!db) { close(fd); return SS_ENOMEM; }
    db->k = h.k; db->layout = 1;
    db->n_rows = h.n_rows; db->n_distinct = h.n_distinct; db->n_slots = h.n_slots; db->capacity = h.n_slots;
    db->n_buckets = h.n_buckets; db->n_dir = h.n_dir;
    hipGetDevice(&db->device);
    bool ok = hipMalloc((void **)&db->d_mkeys, sizes[0]) == hipSuccess && hipMalloc((void **)&db->d_dir, sizes[1]) == hipSuccess &&
              hipMalloc((void **)&db->d_counts, db->n_slots * 4) == hipSuccess &&
              hipMalloc((void **)&db->d_slot_of_row, sizes[2]) == hipSuccess && hipMalloc((void **)&db->d_row_valid, sizes[3]) == hipSuccess &&
              (!h.bloom_bits || hipMalloc((void **)&db->d_bloom, sizes[4]) == hipSuccess);
    if (ok) {
        PinnedReaders rd;
        ok = rd.read(fd, offs[0], db->d_mkeys, sizes[0]) && rd.read(fd, offs[1], db->d_dir, sizes[1]) &&
             rd.read(fd, offs[2], db->d_slot_of_row, sizes[2]) && rd.read(fd, offs[3], db->d_row_valid, sizes[3]) &&
             (!h.bloom_bits || rd.read(fd, offs[4], db->d_bloom, sizes[4])) &&
             hipMemset(db->d_counts, 0, db->n_slots * 4) == hipSuccess;
    }
    close(fd);
    if (ok && h.bloom_bits) db->bloom_bits = h.bloom_bits;
    if (!ok) { ss_db_destroy(db); return SS_EIO; }
    db->device_bytes = db->n_slots * 12 + (uint64_t)db->n_dir * 16 + nr * 5 + sizes[4];
    *out = db;
    return SS_OK;
}

}  // extern "C"
