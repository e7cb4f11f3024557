// ss_mini.hip -- minimizer-paged k-mer index and its scan kernel (k = 31).
//
// Why: with the flat open-address table (ss_scan.hip) every one of the 120 k-mers of a 150-bp
// read costs a random 64-byte HBM sector although ~95 % of them are not in the database
// (profiles/r01a: 213 GB fetched per 22 GB of algorithmic bytes).  Consecutive k-mers of a read
// share their minimizer (the 15-mer with the smallest ordering key inside the k-mer) for ~9 positions
// on average, so a lane probes the index once per RUN of equal minimizers (~13 per read instead of
// 120).  Counting semantics are unchanged and exact: a k-mer is looked up under ITS OWN minimizer,
// which is a function of the k-mer alone, on the database side and on the read side.
//
// What bounds the lookups (scripts/micro/randsec.hip on MI355X): random 64-byte sectors come at
// ~55 G/s = 3.5 TB/s from anything larger than the L2 (the 256 MB Infinity Cache does not help), at
// 250 G/s out of the 4 MB L2 of an XCD, and reading all 64 bytes of a sector costs the same as
// reading 16.  So one run = ONE sector, and that sector answers as much as possible:
//
// Layout in HBM
//   d_pages[n_pages]  64 B  hash-addressed pages of eight 8-byte slots.  A minimizer (the 30-bit m-mer x)
//                         lives in page floor(h * n_pages / 2^30), h = mix30(x) a BIJECTION of the 30 bits, or,
//                         when that page was full at build time, in the next one(s): a lookup goes on to
//                         the next page only if the page it read is full (slot 7 occupied), which at the
//                         build's load (two items per page on average) is one lookup in a thousand.  No
//                         wrap-around (spare pages follow the last home page), and the build grows the table
//                         until every run of full pages is shorter than n_pages / 1024 -- the distance between
//                         the home pages of two minimizers that share their tag bits -- so a lookup can never
//                         reach another minimizer's slots with the same tag.
//                         A slot is 8 bytes spread over four arrays of the page (byte layout: PG_SLOTS below).
//                         inline k-mer:  tag20 | e5 | flank32
//                              the database k-mer itself: tag20 = low 20 bits of h (with the page number they
//                              determine h, hence x, exactly: two minimizers that agree in the low 20 bits of h
//                              lie >= 2^20 apart, i.e. >= n_pages / 1024 pages apart); e = 16 - o, o =
//                              offset of the minimizer inside the k-mer; flank32 = the other 16 bases (those
//                              behind the minimizer in the low bits, then those in front of it).  A read k-mer
//                              equals it iff tag, offset and flank agree: no second load, no key array.  Its
//                              counter is d_counts[n_mslots + 8 * page + slot].
//                         bucket reference:  tag14 | mask17 | multi | start30
//                              minimizers with more than SS_INLINE_MAX database k-mers keep them in a bucket of
//                              d_mkeys (below); the bucket's header travels in the slot.  tag14 is only a
//                              filter: candidates are compared in full.
//   d_mkeys[n_mslots] u64 buckets back to back.  A bucket = 1 header word + its k-mers sorted by
//                         (offset of the minimizer inside the k-mer, rest):
//                           header: bits 0..16 = which offsets occur, bit 17 = some offset occurs more
//                                   than once ("multi"), bits 32..63 = number of k-mers
//                           k-mer : the 62-bit key
//                         With one k-mer per offset (the normal case: a bucket is one super-k-mer)
//                         the k-mer with offset o sits at header + 1 + popcount(mask & ((1 << o) - 1)).
//   d_counts[n_slots] u32 occurrences: [0, n_mslots) parallel to d_mkeys, [n_mslots, n_mslots + 8 n_pages)
//                         parallel to the page slots
//   d_bloom[2^b/32]   u32 bit h >> (30 - b) set for every minimizer -- only built when it is worth its L2
//                         round trip: >= 4 bits per minimizer within 2^25 bits (the 4 MB L2 of an XCD).  A database
//                         of dense node sets (few minimizers, ~9 k-mers each) gets one; a database of sampled
//                         node sets (Build_tree.py:590-591: nearly one minimizer per k-mer) does not -- there
//                         half of a sample's runs find their minimizer anyway.
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
#include <type_traits>
#include <memory>
#include <vector>

namespace ss {

// HDR_MULTI, page_of, the page layout constants, flank_of_key, mini_of_key: ss_scan_dev.h (shared with ss_build_dev.hip)
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
// Measured on MI355X (profiles/): what bounds this kernel is not HBM bytes but, depending on the
// database, the instructions a wave issues per tile (dense node sets: VALU 100 % busy at 8 waves/SIMD) or the
// number of random 64-byte sectors it asks of the memory system (sampled node sets: 54 G/s, all there is).
// So per tile of 62 x 16 = 992 start positions, ONE wave, no barriers that wait for another wave:
//   phase 0  coalesced 16-byte loads of the bases, 2-bit encode, codes + invalid flags -> LDS
//   phase 1a every lane keys the 16 m-mers that start in its bases (one multiply-add each) and tags
//            them with their index: packed = (key & ~31) | index, kept in registers
//   phase 1b minimizer of the lane's 16 k-mers = suffix minima over the own packed words, prefix
//            minima over the next lane's (read through a DPP wave shift; one v_min_u32 per step
//            decides key AND leftmost position); runs of equal minimizers, merged across lane
//            boundaries, are pushed to LDS queue q1 through a DPP wave prefix sum
//            (one 32-bit entry per run: ~94 per tile instead of 992 positions)
//   phase 2a (databases with a Bloom filter) lanes pull runs from q1: the minimizer m-mer is re-read from the
//            codes, one mix, one probe of the Bloom filter (L2).  The ~10 % that pass are compacted into q1b
//   phase 2b ONE 16-byte load per run (all runs, or the Bloom survivors): the head of the minimizer's page.
//            Inline k-mers are settled here (tag + offset + flank = the whole k-mer -> atomicAdd); bucket
//            references go to q2
//   phase 3  16 lanes per found bucket, one per position: candidate slot from the run's offset mask
//            -> 64-bit compare -> atomicAdd
// Round 4, template switches of the same kernel:
//   COMB     tables that expect hits (ss_db_expect_hits: a layer-2 cluster table) scanned by a BINNED read set: a workgroup
//            takes four consecutive tiles, hits are added up in an LDS table keyed by the bucket (QComb) and flushed with one
//            atomic per non-zero counter; found runs of SOLID buckets (PG_SOLID) are verified by one lane each (phase 3a),
//            the others by 16 lanes (3b)
//   MULTI    up to four tables in one pass (ss_scan_reads_multi): phases 0-1b once per tile, phases 2-3 per table
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
// ... and with k at run time (KK = 0, 17 <= k <= 30): a k-mer of k - 14 m-mers changes its minimizer every (k - 13) / 2 positions, so a
// tile has ~2 x 992 / (k - 13) runs -- 165 at k = 25, 198 at k = 23, 248 at k = 21.  256: the most the combining table's byte-sized run
// indices (QComb::ent / rest) can name; 7.2 KB per workgroup with it (22 one-wave workgroups per CU).
#ifndef SS_Q1CAP_RT
#define SS_Q1CAP_RT 256
#endif
constexpr int Q1CAP_RT = SS_Q1CAP_RT > 256 ? 256 : SS_Q1CAP_RT;
// A tile is (MT - 2) x 16 start positions: all MT lanes load 16 bases and key the 16 m-mers that
// START in them; lanes 0..MT-3 own 16 k-mers each, whose 17-m-mer windows end in the NEXT lane's
// m-mers, whose last bases lie in the lane after that.  The last two lanes only feed their
// neighbours: 3 % of the lanes idle is cheaper than a separate halo load + encode, which costs a
// full wave instruction stream for one lane.
constexpr int MLANES = MT - 2;
constexpr int MTILE = MLANES * PPT;

template <int CAP>
struct QSharedT {
    uint32_t code_[MT + 3];            // code_[1 + i] = bases 16 i .. 16 i + 15 of the tile; one word of slack in front (the
                                       // window of 16 bases BEFORE a minimizer near the tile start) and two behind
    uint16_t inv[MT + 2];
    uint32_t q1[CAP + 64];             // run:   len << 12 | tile position of its first k-mer (+64 dump slots)
    alignas(16) uint8_t ib[MT * PPT];  // per tile position: index (0..31, counted from the lane's first m-mer) of the minimizer
    uint64_t q2[CAP];                  // found bucket: bucket start << 32 | multi << 31 | aligned offset mask << 12 | run index (q1b with a
                                       // Bloom filter, q1 without)
    uint32_t cnt[4];                   // [1] = found runs
#ifdef SS_LDS_PAD
    uint32_t pad[SS_LDS_PAD / 4];      // occupancy experiments only
#endif
};
// (a __shared__ variable of its own: instantiations without a Bloom filter do not pay its 1.25 KB)
template <int CAP>
struct QBloomT {
    uint64_t q1b[CAP];                 // run that passed the Bloom filter: h << 32 | minimizer offset in the first k-mer << 17 | q1 entry
};

// ---------------------------------------------------------------------------------------------
// Counting where most read k-mers HIT (layer-2 cluster tables hold every k-mer of their strains; Vote_Strain_L2_Lasso_new_sp.py
// :354-372 scans all reads against one per identified cluster).  What a global atomicAdd costs on this chip
// (scripts/micro/atomics.hip, profiles/r04_atomics_micro_*.txt): 27 G (wave instruction, 64-byte line) pairs per second,
// whatever the footprint (2 MB or 40 MB), the scope, the number of lanes on the line (1..16) -- and the kernel's other work does
// not overlap it.  A read's 120 hits fall into ~13 buckets = ~19 lines, a tile of 6.5 reads pays ~100 line requests: 11 of the
// 17 ms of such a scan.  In a BINNED read set (ss_reorder.hip) the neighbours of a read come from the same locus and hit the same
// buckets, so the hits are first added up in LDS: a small open-address table keyed by the bucket, one byte per bucket slot
// (a run adds at most one to a slot, and the table is flushed before 255 runs have gone into it), flushed with one global
// atomic per non-zero counter every few tiles.  A workgroup works through CONSECUTIVE tiles for that.
// ---------------------------------------------------------------------------------------------
constexpr int COMB_NE = 64;            // entries (= lanes of the wave: the flush looks at one entry per lane)
constexpr int COMB_CH = 4;             // consecutive tiles per workgroup
constexpr uint32_t COMB_NONE = 0xFFu;
template <int CAP>
struct QCombT {
    static_assert(CAP <= 256, "run indices in bytes");
    uint32_t key[COMB_NE];             // bucket start + 1, 0 = free
    uint32_t nrun[COMB_NE];            // found runs that took this entry since the last flush (a byte counter holds 255)
    uint32_t acc[COMB_NE][5];          // byte o = occurrences of the bucket's slot o (1..19; slot 0 is the header)
    uint8_t ent[CAP];                  // per found run of the tile: its entry, COMB_NONE = count in global memory
    uint8_t list[COMB_NE];             // flush: the occupied entries
    uint8_t rest[CAP];                 // phase 3: the found runs whose bucket is not solid, from the front; solid runs whose hits
                                       // go straight to the counters, from the back (indices into q2)
};

// the 31-mer starting at tile position pos as two 32-bit halves (funnel shifts; no 64-bit shifts)
// (khi_mask: the 2 k - 32 bits of the upper half: 0x3FFFFFFF at k = 31)
template <class QS>
__device__ __forceinline__ void kmer_at(const QS &S, uint32_t pos, uint32_t &lo, uint32_t &hi, uint32_t khi_mask = 0x3FFFFFFFu)
{
    const uint32_t w = pos >> 4, sh = 2 * (pos & 15);
    const uint32_t a = S.code_[w + 1], b = S.code_[w + 2], c = S.code_[w + 3];
    lo = __builtin_amdgcn_alignbit(b, a, sh);
    hi = __builtin_amdgcn_alignbit(c, b, sh) & khi_mask;
}

// the 15-mer starting at tile position p (one funnel shift over two code words)
template <class QS>
__device__ __forceinline__ uint32_t mmer_at(const QS &S, uint32_t p)
{
    const uint32_t w = p >> 4;
    return __builtin_amdgcn_alignbit(S.code_[w + 2], S.code_[w + 1], 2 * (p & 15)) & ss::M30;
}
// the 16 bases starting at tile position p (p >= -16)
template <class QS>
__device__ __forceinline__ uint32_t win16_at(const QS &S, int32_t p)
{
    const int32_t w = (p >> 4) + 1;
    return __builtin_amdgcn_alignbit(S.code_[w + 1], S.code_[w], 2 * (p & 15));
}

// A found run carries the bucket's offset mask shifted so that the offset of the run's FIRST k-mer
// sits at bit 16: the k-mer q positions further has offset bit 16 - q (the minimizer stands still
// while the k-mer start moves right).  Slot of the database k-mer that k-mer q would be, 0 = none.
// (F = k - 15, the largest minimizer offset: 16 at k = 31, where the comments above are written)
__device__ __forceinline__ uint32_t cand_slot(uint32_t bstart, uint32_t amask, uint32_t q, uint32_t F = 16u)
{
    const uint32_t o = F - q;
    return ((amask >> o) & 1u) ? bstart + 1u + (uint32_t)__popc(amask & ((1u << o) - 1u)) : 0u;
}
__device__ __forceinline__ uint32_t aligned_mask(uint32_t hdr, uint32_t o0, uint32_t F = 16u)
{
    const uint32_t wm = (2u << F) - 1u;
    return ((hdr & wm) << (F - o0)) & wm;
}

// one occurrence of the database k-mer in slot `cpos` of the bucket at `bstart`: into the LDS entry `ent` of the bucket when it
// has one (QComb), else straight to the counter
template <bool COMB>
__device__ __forceinline__ void count_slot(uint32_t cpos, uint32_t bstart, uint32_t ent, uint32_t *__restrict__ acc5,
                                           uint32_t *__restrict__ counts)
{
    const uint32_t off = cpos - bstart;
    if (COMB && ent != COMB_NONE && off < 20u) atomicAdd(&acc5[ent * 5u + (off >> 2)], 1u << ((off & 3u) * 8u));
    else atomicAdd(&counts[cpos], 1u);
}
// compare the candidate (already loaded) with k-mer `pos`; count; fall back to a bucket scan when
// several database k-mers share a minimizer offset (repeated / colliding minimizer)
template <bool COMB, class QS>
__device__ __forceinline__ void settle_item(const QS &S, uint32_t pos, uint32_t bstart, uint32_t multi, uint32_t cpos,
                                            uint64_t cand, const uint64_t *__restrict__ mkeys,
                                            uint32_t *__restrict__ counts, uint32_t ent = COMB_NONE, uint32_t *__restrict__ acc5 = nullptr,
                                            uint32_t khi_mask = 0x3FFFFFFFu)
{
    uint32_t klo, khi;
    kmer_at(S, pos, klo, khi, khi_mask);
    if ((uint32_t)cand == klo && (uint32_t)(cand >> 32) == khi) {
        count_slot<COMB>(cpos, bstart, ent, acc5, counts);
    } else if (multi) {
        const uint64_t km = ((uint64_t)khi << 32) | klo;
        const uint32_t cnt = (uint32_t)(mkeys[bstart] >> 32);
        for (uint32_t q = 0; q < cnt; q++)
            if (mkeys[bstart + 1 + q] == km) { count_slot<COMB>(bstart + 1 + q, bstart, ent, acc5, counts); break; }
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

// found runs of a probe launch (choose_comb), 64 words so that the tiles' single atomics do not queue on one address
__device__ uint32_t ss_probe_runs[64];

#ifdef SS_COMB_STATS
// debug build: what the combining scan did -- [0] tiles, [1] flushes, [2] entries flushed, [3] counters flushed, [4] found runs,
// [5] runs without an entry, [6] flushes because the table was full; and the trip counts of every scan's loops (for the
// instruction budget, scripts/archive/r4/isa_budget.py): [8] tiles, [9] runs queued (n1), [10] runs looked up (ns: all, or the Bloom
// filter's survivors), [11] found runs (n2), [12] lookup rounds, [13] candidate rounds
__device__ unsigned long long ss_comb_stats[16];
#define SS_CS(i, v) do { if (t == 0) atomicAdd(&ss_comb_stats[i], (unsigned long long)(v)); } while (0)
#else
#define SS_CS(i, v)
#endif
#define SS_MARK(i) asm volatile("; SSMARK " #i)      // a label in the ISA only (scripts/archive/r4/isa_budget.py)
#ifdef SS_TIMING
// debug build: cycles a wave spends between the phase markers, accumulated in registers and flushed once per block
__device__ unsigned long long ss_timing[32];
#define SS_T(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); t_acc[(i) & 15] += (uint32_t)(now_ - t_prev); t_prev = now_; } while (0)
#else
#define SS_T(i) asm volatile("; SSMARK " #i)
#endif
// debug builds -DSS_STOP_AFTER=n end every tile after phase n (1 = m-mer keys, 2 = runs queued, 3 = page lookups):
// dynamic instruction counts and times per phase (scripts/gpu_stop.sh); results are then of course wrong
#ifdef SS_STOP_AFTER
#define SS_STOP(n) if (SS_STOP_AFTER == (n)) { __syncthreads(); continue; }
#else
#define SS_STOP(n)
#endif

// Several tables in ONE pass over the reads (layer 2: the reference re-reads the FASTQ once per identified cluster,
// Vote_Strain_L2_Lasso_new_sp.py:295-296,354-372): bases, codes, minimizers and runs of a tile are made once, the page lookups
// and candidate checks repeat per table.  Up to four tables per launch (the LDS counters keep the table in the two bits above
// a bucket start's thirty).
constexpr int MULTI_MAX = 4;
struct ScanTabs {
    const uint64_t *mkeys[MULTI_MAX];
    const uint4 *pages[MULTI_MAX];
    uint32_t *counts[MULTI_MAX];
    uint32_t n_pages[MULTI_MAX], cbase[MULTI_MAX];
    int n;
};

// SGPRs decide the residency of this kernel: a SIMD admits floor(800 / (ceil(sgprs / 16) * 16 + 16)) waves
// (MI355X_MICROARCH.md, residency), i.e. 8 waves at <= 80, 7 at <= 96, 6 beyond; VGPRs (58) and LDS (4.9 KB per
// one-wave workgroup = 32 per CU) allow 8.
// Minimizers of a lane's 16 k-mers when a k-mer has W < 17 m-mers (k < 31; the run-time-k instantiations, KK = 0): the window of
// k-mer j = packed keys x[j .. j + W - 1] of the 32 the lane sees (its own 16, the next lane's 16).  A window of 17 is "own suffix +
// neighbour's prefix" (the k = 31 network below); a shorter one may lie inside the lane's own 16, so: minima over 2, 4, 8(, 16)
// consecutive keys by doubling, and every window = two overlapping power-of-two stretches.  W is a template argument here (the indices
// must be constants: the keys live in registers), the kernel switches on it once per tile.
template <int W>
__device__ __forceinline__ void sliding_min(const uint32_t (&x)[32], uint32_t (&out)[PPT])
{
    constexpr int P = W >= 16 ? 16 : W >= 8 ? 8 : W >= 4 ? 4 : 2;
    uint32_t m[32];
#pragma unroll
    for (int i = 0; i < 32; i++) m[i] = x[i];
#pragma unroll
    for (int s2 = 1; s2 < P; s2 <<= 1) {
#pragma unroll
        for (int i = 0; i + s2 < 32; i++) m[i] = min(m[i], m[i + s2]);      // (ascending i: m[i + s2] is still the narrower minimum)
    }
#pragma unroll
    for (int j = 0; j < PPT; j++) out[j] = min(m[j], m[j + W - P]);
}

#ifndef SS_NUM_SGPR
#define SS_NUM_SGPR 80
#endif
// KK: 31 = the k this kernel was tuned for, everything about k a constant; 0 = k is the kernel argument k_rt (17 <= k <= 30, round 6)
template <bool ALIGNED, bool BLOOM, bool COMB, int WAVES_PER_SIMD, bool MULTI = false, int KK = 31>
__global__ __launch_bounds__(MT, WAVES_PER_SIMD) __attribute__((amdgpu_num_sgpr(SS_NUM_SGPR))) void scan_mini_kernel(
    const uint8_t *__restrict__ bases, uint64_t n, uint64_t n_tiles, const uint64_t *__restrict__ mkeys0,
    const uint4 *__restrict__ pages0, uint32_t n_pages0, uint32_t *__restrict__ counts0, uint32_t cbase0,
    const uint32_t *__restrict__ bloom, uint32_t bloom_shift, uint32_t xcd_swizzle, const ScanTabs tabs, int k_rt)
{
    static_assert(!(MULTI && BLOOM), "several tables: no minimizer filter (they are tables that expect hits)");
    static_assert(KK == 31 || KK == 0, "k = 31, or k at run time");
    // the table of this pass over the tile's runs (MULTI: tabs.* in turn)
    const uint64_t *__restrict__ mkeys = mkeys0;
    const uint4 *__restrict__ pages = pages0;
    uint32_t *__restrict__ counts = counts0;
    uint32_t n_pages = n_pages0, cbase = cbase0, tab_key = 0u;
    const int K = KK ? KK : k_rt;                    // k = 31: 17 m-mers of length 15 per k-mer
    const uint32_t W = (uint32_t)(K - ss::MINI_M + 1), F = W - 1u;      // m-mers per k-mer, largest minimizer offset (= flank bases)
    const uint32_t khi_mask = (1u << (2 * K - 32)) - 1u;                // upper half of a 2 k-bit key
    const uint32_t fmask = F >= 16u ? 0xFFFFFFFFu : (1u << (2u * F)) - 1u;      // a flank's 2 F bits
    static_assert(31 - ss::MINI_M + 1 == PPT + 1, "k = 31: a k-mer window = own m-mers j..15 + neighbour m-mers 0..j");
    constexpr int QC = KK == 31 ? Q1CAP : Q1CAP_RT;  // runs a tile's queues hold
    __shared__ QSharedT<QC> S;
    __shared__ QBloomT<QC> SB;                       // (dropped from the instantiations that never touch it)
    __shared__ QCombT<QC> C;
    const int t = threadIdx.x;
    const uint32_t vc1 = ss::MMK_C1;
    constexpr uint64_t CH = COMB ? COMB_CH : 1;      // consecutive tiles per workgroup

    // the 16 bases of this lane are fetched one tile AHEAD: the HBM
    // round trip of the stream overlaps the previous tile's phases
    // Workgroup b runs on XCD b % 8 (observed; a speed matter only): with the swizzle, XCD x works through the x-th
    // eighth of the TILES in order (its workgroups stride through that eighth), so neighbouring tiles -- reads that a
    // resident set keeps binned by their first minimizer (ss_reorder.hip) and that share their page sectors -- meet in
    // ONE 4 MB L2 instead of eight.  (The grid is a multiple of 8 then; every XCD gets the same number of tiles.)
    // A workgroup takes CH consecutive tiles, then the next CH `stride` chunks further on (CH = 1: a plain grid stride)
    uint64_t tile = (uint64_t)blockIdx.x * CH, tile_end = n_tiles, stride = gridDim.x, tile0 = 0;
    if (xcd_swizzle & 1u) {
        const uint64_t per = (((n_tiles + 7) >> 3) + CH - 1) / CH * CH, x = blockIdx.x & 7u;
        stride = gridDim.x >> 3;
        tile0 = x * per;
        tile = tile0 + (uint64_t)(blockIdx.x >> 3) * CH;
        tile_end = min(n_tiles, (x + 1) * per);
    }
    auto next_tile = [&](uint64_t tl) -> uint64_t {
        if (CH == 1) return tl + stride;
        return ((tl + 1 - tile0) % CH) ? tl + 1 : tl + 1 + (stride - 1) * CH;
    };
#ifdef SS_TIMING
    unsigned long long t_prev = __builtin_readcyclecounter();
    uint32_t t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    uint32_t comb_runs = 0;                          // found runs added to the LDS counters since their last flush
    if (COMB) {
        C.key[t] = 0u;
        C.nrun[t] = 0u;
#pragma unroll
        for (int w = 0; w < 5; w++) C.acc[t][w] = 0u;
    }
    // the non-zero LDS counters go to global memory (16 lanes per occupied entry: slots 1..16, then a lane per entry for the
    // slots 17..19), the table is emptied
    auto comb_flush = [&]() {
        const uint32_t key = C.key[t];
        const uint64_t occ = __ballot(key != 0u);
        const uint32_t nocc = (uint32_t)__popcll(occ);
        SS_CS(1, 1); SS_CS(2, nocc);
        if (key) C.list[__builtin_amdgcn_mbcnt_hi((uint32_t)(occ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)occ, 0u))] = (uint8_t)t;
        __syncthreads();
        for (uint32_t g0 = 0; g0 < nocc * 16u; g0 += MT) {
            const uint32_t g = g0 + (uint32_t)t;
            if ((g >> 4) < nocc) {
                const uint32_t e = C.list[g >> 4], off = 1u + (g & 15u);
                const uint32_t c = (C.acc[e][off >> 2] >> ((off & 3u) * 8u)) & 0xFFu;
#ifdef SS_COMB_STATS
                { const unsigned long long nz = __popcll(__ballot(c != 0u)); SS_CS(3, nz); }
#endif
                if (c) {
                    const uint32_t k1 = C.key[e] - 1u;
                    uint32_t *cb = MULTI ? tabs.counts[k1 >> 30] : counts;
                    atomicAdd(&cb[(k1 & ss::START_MASK) + off], c);
                }
            }
        }
        if (key) {
            const uint32_t w = C.acc[t][4] >> 8;
            if (w) {
                uint32_t *cb = MULTI ? tabs.counts[(key - 1u) >> 30] : counts;
#pragma unroll
                for (uint32_t o = 0; o < 3; o++)
                    if ((w >> (8u * o)) & 0xFFu) atomicAdd(&cb[((key - 1u) & ss::START_MASK) + 17u + o], (w >> (8u * o)) & 0xFFu);
            }
        }
        __syncthreads();
        if (key) {
            C.key[t] = 0u;
            C.nrun[t] = 0u;
#pragma unroll
            for (int w = 0; w < 5; w++) C.acc[t][w] = 0u;
        }
        __syncthreads();
        comb_runs = 0;
    };
    uint32_t wn[4];
    if (tile < tile_end) {
        const uint64_t b0 = tile * MTILE;
        load16<ALIGNED>(bases, b0 + (uint64_t)t * 16, n, wn);
    }
    for (; tile < tile_end; tile = next_tile(tile)) {
        SS_MARK(20);
        // ---- phase 0: bases -> 2-bit codes in LDS ------------------------------------------------
        {
            uint32_t code, inv;
            encode16(wn, code, inv);
            S.code_[t + 1] = code;
            S.inv[t] = (uint16_t)inv;
            if (t == 1) S.cnt[1] = 0;
#ifdef SS_LDS_PAD
        if (n == 1) S.pad[t] = 1;
#endif
            const uint64_t nt = next_tile(tile);
            if (nt < tile_end) {
                const uint64_t nb = nt * (uint64_t)MTILE;
                load16<ALIGNED>(bases, nb + (uint64_t)t * 16, n, wn);
            }
        }
        __syncthreads();
        SS_T(0);

        // ---- phase 1a: key the 16 m-mers that start in this lane's 16 bases -----------------------
        uint32_t hm[PPT];
        {
            const uint32_t c0 = S.code_[t + 1], c1 = S.code_[t + 2];
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
        if (KK == 31 && t < MLANES) {
            uint32_t lo;
            __builtin_memcpy(&lo, &S.inv[t], 4);
            uint32_t hi = __builtin_amdgcn_alignbit(S.inv[t + 2], lo, 31);
            lo = __builtin_bitreverse32(lo & 0x7FFFFFFFu);
            lo |= lo << 1; lo |= lo << 2; lo |= lo << 4; lo |= lo << 8; lo |= lo << 16;
            hi |= hi << 1; hi |= hi << 2; hi |= hi << 4; hi |= hi << 8;
            live = ~(__builtin_bitreverse32(lo) | (hi << 1)) & 0xFFFFu;
        } else if (KK != 31 && t < MLANES) {
            // any k: a sliding OR of width k over the 48 flags from the lane's first base on = width 16 by doubling, then two
            // stretches of 16 that overlap (17 <= k <= 31)
            uint64_t y = (uint64_t)S.inv[t] | ((uint64_t)S.inv[t + 1] << 16) | ((uint64_t)S.inv[t + 2] << 32);
            y |= y >> 1; y |= y >> 2; y |= y >> 4; y |= y >> 8;
            y |= y >> (K - 16);
            live = ~(uint32_t)y & 0xFFFFu;
        }
        // k-mer j covers m-mers j..j+16 = own m-mers j..15 (index j..15) and the next lane's m-mers 0..j
        // (index 16..16+j): suffix minima over the own packed words, prefix minima over the neighbour's;
        // equal keys resolve to the smaller index = the leftmost m-mer (the database side uses the same
        // rule).  Every lane computes the prefix minima of ITS keys as its left neighbour counts them
        // (index + 16); the neighbour reads them with a DPP wave shift inside the v_min: the keys never
        // go through LDS.  All lanes execute this (a DPP source lane must be enabled).
        uint32_t need = 0;
        uint32_t mh[PPT];
        if (KK == 31) {
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
        } else {
            // any k: the next lane's 16 keys through DPP wave shifts (index + 16, as the left lane counts them), then the window minima
            uint32_t x[32];
#pragma unroll
            for (int i = 0; i < PPT; i++) {
                x[i] = hm[i];
                x[PPT + i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hm[i], 0x130, 0xf, 0xf, false) + 16u;      // wave_shl:1 (lane 63 reads 0: unused)
            }
            switch (W) {
            case 3: sliding_min<3>(x, mh); break;   case 4: sliding_min<4>(x, mh); break;   case 5: sliding_min<5>(x, mh); break;
            case 6: sliding_min<6>(x, mh); break;   case 7: sliding_min<7>(x, mh); break;   case 8: sliding_min<8>(x, mh); break;
            case 9: sliding_min<9>(x, mh); break;   case 10: sliding_min<10>(x, mh); break; case 11: sliding_min<11>(x, mh); break;
            case 12: sliding_min<12>(x, mh); break; case 13: sliding_min<13>(x, mh); break; case 14: sliding_min<14>(x, mh); break;
            case 15: sliding_min<15>(x, mh); break; default: sliding_min<16>(x, mh); break;
            }
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
            n1 = min((uint32_t)__builtin_amdgcn_readlane((int)incl, 63), (uint32_t)QC);
            const uint32_t mybase = incl - mine;
            if (need) {
                const uint32_t dummy = (uint32_t)QC + (uint32_t)t;   // where entries beyond the capacity go
                const uint32_t stop1 = stop >> 1, tbase = (uint32_t)(t * PPT);
                uint32_t idx = mybase;
                for (uint32_t nd = need; nd; idx++) {
                    const uint32_t j = (uint32_t)__ffs(nd) - 1u;
                    nd &= nd - 1u;
                    const uint32_t len = (uint32_t)__ffs(stop1 >> j);                       // until the next run / gap
                    S.q1[min(idx, dummy)] = (len << 12) + (tbase + j + (nd ? 0u : ext12));
                }
                if (mybase + mine > (uint32_t)QC) {                          // rare: which of my runs did not fit
                    uint32_t r = mybase;
                    for (uint32_t nd = need; nd; nd &= nd - 1u, r++)
                        if (r >= (uint32_t)QC) ovf |= nd & (0u - nd);
                }
            }
        }
        __syncthreads();
        SS_T(2);
        SS_STOP(2)

        // minimizer offset (in the run's first k-mer) from the index byte of the run's first position:
        // meta = offset << 17 | len << 12 | position
        auto run_meta = [&](uint32_t e) {
            const uint32_t rpos = e & 0xFFFu;
            return e | ((((uint32_t)S.ib[rpos] & 31u) - (rpos & 15u)) << 17);
        };
        // a bucket reference whose tag bits match becomes a found run in q2 (phase 3 gives it 16 lanes, one per position)
        auto push_found = [&](uint32_t lo, uint32_t mask17, uint32_t meta, uint32_t ridx, bool queued) {
            const uint32_t bstart = lo & ss::START_MASK, multi = lo >> 31;
            const uint32_t amask = aligned_mask(mask17, (meta >> 17) & 31u, F);
            uint32_t i2 = QC;
            if (queued) i2 = atomicAdd(&S.cnt[1], 1u);
            if (i2 < QC) {
                S.q2[i2] = ((uint64_t)(lo & (ss::START_MASK | ss::PG_SOLID)) << 32) | (multi << 31) | (amask << 12) | ridx;      // (bit 62: solid)
            } else {
                // the queue is full (only with floods of tag collisions), or phase 3 is already
                // over (runs that overflowed q1): settle this run here, so that no k-mer is ever dropped
                const uint32_t rpos = meta & 0xFFFu, len = (meta >> 12) & 31u;
                for (uint32_t q = 0; q < len; q++) {
                    const uint32_t cpos = cand_slot(bstart, amask, q, F);
                    if (cpos) settle_item<false>(S, rpos + q, bstart, multi, cpos, mkeys[cpos], mkeys, counts, COMB_NONE, nullptr, khi_mask);
                }
            }
        };
        // One page against one run; tg = the page's first 16 bytes (tag8[8], hi8[8]).  Inline slots are whole database
        // k-mers: the k-mer of the run with minimizer offset o = 16 - e (if the run has one) equals the slot iff the
        // remaining tag bits and the flank agree -- the flank of EVERY k-mer of the run is a bit-select between the 16
        // bases behind the minimizer (fa) and the 16 in front of it (fb).  Returns true when the page is full (the
        // minimizer's slots may go on in the next page).
        auto scan_page = [&](const uint4 tg, uint32_t page, uint32_t meta, uint32_t h, uint32_t ridx, bool queued) -> bool {
            const uint32_t tt = (h & 0xFFu) * 0x01010101u;
            const uint32_t x0 = tg.x ^ tt, x1 = tg.y ^ tt;                      // zero byte = tag8 matches
            const uint32_t z0 = ~(((x0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x0) & 0x80808080u;
            const uint32_t z1 = ~(((x1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x1) & 0x80808080u;
            uint32_t hit = (z0 >> 7) | (z1 >> 3);                               // slot s at bit 8 (s & 3) + 4 (s >> 2)
            if (hit) {
                const uint32_t o0 = (meta >> 17) & 31u, len = (meta >> 12) & 31u;
                const char *pb = reinterpret_cast<const char *>(pages) + (uint64_t)page * 64u;
                do {
                    const uint32_t b = (uint32_t)__ffs(hit) - 1u, sl = (b >> 3) + (b & 4u);
                    hit &= hit - 1u;
                    const uint32_t hi8 = (((b & 4u) ? tg.w : tg.z) >> (b & 24u)) & 0xFFu;
                    const bool ref = hi8 & 0x80u;
                    const uint32_t e = hi8 & 31u, j = o0 + e - F;               // inline: k-mer j of the run has offset o0 - j (e = F - offset)
                    if (ref ? ((hi8 ^ (h >> 8)) & 0x3Fu) == 0u : j < len) {
                        const uint32_t lo = reinterpret_cast<const uint32_t *>(pb + 16)[sl];
                        const uint32_t mid = reinterpret_cast<const uint16_t *>(pb + 48)[sl];
                        if (ref) {
                            push_found(lo, mid | ((hi8 & 0x40u) << 10), meta, ridx, queued);
                        } else if ((mid >> 4) == ((h >> 8) & 0xFFFu)) {
                            const int32_t q = (int32_t)((meta & 0xFFFu) + o0);  // tile position of the minimizer
                            const uint32_t fa = win16_at(S, q + ss::MINI_M), fb = win16_at(S, q - (int32_t)F);      // the bases behind / the F bases in front
                            const uint32_t m = ((1u << e) << e) - 1u;           // low 2 e bits (e = 16: all)
                            if ((((fa & m) | (fb & ~m)) & fmask) == lo) atomicAdd(&counts[cbase + page * 8u + sl], 1u);
                        }
                    }
                } while (hit);
            }
            return (tg.w >> 24) != (uint32_t)ss::PG_EMPTY_HI;
        };
        // the rare continuation: the page was full, the minimizer's slots may go on in the next one(s)
        auto scan_more = [&](uint32_t page, uint32_t meta, uint32_t h, uint32_t ridx, bool queued) {
            uint4 tg;
            do {
                page++;                                   // the build guarantees a non-full page before the array ends
                tg = pages[(uint64_t)page * 4u];
            } while (scan_page(tg, page, meta, h, ridx, queued));
        };

        SS_T(7);
        for (int tb = 0; tb < (MULTI ? tabs.n : 1); tb++) {
        if (MULTI) {
            mkeys = tabs.mkeys[tb]; pages = tabs.pages[tb]; counts = tabs.counts[tb];
            n_pages = tabs.n_pages[tb]; cbase = tabs.cbase[tb]; tab_key = (uint32_t)tb << 30;
            if (tb) {                                   // the found-run queue of the table before has been worked off
                if (t == 1) S.cnt[1] = 0;
                __syncthreads();
            }
        }
        // ---- phase 2a (databases with a Bloom filter): one probe of an L2-resident bit array kills most of
        // the ~90 % of the runs whose minimizer is not in the database before they cost a random HBM
        // sector each.  Survivors are compacted into q1b (ballot + lane count: one wave per workgroup)
        uint32_t ns = n1;
#ifndef SS_RPL
#define SS_RPL 2
#endif
        constexpr int RPL = SS_RPL;
        if (BLOOM) {
            ns = 0;
            for (uint32_t r0 = 0; r0 < n1; r0 += RPL * MT) {
                SS_T(11);
                uint32_t meta[RPL], hs[RPL], bw[RPL];
                bool ok[RPL];
#pragma unroll
                for (int u = 0; u < RPL; u++) {
                    const uint32_t r = r0 + u * MT + t;
                    ok[u] = r < n1;
                    meta[u] = run_meta(S.q1[ok[u] ? r : 0u]);
                    hs[u] = ss::mix30(mmer_at(S, (meta[u] & 0xFFFu) + (meta[u] >> 17)));   // one mix per run: Bloom bit, page, tags
                }
#pragma unroll
                for (int u = 0; u < RPL; u++) bw[u] = bloom[hs[u] >> (bloom_shift + 5)];
#pragma unroll
                for (int u = 0; u < RPL; u++) ok[u] = ok[u] && ((bw[u] >> ((hs[u] >> bloom_shift) & 31u)) & 1u);
#pragma unroll
                for (int u = 0; u < RPL; u++) {
                    const uint64_t pass = __ballot(ok[u]);
                    if (ok[u]) SB.q1b[ns + __builtin_amdgcn_mbcnt_hi((uint32_t)(pass >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pass, 0u))] =
                        ((uint64_t)hs[u] << 32) | meta[u];
                    ns += (uint32_t)__popcll(pass);
                }
            }
            __syncthreads();
        }
        // ---- phase 2b: ONE sector per run (all runs, or the Bloom survivors), RPL2 page heads in flight per lane ----
#ifndef SS_RPL2
#define SS_RPL2 1
#endif
        constexpr int RPL2 = SS_RPL2;
        SS_CS(8, 1); SS_CS(9, n1); SS_CS(10, ns); SS_CS(12, (ns + RPL2 * MT - 1) / (RPL2 * MT));
        for (uint32_t i0 = 0; i0 < ns; i0 += RPL2 * MT) {
            SS_T(12);
            uint32_t meta[RPL2], hs[RPL2], pgi[RPL2];
            uint4 tg[RPL2];
            bool ok[RPL2];
#pragma unroll
            for (int u = 0; u < RPL2; u++) {
                const uint32_t i = i0 + u * MT + t;
                ok[u] = i < ns;
                if (BLOOM) {
                    const uint64_t sv = SB.q1b[ok[u] ? i : 0u];
                    meta[u] = (uint32_t)sv;
                    hs[u] = (uint32_t)(sv >> 32);
                } else {
                    meta[u] = run_meta(S.q1[ok[u] ? i : 0u]);
                    hs[u] = ss::mix30(mmer_at(S, (meta[u] & 0xFFFu) + (meta[u] >> 17)));
                }
                pgi[u] = ss::page_of(hs[u], n_pages);
                if (ok[u]) tg[u] = pages[(uint64_t)pgi[u] * 4u];
            }
            SS_T(13);
#pragma unroll
            for (int u = 0; u < RPL2; u++)
                if (ok[u] && scan_page(tg[u], pgi[u], meta[u], hs[u], i0 + u * MT + t, true)) scan_more(pgi[u], meta[u], hs[u], i0 + u * MT + t, true);
            SS_T(14);
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
            const uint32_t n2 = min(S.cnt[1], (uint32_t)QC);
            // a probe launch (choose_comb: the first tiles of a binned set against a table nobody has flagged) reports its found runs
            if (!COMB && (xcd_swizzle & 2u) && t == 0) atomicAdd(&ss_probe_runs[blockIdx.x & 63u], n2);
            bool comb_full = false;
            if (COMB) {
                // every found run claims the LDS entry of its bucket (the runs of a locus' reads share theirs); a byte
                // counter takes at most one from a run and an ENTRY takes 255 runs between two flushes (round 5: counted per
                // entry -- a bucket sees the ~7 reads of its locus per tile; the bound used to be 255 runs for the whole table,
                // a flush every 2.4 tiles), the run that finds its entry full counts in global memory like one without an entry.
                // The table is flushed when the workgroup leaves its CH consecutive tiles (the next ones are other loci).
                SS_T(10);
                comb_runs += n2;
                for (uint32_t r = (uint32_t)t; r < n2; r += MT) {
                    const uint32_t key = (((uint32_t)(S.q2[r] >> 32) & ss::START_MASK) | tab_key) + 1u;
                    uint32_t i = (key * 0x9E3779B1u) >> 26, e = COMB_NONE;
                    static_assert(COMB_NE == 64, "hash: top six bits");
                    for (int pr = 0; pr < 8; pr++) {
                        const uint32_t old = atomicCAS(&C.key[i], 0u, key);
                        if (old == 0u || old == key) { e = atomicAdd(&C.nrun[i], 1u) < 255u ? i : COMB_NONE; break; }
                        i = (i + 1u) & (COMB_NE - 1u);
                    }
                    C.ent[r] = (uint8_t)e;
                    comb_full = comb_full || e == COMB_NONE;
                }
#ifdef SS_COMB_STATS
                { const unsigned long long nf = __popcll(__ballot(comb_full)); SS_CS(0, 1); SS_CS(4, n2); SS_CS(5, nf); SS_CS(6, nf ? 1 : 0); }
#endif
                comb_full = __ballot(comb_full) != 0ull;      // runs without an entry count in global memory; the table is emptied after the tile
                __syncthreads();
                SS_T(9);
            }
            SS_CS(11, n2); SS_CS(13, (n2 * 16u + U3 * MT - 1) / (U3 * MT));
            uint32_t n_rest = 0, n_hit = 0;                // found runs whose bucket is not solid (S.rest from the front); solid runs with
                                                           // hits for the counters (S.rest from the back: together at most n2 entries)
            // U runs' worth of positions per lane and round: all their candidate loads in flight before any compare
            auto cand_round = [&](auto UC, uint32_t g0) {
                constexpr int U = decltype(UC)::value;
                SS_T(15);
                uint32_t pos[U], bst[U], mul[U], cps[U], ent[U];
                uint64_t cnd[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t g = g0 + u * MT + t, q = g & 15u;
                    const bool v = (g >> 4) < n_rest;
                    const uint32_t fi = v ? (COMB ? (uint32_t)C.rest[g >> 4] : (g >> 4)) : 0u;      // the found run (index into q2)
                    const uint64_t r = S.q2[fi];
                    const uint32_t run = BLOOM ? (uint32_t)SB.q1b[(uint32_t)r & 0xFFFu] : S.q1[(uint32_t)r & 0xFFFu];
                    pos[u] = (run & 0xFFFu) + q;
                    bst[u] = (uint32_t)(r >> 32) & ss::START_MASK;
                    mul[u] = (uint32_t)r >> 31;
                    cps[u] = (v && q < ((run >> 12) & 31u)) ? cand_slot(bst[u], ((uint32_t)r >> 12) & 0x1FFFFu, q, F) : 0u;
                    cnd[u] = mkeys[cps[u]];
                    ent[u] = COMB ? (uint32_t)C.ent[fi] : COMB_NONE;
                }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (cps[u]) settle_item<COMB>(S, pos[u], bst[u], mul[u], cps[u], cnd[u], mkeys, counts, ent[u], &C.acc[0][0], khi_mask);
                SS_MARK(16);
            };
            // (only in the combining scans: a tree scan finds ~6 runs per tile, and six lanes walking through this path cost more
            //  than the 16-lanes-per-run check of 96: contiguous node sets 3.37 -> 3.70 ms with it)
            // ---- 3a: runs whose bucket is SOLID (PG_SOLID: its k-mers are one stretch of bases), ONE LANE per run: the
            // stretch is rebuilt from the bucket's first and last candidate key (they overlap), compared with the read's bases
            // in three words, and a mismatching base strikes the k-mers that cover it.  The other runs are compacted into
            // S.rest for the 16-lanes-per-run check below.
            if (!COMB) n_rest = n2;                        // (S.rest is not used: run i is found run i)
            for (uint32_t r0 = 0; COMB && r0 < n2; r0 += MT) {
                const uint32_t r = r0 + (uint32_t)t;
                const bool v = r < n2;
                const uint64_t q = S.q2[v ? r : 0u];
                const bool solid = v && ((q >> 62) & 1ull);
                uint32_t hitmask = 0;                      // this lane's run: its matching k-mers, when they go straight to the counters
                const uint64_t om = __ballot(v && !solid);
                if (v && !solid) C.rest[n_rest + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u))] = (uint8_t)r;
                n_rest += (uint32_t)__popcll(om);
                if (solid) {
                    const uint32_t bstart = (uint32_t)(q >> 32) & ss::START_MASK, amask = ((uint32_t)q >> 12) & 0x1FFFFu;
                    const uint32_t run = BLOOM ? (uint32_t)SB.q1b[(uint32_t)q & 0xFFFu] : S.q1[(uint32_t)q & 0xFFFu];
                    const uint32_t rpos = run & 0xFFFu, len = (run >> 12) & 31u;
                    // position p of the run has offset F - p (16 - p at k = 31): the candidates are the offsets of amask at or above W - len
                    const uint32_t am = amask & ~((1u << (W - len)) - 1u);
                    if (am) {
                        const uint32_t o_hi = 31u - (uint32_t)__clz(am), o_lo = (uint32_t)__ffs(am) - 1u, d = o_hi - o_lo;
                        const uint32_t slot_a = bstart + 1u + (uint32_t)__popc(amask & ((1u << o_hi) - 1u));      // the first position's k-mer
                        const uint64_t k1 = mkeys[slot_a], k2 = mkeys[slot_a - d];                                // ... and the last one's
                        const int32_t P = (int32_t)(rpos + F - o_hi);
                        const uint64_t lo64 = k1 | (k2 << (2u * d));
                        const uint32_t hi32 = d ? (uint32_t)(k2 >> (64u - 2u * d)) : 0u;
                        uint32_t x0 = win16_at(S, P) ^ (uint32_t)lo64;
                        // the stretch is d + k bases: 16 in x0, the next min(16, d + k - 16) in x1, the d + k - 32 beyond (if any) in x2
                        const uint32_t nb1 = d + (uint32_t)K - 16u, nb2 = d + (uint32_t)K > 32u ? d + (uint32_t)K - 32u : 0u;
                        uint32_t x1 = (win16_at(S, P + 16) ^ (uint32_t)(lo64 >> 32)) & (nb1 >= 16u ? 0xFFFFFFFFu : (1u << (2u * nb1)) - 1u);
                        uint32_t x2 = nb2 ? (win16_at(S, P + 32) ^ hi32) & ((1u << (2u * nb2)) - 1u) : 0u;
                        uint32_t match = (2u << d) - 1u;                                                           // k-mers 0 .. d of the stretch
                        // a base that differs strikes the k-mers covering it: j in [m - 30, m]
                        x0 = (x0 | (x0 >> 1)) & 0x55555555u; x1 = (x1 | (x1 >> 1)) & 0x55555555u; x2 = (x2 | (x2 >> 1)) & 0x55555555u;
                        auto strike = [&](uint32_t xb, uint32_t base0) {
                            while (xb) {
                                const uint32_t m = base0 + (((uint32_t)__ffs(xb) - 1u) >> 1);
                                xb &= xb - 1u;
                                const uint32_t hi = min(m, d), lw = m > (uint32_t)K - 1u ? m - ((uint32_t)K - 1u) : 0u;
                                if (lw <= hi) match &= ~(((2u << hi) - 1u) & ~((1u << lw) - 1u));
                            }
                        };
                        strike(x0, 0u); strike(x1, 16u); strike(x2, 32u);
                        if (match) {
                            const uint32_t w_ = 1u;
                            const uint32_t off_a = slot_a - bstart;                                                // k-mer j sits at slot offset off_a - j
                            const uint32_t e_ = COMB ? (uint32_t)C.ent[r] : COMB_NONE;
                            if (COMB && e_ != COMB_NONE && off_a < 20u) {
                                const uint32_t bits = __brev(match) >> (31u - off_a);                              // bit o = the k-mer at slot offset o
#pragma unroll
                                for (uint32_t wd = 0; wd < 5; wd++) {
                                    const uint32_t nib = (bits >> (4u * wd)) & 15u;
                                    if (nib) atomicAdd(&C.acc[e_][wd], ((nib * 0x00204081u) & 0x01010101u) * w_);
                                }
                            } else {
                                // straight to the counters, but NOT one k-mer after the other from this lane: an atomic costs
                                // per (instruction, line), and a lane's hits share a line -- the run's result goes back into
                                // its q2 entry and 16 lanes count it below (3c)
                                hitmask = match;
                                S.q2[r] = ((uint64_t)slot_a << 32) | match;
                            }
                        }
                    }
                }
                const uint64_t hm = __ballot(hitmask != 0u);
                if (hitmask) C.rest[QC - 1u - (n_hit + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u)))] = (uint8_t)r;
                n_hit += (uint32_t)__popcll(hm);
            }
            if (COMB) __syncthreads();
            // ---- 3c: the solid runs' hits that go straight to the counters, 16 lanes per run (k-mer 16 of a full stretch: lane 0 again)
            for (uint32_t g0 = 0; g0 < n_hit * 16u; g0 += MT) {
                const uint32_t g = g0 + (uint32_t)t;
                if ((g >> 4) < n_hit) {
                    const uint64_t e = S.q2[C.rest[QC - 1u - (g >> 4)]];
                    const uint32_t j = g & 15u, slot_a = (uint32_t)(e >> 32), mt = (uint32_t)e;
                    if ((mt >> j) & 1u) atomicAdd(&counts[slot_a - j], 1u);
                    if (j == 0u && (mt >> 16)) atomicAdd(&counts[slot_a - 16u], 1u);
                }
            }
            // ---- 3b: the other found runs, 16 lanes each
            // (a tree scan finds ~5 runs per tile = 80 of a round's 256 positions; rounds of 64 for the rest were tried in
            //  round 4 and change nothing: the empty quarters of a round are branched over)
            const uint32_t total = n_rest * 16u;
            uint32_t g0 = 0;
            for (; g0 < total; g0 += U3 * MT) cand_round(std::integral_constant<int, U3>(), g0);
            if (COMB && comb_full) { __syncthreads(); SS_T(4); comb_flush(); SS_T(8); }
        }
        // ---- overflow: runs that did not fit q1 (pathological inputs only) are done in place ------
        if (ovf) {
            for (int j = 0; j < PPT; j++) {
                if (!((ovf >> j) & 1u)) continue;
                const uint32_t len12 = ((uint32_t)__ffs(stop >> (j + 1)) << 12) + ((need >> (j + 1)) ? 0u : ext12);
                const uint32_t meta = run_meta(len12 + (uint32_t)(t * PPT + j));
                const uint32_t h = ss::mix30(mmer_at(S, (meta & 0xFFFu) + (meta >> 17)));
                const uint32_t page = ss::page_of(h, n_pages);
                if (scan_page(pages[(uint64_t)page * 4u], page, meta, h, 0u, false)) scan_more(page, meta, h, 0u, false);
            }
        }
        if (MULTI && tb + 1 < tabs.n) __syncthreads();
        }   // tables
        SS_T(4);
        __syncthreads();   // queues and codes are rewritten by the next tile
        SS_T(5);
        if (COMB && CH > 1 && comb_runs && ((tile + 1 - tile0) % CH) == 0) { comb_flush(); SS_T(8); }      // the chunk ends: other loci next
    }
    if (COMB && comb_runs) comb_flush();
#ifdef SS_TIMING
    if (t == 0)
        for (int i = 0; i < 16; i++) atomicAdd(&ss_timing[i], (unsigned long long)t_acc[i]);
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
    // the device build first (ss_build_dev.hip: the same image, byte for byte, in a fraction of the time); SS_BUILD=host, or
    // anything it could not do (no memory, an empty table, a HIP error), leaves the work to the host build below
    {
        const char *b = getenv("SS_BUILD");
        if (!(b && !strcmp(b, "host")) && db->k == 31) {
            const int rc = build_mini_dev(db, keys, flags, n_rows, upper_keys);
            if (rc == SS_OK || rc == SS_EKEY) return rc;
            if (getenv("SS_BUILD_TRACE")) fprintf(stderr, "[build] device build declined (%d): host build\n", rc);
        }
    }
    const int k = db->k;
    static const bool trace = getenv("SS_BUILD_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[build] %-28s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    constexpr int PB = 8, NP = 1 << PB;
    unsigned nthreads = std::min<unsigned>(ss::host_cpus(), 32u);
    if (const char *e = getenv("SS_BUILD_THREADS")) nthreads = (unsigned)std::max(1, std::min(64, atoi(e)));   // tests: the image must not depend on it
    uint32_t inline_max = 2;                        // minimizers with at most this many database k-mers keep them in page slots
                                                    // (decided below, once the minimizers' sizes are known: choose_inline_max)
    double lambda = 2.0;                            // page items per page on average (eight slots: one page in a thousand full)
    if (const char *e = getenv("SS_PAGE_LAMBDA")) lambda = std::max(0.25, std::min(7.8, atof(e)));   // < 8: the pages must hold all items
    // 1. entries of valid rows with their minimizer
    std::vector<uint64_t> pos(n_rows + 1, 0);
    for (uint64_t i = 0; i < n_rows; i++) pos[i + 1] = pos[i] + ((flags[i] & SS_ROW_VALID) ? 1 : 0);
    const uint64_t nv = pos[n_rows];
    // (plain arrays: a std::vector would zero 2 x 0.8 GB on one thread first)
    std::unique_ptr<Ent[]> ents_buf(new (std::nothrow) Ent[std::max<uint64_t>(nv, 1)]), sorted_buf(new (std::nothrow) Ent[std::max<uint64_t>(nv, 1)]);
    if (!ents_buf || !sorted_buf) return SS_ENOMEM;
    Ent *ents = ents_buf.get(), *sorted = sorted_buf.get();
    parallel_for(nthreads, n_rows, [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t i = lo; i < hi; i++)
            if (flags[i] & SS_ROW_VALID) {
                uint32_t o;
                const uint32_t mx = mini_of_key(keys[i], k, &o);
                ents[pos[i]] = Ent{mx, (uint32_t)i, keys[i], o, mix30(mx) >> (30 - PB)};
            }
    });
    lap("1 minimizers");
    // 2. counting partition on the top 8 bits of h = mix30(minimizer) -- the page order --, then per-partition sort
    //    (threads over index ranges, each with its own counts and cursors: within a partition the entries keep their
    //     index order, whatever the thread count -- a serial pass took 0.15 s of scattered 32-byte writes)
    std::vector<uint64_t> pcount(NP + 1, 0);
    {
        const unsigned T = nv < (1u << 20) ? 1u : nthreads;
        const uint64_t per = (nv + T - 1) / T;
        std::vector<std::vector<uint64_t>> cnt(T, std::vector<uint64_t>(NP, 0));
        auto each_thread = [&](const std::function<void(unsigned)> &fn) {
            std::vector<std::thread> pool;
            for (unsigned w = 1; w < T; w++) pool.emplace_back(fn, w);
            fn(0);
            for (auto &th : pool) th.join();
        };
        each_thread([&](unsigned w) {
            for (uint64_t i = std::min(nv, per * w), e = std::min(nv, per * (w + 1)); i < e; i++) cnt[w][ents[i].part]++;
        });
        uint64_t run = 0;
        for (int p = 0; p < NP; p++) {
            pcount[p] = run;
            for (unsigned w = 0; w < T; w++) { const uint64_t c = cnt[w][p]; cnt[w][p] = run; run += c; }      // -> this thread's cursor
        }
        pcount[NP] = run;
        each_thread([&](unsigned w) {
            for (uint64_t i = std::min(nv, per * w), e = std::min(nv, per * (w + 1)); i < e; i++) sorted[cnt[w][ents[i].part]++] = ents[i];
        });
    }
    ents_buf.reset();
    auto for_partitions = [&](const std::function<void(int)> &fn) {
        std::atomic<int> next(0);
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < nthreads; w++)
            pool.emplace_back([&] { for (int p; (p = next.fetch_add(1)) < NP;) fn(p); });
        for (auto &th : pool) th.join();
    };
    for_partitions([&](int p) {
        std::sort(sorted + pcount[p], sorted + pcount[p + 1], [](const Ent &a, const Ent &b) {
            if (a.mini != b.mini) return a.mini < b.mini;
            if (a.off != b.off) return a.off < b.off;
            if (a.key != b.key) return a.key < b.key;
            return a.row < b.row;
        });
    });
    lap("2 partition + sort");
    // 3. distinct k-mers per minimizer.  Small sets become inline page items (one per k-mer), larger ones a bucket
    //    of d_mkeys (header + k-mers) plus ONE page item, the reference.  Row bookkeeping: dict overwrite, the last
    //    allowed row owns the count.  A minimizer lives in one partition, so the partitions are independent: count,
    //    prefix-sum, fill in parallel (same order as a serial walk: the image does not depend on the thread count).
    struct Item { uint32_t h, lo; uint16_t mid; uint8_t hi8; uint64_t e0, e1; };   // page item; [e0, e1) = its rows in `sorted` (inline items)
    std::vector<uint64_t> p_slots(NP + 1, 0), p_items(NP + 1, 0), p_minis(NP + 1, 0);
    auto walk = [&](int p, const std::function<void(uint64_t, uint64_t, uint32_t)> &bucket) {   // [i, e) = one minimizer, nd distinct k-mers
        for (uint64_t i = pcount[p]; i < pcount[p + 1];) {
            uint64_t e = i;
            uint32_t nd = 0;
            uint64_t last = ~0ull;
            while (e < pcount[p + 1] && sorted[e].mini == sorted[i].mini) {
                if (sorted[e].key != last) { nd++; last = sorted[e].key; }
                e++;
            }
            bucket(i, e, nd);
            i = e;
        }
    };
    {
        std::atomic<uint64_t> small_a(0), all_a(0);
        for_partitions([&](int p) {
            uint64_t sm = 0, al = 0;
            walk(p, [&](uint64_t, uint64_t, uint32_t nd) { al += nd; if (nd <= 2) sm += nd; });
            small_a += sm; all_a += al;
        });
        inline_max = choose_inline_max(small_a.load(), all_a.load());
    }
    for_partitions([&](int p) {
        uint64_t ns_ = 0, ni = 0, nm = 0;
        walk(p, [&](uint64_t, uint64_t, uint32_t nd) {
            nm++;
            if (nd <= inline_max) ni += nd;
            else { ni++; ns_ += 1 + nd; }
        });
        p_slots[p + 1] = ns_; p_items[p + 1] = ni; p_minis[p + 1] = nm;
    });
    for (int p = 0; p < NP; p++) { p_slots[p + 1] += p_slots[p]; p_items[p + 1] += p_items[p]; p_minis[p + 1] += p_minis[p]; }
    const uint64_t n_mslots = std::max<uint64_t>(1, p_slots[NP]), n_items = p_items[NP], n_minis = p_minis[NP];
    if (n_mslots >= (uint64_t)START_MASK) return SS_ERANGE;
    uint64_t n_pages = std::max<uint64_t>(PG_MIN_PAGES, (uint64_t)((double)n_items / lambda) + 1);
    std::vector<uint64_t> mkeys(n_mslots, 0);
    std::vector<Item> items(n_items);
    std::vector<uint32_t> slot_of_row(std::max<uint64_t>(1, n_rows), SS_NO_SLOT);
    std::vector<uint8_t> row_valid(std::max<uint64_t>(1, n_rows), 0);
    std::atomic<uint64_t> orphans_a(0), n_distinct_a(0);
    for_partitions([&](int p) {
        uint64_t ms = p_slots[p], it = p_items[p], orph = 0, ndist = 0;
        walk(p, [&](uint64_t i, uint64_t e, uint32_t nd) {
            const uint32_t h = mix30(sorted[i].mini);
            const bool inl = nd <= inline_max;
            const uint32_t hslot = (uint32_t)ms;
            if (!inl) ms++;
            uint32_t mask = 0, multi = 0;
            for (uint64_t a2 = i; a2 < e;) {
                uint64_t b2 = a2;
                int64_t owner = -1;
                while (b2 < e && sorted[b2].key == sorted[a2].key) {
                    const uint32_t r = sorted[b2].row;
                    if (upper_keys == 1 || !(flags[r] & SS_ROW_LOWER)) owner = r;   // rows ascend within equal k-mers
                    b2++;
                }
                const uint32_t o = sorted[a2].off;
                if (inl) {
                    items[it++] = Item{h, flank_of_key_k(sorted[a2].key, o, k), (uint16_t)(((h >> 8) & 0xFFFu) << 4), (uint8_t)((uint32_t)(k - MINI_M) - o), a2, b2};      // (e = k - 15 - o: 16 - o at k = 31)
                } else {
                    if ((mask >> o) & 1u) multi = 1u;
                    mask |= 1u << o;
                    const uint32_t slot = (uint32_t)ms;
                    mkeys[ms++] = sorted[a2].key;
                    for (uint64_t q = a2; q < b2; q++) slot_of_row[sorted[q].row] = slot;
                }
                if (owner >= 0) row_valid[owner] = 1;
                else orph++;
                ndist++;
                a2 = b2;
            }
            if (!inl) {
                mkeys[hslot] = ((uint64_t)nd << 32) | (multi ? HDR_MULTI : 0u) | mask;
                items[it++] = Item{h, (multi << 31) | hslot, (uint16_t)(mask & 0xFFFFu), (uint8_t)(0x80u | ((mask >> 16) << 6) | ((h >> 8) & 0x3Fu)), 0, 0};
            }
        });
        // page order inside the partition (the partitions themselves are h ranges); stable: a minimizer's items stay together
        std::stable_sort(items.begin() + p_items[p], items.begin() + p_items[p + 1], [](const Item &a, const Item &b) { return a.h < b.h; });
        orphans_a += orph;
        n_distinct_a += ndist;
    });
    const uint64_t orphans = orphans_a.load();
    if (orphans && upper_keys == 0) return SS_EKEY;
    db->n_distinct = n_distinct_a.load();
    lap("3 buckets + items");
    // 4. place the items: home page = page_of(h), or the first page behind it that is not full (a lookup reads on while
    //    the page it sees is full; no wrap-around: a few spare pages follow the last home page).  Serial in h order:
    //    ~20 ns per item.  Exactness of the inline slots: two minimizers whose h agree in the 20 tag bits have home pages
    //    >= D = n_pages / 1024 apart, so neither's lookup can reach the other's slots as long as every run of
    //    consecutive full pages is shorter than D -- checked here; the table grows until it holds (at two items per
    //    page a run of four full pages has probability 1e-12).
    std::vector<uint8_t> pages;
    uint64_t n_alloc = 0;
    for (;; n_pages += n_pages / 4) {
        if (n_mslots + (n_pages + n_pages / 1024) * PG_SLOTS >= 0xFFFFFFF0ull) return SS_ERANGE;
        const uint64_t D = n_pages / 1024;
        n_alloc = n_pages + D;
        pages.resize(n_alloc * 64);
        std::vector<uint8_t> fill(n_alloc, 0);
        parallel_for(nthreads, n_alloc, [&](uint64_t lo, uint64_t hi, unsigned) {
            for (uint64_t pg = lo; pg < hi; pg++) {
                memset(&pages[pg * 64], PG_EMPTY_TAG, 8);
                memset(&pages[pg * 64 + 8], PG_EMPTY_HI, 8);
                memset(&pages[pg * 64 + 16], 0, 48);
            }
        });
        // partition p (the items whose h has top byte p) owns the pages [lo(p), lo(p + 1)); its thread places its items
        // there; items that run past the end of the range (or whose home page straddles into the next range) are
        // placed afterwards, serially, in h order -- the same image for any thread count
        auto place = [&](const Item &it, uint64_t pg, uint64_t end) -> bool {
            while (pg < end && fill[pg] == PG_SLOTS) pg++;
            if (pg >= end) return false;
            const uint32_t sl = fill[pg]++;
            uint8_t *pp = &pages[pg * 64];
            pp[sl] = (uint8_t)(it.h & 0xFFu);
            pp[8 + sl] = it.hi8;
            memcpy(pp + 16 + 4 * sl, &it.lo, 4);
            memcpy(pp + 48 + 2 * sl, &it.mid, 2);
            for (uint64_t q = it.e0; q < it.e1; q++) slot_of_row[sorted[q].row] = (uint32_t)(n_mslots + pg * PG_SLOTS + sl);
            return true;
        };
        auto lo_of = [&](int pt) -> uint64_t { return pt >= NP ? n_pages : page_of((uint32_t)pt << (30 - PB), (uint32_t)n_pages); };
        std::vector<std::vector<uint64_t>> spill(NP);
        for_partitions([&](int pt) {
            const uint64_t end = lo_of(pt + 1);
            for (uint64_t i = p_items[pt]; i < p_items[pt + 1]; i++)
                if (!place(items[i], page_of(items[i].h, (uint32_t)n_pages), end)) spill[pt].push_back(i);
        });
        bool ok = true;
        for (int pt = 0; pt < NP && ok; pt++)
            for (uint64_t i : spill[pt])
                if (!place(items[i], std::max<uint64_t>(page_of(items[i].h, (uint32_t)n_pages), lo_of(pt + 1)), n_alloc)) { ok = false; break; }
        uint64_t run = 0, longest = 0;
        for (uint64_t pg = 0; pg < n_alloc && ok; pg++) {
            run = fill[pg] == PG_SLOTS ? run + 1 : 0;
            longest = std::max(longest, run);
        }
        if (ok && longest < D && fill[n_alloc - 1] < PG_SLOTS) break;
    }
    sorted_buf.reset();
    db->n_mslots = n_mslots;
    db->n_inline = (db->n_distinct + n_items - p_slots[NP]) / 2;   // items = inline k-mers + references; bucket slots = references + their k-mers
    db->n_slots = n_mslots + n_alloc * PG_SLOTS;
    db->n_dir = (uint32_t)n_pages;
    db->n_dir_alloc = (uint32_t)n_alloc;
    db->dirbits = 0;
    db->n_buckets = n_minis;
    db->capacity = db->n_slots;
    lap("4 pages");
    // 5. upload
    const uint64_t nr = std::max<uint64_t>(1, n_rows);
    SS_HIP(hipMalloc((void **)&db->d_mkeys, n_mslots * sizeof(uint64_t)));
    SS_HIP(hipMalloc((void **)&db->d_dir, pages.size()));
    SS_HIP(hipMalloc((void **)&db->d_counts, db->n_slots * sizeof(uint32_t)));
    SS_HIP(hipMalloc((void **)&db->d_slot_of_row, nr * sizeof(uint32_t)));
    SS_HIP(hipMalloc((void **)&db->d_row_valid, nr));
    db->device_bytes = n_mslots * 8 + db->n_slots * 4 + pages.size() + nr * 5;
    SS_HIP(hipMemcpy(db->d_mkeys, mkeys.data(), n_mslots * sizeof(uint64_t), hipMemcpyHostToDevice));
    SS_HIP(hipMemcpy(db->d_dir, pages.data(), pages.size(), hipMemcpyHostToDevice));
    SS_HIP(hipMemset(db->d_counts, 0, db->n_slots * sizeof(uint32_t)));
    {
        // Bloom filter over the minimizers, at most 2^25 bits = 4 MB (the L2 of one XCD; measured on a 25 M-row table of
        // dense node sets -- 2.8 M minimizers -- 2^23: 4.71 ms, 2^25: 4.60 ms, 2^27: 5.29 ms, none: 6.0 ms), and only
        // with >= 4 bits per minimizer: on a table of SAMPLED node sets (17 M minimizers) the filter passes 40 % of the
        // absent minimizers, half of the runs find theirs anyway, and the scan is 7 % faster without it (7.73 -> 7.18 ms).
        // SS_BLOOM_BITS=0 disables, = n forces 2^n bits.
        int bits = 10;
        while (bits < 25 && (1ull << bits) < 8 * n_minis) bits++;
        if ((1ull << bits) < 4 * n_minis) bits = 0;
        const char *bb = getenv("SS_BLOOM_BITS");
        if (bb) bits = atoi(bb);
        if (bits >= 10 && bits <= 30) {
            std::vector<uint32_t> bloom((size_t)1 << (bits - 5), 0);
            uint32_t last = ~0u;
            for (const auto &it : items) {
                if (it.h == last) continue;
                last = it.h;
                const uint32_t hb = it.h >> (30 - bits);
                bloom[hb >> 5] |= 1u << (hb & 31u);
            }
            SS_HIP(hipMalloc((void **)&db->d_bloom, bloom.size() * 4));
            SS_HIP(hipMemcpy(db->d_bloom, bloom.data(), bloom.size() * 4, hipMemcpyHostToDevice));
            db->bloom_bits = (uint32_t)bits;
            db->device_bytes += bloom.size() * 4;
        }
    }
    SS_HIP(hipMemcpy(db->d_slot_of_row, slot_of_row.data(), nr * sizeof(uint32_t), hipMemcpyHostToDevice));
    SS_HIP(hipMemcpy(db->d_row_valid, row_valid.data(), nr, hipMemcpyHostToDevice));
    lap("5 bloom + upload");
    return mark_solid(db);
}

// PG_SOLID for every bucket whose k-mers are one stretch of bases (ss_scan_dev.h): one thread per page slot, after either
// build has put pages and buckets on the device -- the same flags whichever build made the image.
__global__ __launch_bounds__(256) void mark_solid_kernel(uint8_t *__restrict__ pages, uint64_t n_page_slots, const uint64_t *__restrict__ mkeys, int k_of_db)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_page_slots) return;
    uint8_t *pp = pages + (i >> 3) * 64;
    const uint32_t sl = (uint32_t)(i & 7u);
    if (!(pp[8 + sl] & 0x80u)) return;                                   // an inline k-mer, or empty
    uint32_t *lo32 = reinterpret_cast<uint32_t *>(pp + 16) + sl;
    const uint32_t lo = *lo32;
    if (lo >> 31) return;                                                // several k-mers per offset
    const uint32_t b = lo & ss::START_MASK;
    const uint64_t hdr = mkeys[b];
    const uint32_t mask = (uint32_t)hdr & 0x1FFFFu, cnt = (uint32_t)(hdr >> 32);
    if ((hdr & ss::HDR_MULTI) || !mask || cnt != (uint32_t)__popc(mask)) return;
    const uint32_t m = mask >> (__ffs(mask) - 1);
    if (m & (m + 1u)) return;                                            // a gap in the offsets
    // slots ascend with the offset; the k-mer of offset o + 1 begins one base before the k-mer of offset o
    for (uint32_t k = 1; k < cnt; k++)
        if ((mkeys[b + k] & ((1ull << (2 * k_of_db - 2)) - 1ull)) != (mkeys[b + k + 1] >> 2)) return;
    *lo32 = lo | ss::PG_SOLID;
}

int mark_solid(ss_db *db)
{
    const uint64_t n = (uint64_t)db->n_dir_alloc * ss::PG_SLOTS;
    if (!n) return SS_OK;
    hipLaunchKernelGGL(mark_solid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (uint8_t *)db->d_dir, n, db->d_mkeys, db->k);
    SS_HIP(hipGetLastError());
    SS_HIP(hipDeviceSynchronize());
    return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// The page index at ANY k from 17 to 30 (round 6; `-k`, StrainScan.py:136,266-271, reaches the layer-2 scans:
// Vote_Strain_L2_Lasso_new_sp.py:359-371).  Until now every k but 31 went through the flat table (ss_scan.hip): one random
// 64-byte sector per k-mer, 0.08 of the HBM peak by SURVEY 8(d)'s bytes.  The index itself never depended on k = 31 -- a
// k-mer has k - 14 m-mers, its flank k - 15 bases (<= 32 bits), offsets 0..k-15 (<= 17 mask bits) -- only the scan kernel above
// does, in every phase (17-m-mer windows split over two lanes, 16-lane candidate checks, the combining table).  This kernel is
// the plain statement of the same lookups with k as a run-time value: ONE LANE PER START POSITION.  The m-mer keys of a
// 1024-position tile go to LDS once; a position takes the minimum of its k - 14 keys (leftmost on ties, as the build),
// mixes the minimizer, reads the head of its page and settles its own k-mer against the slots that match.  Consecutive
// positions of a read share their minimizer for ~(k - 14) / 2 positions and sit in neighbouring lanes: their page loads are
// the same address in one wave instruction -- one sector from L2 / HBM per run, as above, with no run queues at all.
// ~70 lane instructions per position at k = 25 against the tuned kernel's 45 at k = 31.
// ---------------------------------------------------------------------------------------------
constexpr int KT = 64, KPOS = 1024, KW = KPOS / 16 + 3;      // one wave per workgroup; start positions per tile; 16-base code words per tile (tile + 48 bases)
constexpr int KQ = 320;                                       // candidates (positions whose page has a slot with their tag) queued at a time:
                                                              // a round of 256 positions adds at most 256 to fewer than 64
struct KShared {
    uint32_t code[KW + 1];
    alignas(8) uint16_t inv[KW + 5];
    alignas(16) uint32_t key[KPOS + 32];
    uint2 q[KQ];                                              // position | minimizer offset << 10, h
};

// How the time of a first version went (4 M reads, k = 25, profiles/r06_ab_log.md): one lane per position, four positions of a
// thread one after the other: 4.0 ms -- 1.9 of it the minimizers (a loop of k - 14 dependent LDS reads per position at five waves
// per SIMD), 0.2 the page sectors, 2.4 the slots: 4 % of the positions hit, so nearly every wave walked the whole hit path, four
// times per tile.  Hence: a lane owns FOUR ADJACENT positions and reads their k - 11 keys once, as five 16-byte LDS loads (the
// four windows share all but three keys on either side); the four page heads are in flight together; positions whose page shows
// their tag (or is full) are compacted into an LDS queue with ballots and settled ONCE per tile, one candidate per lane.
template <bool ALIGNED, bool BLOOM>
__global__ __launch_bounds__(KT) void scan_minik_kernel(const uint8_t *__restrict__ bases, uint64_t n, uint64_t n_tiles, int k,
                                                        const uint64_t *__restrict__ mkeys, const uint4 *__restrict__ pages, uint32_t n_pages,
                                                        uint32_t *__restrict__ counts, uint32_t cbase, const uint32_t *__restrict__ bloom,
                                                        uint32_t bloom_shift)
{
    __shared__ KShared S;
    const int t = threadIdx.x;
    const uint32_t W = (uint32_t)(k - ss::MINI_M + 1), F = W - 1u;      // m-mers per k-mer (3..17), flank bases
    const uint64_t kmask = (1ull << (2 * k)) - 1ull, vmask = (1ull << k) - 1ull;
    // one k-mer against the slots of its minimizer's page(s): position p of the tile, minimizer offset o, h = mix30(minimizer)
    auto settle = [&](uint32_t p, uint32_t o, uint32_t h, uint32_t page) {
        const uint32_t w0 = p >> 4, sh = 2 * (p & 15);
        const uint32_t lo = __builtin_amdgcn_alignbit(S.code[w0 + 1], S.code[w0], sh), hi = __builtin_amdgcn_alignbit(S.code[w0 + 2], S.code[w0 + 1], sh);
        const uint64_t key = (((uint64_t)hi << 32) | lo) & kmask;      // bases p .. p + k - 1, base i at bits 2 i
        const uint32_t tt = (h & 0xFFu) * 0x01010101u;
        bool full;
        do {
            const uint4 tg = pages[(uint64_t)page * 4u];                        // (in L1 / L2: the lookup has just read it)
            const uint32_t x0 = tg.x ^ tt, x1 = tg.y ^ tt;                      // zero byte = tag8 matches
            const uint32_t z0 = ~(((x0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x0) & 0x80808080u;
            const uint32_t z1 = ~(((x1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x1) & 0x80808080u;
            uint32_t hit = (z0 >> 7) | (z1 >> 3);                               // slot s at bit 8 (s & 3) + 4 (s >> 2)
            const char *pb = reinterpret_cast<const char *>(pages) + (uint64_t)page * 64u;
            while (hit) {
                const uint32_t b = (uint32_t)__ffs(hit) - 1u, sl = (b >> 3) + (b & 4u);
                hit &= hit - 1u;
                const uint32_t hi8 = (((b & 4u) ? tg.w : tg.z) >> (b & 24u)) & 0xFFu;
                if (hi8 & 0x80u) {                                              // bucket reference
                    if ((hi8 ^ (h >> 8)) & 0x3Fu) continue;
                    const uint32_t l32 = reinterpret_cast<const uint32_t *>(pb + 16)[sl];
                    const uint32_t mask = reinterpret_cast<const uint16_t *>(pb + 48)[sl] | ((hi8 & 0x40u) << 10);
                    const uint32_t bstart = l32 & ss::START_MASK;
                    bool found = false;
                    if ((mask >> o) & 1u) {
                        const uint32_t cpos = bstart + 1u + (uint32_t)__popc(mask & ((1u << o) - 1u));
                        if (mkeys[cpos] == key) { atomicAdd(&counts[cpos], 1u); found = true; }
                    }
                    if (!found && (l32 >> 31)) {                                // several k-mers per offset: look through the bucket
                        const uint32_t cnt = (uint32_t)(mkeys[bstart] >> 32);
                        for (uint32_t c = 0; c < cnt; c++)
                            if (mkeys[bstart + 1u + c] == key) { atomicAdd(&counts[bstart + 1u + c], 1u); break; }
                    }
                } else if ((hi8 & 31u) == F - o) {                              // an inline k-mer with this minimizer offset
                    const uint32_t mid = reinterpret_cast<const uint16_t *>(pb + 48)[sl];
                    if ((mid >> 4) == ((h >> 8) & 0xFFFu) && reinterpret_cast<const uint32_t *>(pb + 16)[sl] == ss::flank_of_key_k(key, o, k))
                        atomicAdd(&counts[cbase + page * 8u + sl], 1u);
                }
            }
            full = (tg.w >> 24) != (uint32_t)ss::PG_EMPTY_HI;
            page++;                                                             // (the build guarantees a non-full page before the array ends)
        } while (full);
    };
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t b0 = tile * (uint64_t)KPOS;
        __syncthreads();                                   // (the tile before is done with S)
        // ---- bases -> codes + invalid flags: 16 bases per lane, the 48 behind the tile by lanes 0..2
        {
            uint32_t w[4], code, inv;
            load16<ALIGNED>(bases, b0 + (uint64_t)t * 16, n, w);
            encode16(w, code, inv);
            S.code[t] = code;
            S.inv[t] = (uint16_t)inv;
            if (t < 3) {
                load16<ALIGNED>(bases, b0 + (uint64_t)(KT + t) * 16, n, w);
                encode16(w, code, inv);
                S.code[KT + t] = code;
                S.inv[KT + t] = (uint16_t)inv;
            } else if (t < 8) {
                if (t == 3) S.code[KW] = 0u;
                S.inv[KT + t] = 0xFFFFu;
            }
        }
        __syncthreads();
        // ---- ordering keys of the m-mers that start in the lane's 16 bases (and the 16 behind the tile: lanes 0..15, one each)
        {
            const uint32_t c0 = S.code[t], c1 = S.code[t + 1];
            uint32_t kk[16];
            kk[0] = ss::mmkey(c0) & ss::KEY_MASK;
#pragma unroll
            for (int i = 1; i < 16; i++) kk[i] = ss::mmkey(__builtin_amdgcn_alignbit(c1, c0, 2 * i)) & ss::KEY_MASK;      // (mmkey looks at the low 24 bits only)
#pragma unroll
            for (int i = 0; i < 4; i++) reinterpret_cast<uint4 *>(&S.key[16 * t])[i] = make_uint4(kk[4 * i], kk[4 * i + 1], kk[4 * i + 2], kk[4 * i + 3]);
            if (t < 16) {
                const uint32_t q = (uint32_t)KPOS + (uint32_t)t;
                S.key[q] = ss::mmkey(__builtin_amdgcn_alignbit(S.code[(q >> 4) + 1], S.code[q >> 4], 2 * (q & 15))) & ss::KEY_MASK;
            } else if (t < 32) {
                S.key[KPOS + t] = 0xFFFFFFFFu;
            }
        }
        __syncthreads();
        uint32_t nq = 0;                                   // candidates queued (the same in every lane)
#pragma unroll 1
        for (uint32_t g = 0; g <= (uint32_t)(KPOS / (4 * KT)); g++) {
            if (g < (uint32_t)(KPOS / (4 * KT))) {
            const uint32_t p0 = 4u * ((uint32_t)t + (uint32_t)KT * g);
            // the 20 keys from p0 on, each tagged with its distance from p0 in its five free low bits: ONE v_min decides key
            // and leftmost position.  Window j = keys j .. j + W - 1 = {j..2} + {3..W-1} (common to the four) + {W..W+j-1}
            // (W is the same for the whole launch: the loops below leave through SCALAR branches -- no lane predicate, one v_min per key)
            uint32_t K[20];
#pragma unroll
            for (int i = 0; i < 5; i++) {
                if (i && (uint32_t)(4 * i) >= W) break;
                const uint4 v = reinterpret_cast<const uint4 *>(&S.key[p0])[i];
                K[4 * i] = v.x | (uint32_t)(4 * i); K[4 * i + 1] = v.y | (uint32_t)(4 * i + 1);
                K[4 * i + 2] = v.z | (uint32_t)(4 * i + 2); K[4 * i + 3] = v.w | (uint32_t)(4 * i + 3);
            }
            const uint32_t T0 = S.key[p0 + W] | W, T1 = S.key[p0 + W + 1u] | (W + 1u), T2 = S.key[p0 + W + 2u] | (W + 2u);
            uint32_t common = 0xFFFFFFFFu;
#pragma unroll
            for (int i = 3; i < 17; i++) {
                if ((uint32_t)i >= W) break;
                common = min(common, K[i]);
            }
            uint32_t m_[4];
            m_[0] = min(min(K[0], K[1]), min(K[2], common));
            m_[1] = min(min(K[1], K[2]), min(common, T0));
            m_[2] = min(min(K[2], common), min(T0, T1));
            m_[3] = min(min(common, T0), min(T1, T2));
            uint64_t iv;
            __builtin_memcpy(&iv, &S.inv[p0 >> 4], 8);
            iv >>= (p0 & 15u);
            uint32_t h_[4], page_[4];
            uint4 tg_[4];
            bool go_[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                go_[j] = ((iv >> j) & vmask) == 0;          // live: the k bases from p0 + j on are all ACGT (bytes beyond the block read as '\n')
                const uint32_t q = p0 + (m_[j] & 31u);       // tile position of the minimizer
                const uint32_t x = __builtin_amdgcn_alignbit(S.code[(q >> 4) + 1], S.code[q >> 4], 2 * (q & 15)) & ss::M30;
                h_[j] = ss::mix30(x);
                page_[j] = ss::page_of(h_[j], n_pages);
            }
#if defined(SS_KSTOP) && SS_KSTOP == 1      // (debug builds: the time of the phases up to here; results are then of course wrong)
            { uint32_t acc = 0; for (int r = 0; r < 4; r++) acc ^= h_[r] ^ m_[r] ^ (uint32_t)go_[r]; if (acc == 0x12345678u) atomicAdd(&counts[0], 1u); continue; }
#endif
            if (BLOOM) {
                uint32_t bw[4];
#pragma unroll
                for (int j = 0; j < 4; j++) bw[j] = go_[j] ? bloom[h_[j] >> (bloom_shift + 5)] : 0u;
#pragma unroll
                for (int j = 0; j < 4; j++) go_[j] = go_[j] && ((bw[j] >> ((h_[j] >> bloom_shift) & 31u)) & 1u);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                tg_[j] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0x7F7F7F7Fu, 0x7F7F7F7Fu);      // (an empty page: nothing matches, not full)
                if (go_[j]) tg_[j] = pages[(uint64_t)page_[j] * 4u];
            }
#if defined(SS_KSTOP) && SS_KSTOP == 2
            { uint32_t acc = 0; for (int r = 0; r < 4; r++) acc ^= tg_[r].x ^ tg_[r].w ^ m_[r]; if (acc == 0x12345678u) atomicAdd(&counts[0], 1u); continue; }
#endif
            bool cand_[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // a candidate: some slot of the page carries the minimizer's tag8 AND is either an inline k-mer with THIS k-mer's
                // minimizer offset (hi8 == e = F - o) or a bucket reference with the minimizer's filter bits (hi8 = 0x80 | mask bit
                // 16 << 6 | h[13:8]) -- all eight slots at once, on the 16 bytes the lookup has read (a read k-mer shares its
                // minimizer with a database k-mer six times as often as it IS one); or the page is full (its slots may go on)
                auto zb = [](uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; };      // 0x80 where a byte is zero
                const uint32_t tt = (h_[j] & 0xFFu) * 0x01010101u;
                const uint32_t e4 = (F + (uint32_t)j - (m_[j] & 31u)) * 0x01010101u, r4 = (0x80u | ((h_[j] >> 8) & 0x3Fu)) * 0x01010101u;
                const uint32_t c0 = zb(tg_[j].x ^ tt) & (zb(tg_[j].z ^ e4) | zb((tg_[j].z ^ r4) & 0xBFBFBFBFu));
                const uint32_t c1 = zb(tg_[j].y ^ tt) & (zb(tg_[j].w ^ e4) | zb((tg_[j].w ^ r4) & 0xBFBFBFBFu));
                cand_[j] = go_[j] && ((c0 | c1) != 0u || (tg_[j].w >> 24) != (uint32_t)ss::PG_EMPTY_HI);
            }
            // The queue is kept in POSITION order (a lane's candidates side by side, the lanes in order: a wave prefix sum of the
            // lanes' counts), so that the lanes of a drain hold neighbouring positions: the k-mers of a run hit neighbouring
            // counters of ONE bucket, and what an atomic costs on this chip is (instruction, 64-byte line) pairs (27 G/s,
            // profiles/r04_atomics_micro_*.txt).  Queued position class by position class (0, 4, 8, ... then 1, 5, 9, ...) a cluster
            // table's 111 hits per read were ~1.5 hits per pair: 18 ms per 8 M reads.
            {
                const uint32_t mine = (uint32_t)cand_[0] + (uint32_t)cand_[1] + (uint32_t)cand_[2] + (uint32_t)cand_[3];
                const uint32_t incl = wave_inclusive_sum(mine);
                uint32_t idx = nq + incl - mine;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (cand_[j]) S.q[idx++] = make_uint2((p0 + (uint32_t)j) | (((m_[j] & 31u) - (uint32_t)j) << 10), h_[j]);
                nq += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
            }
#if defined(SS_KSTOP) && SS_KSTOP == 3
            if (nq == 0x12345678u) atomicAdd(&counts[0], S.q[t].x);
            nq = 0;
            continue;
#endif
            // ---- the candidates, one per lane, whenever a wave's worth has come together (and at the end of the tile)
            if (nq >= (uint32_t)KT || g == (uint32_t)(KPOS / (4 * KT))) {
                __syncthreads();
                for (uint32_t e = (uint32_t)t; e < nq; e += KT) {
                    const uint2 c = S.q[e];
                    settle(c.x & 1023u, c.x >> 10, c.y, ss::page_of(c.y, n_pages));
                }
                __syncthreads();
                nq = 0;
            }
        }
    }
}

std::atomic<long long> g_hook_generic_k{0};      // ss_test_hook(4, ...)
static int launch_scan_minik(ss_db *db, const uint8_t *b, uint64_t n, hipStream_t stream)
{
    const uint64_t n_tiles = (n + KPOS - 1) / KPOS;
    // (one-wave workgroups, grid stride; 8 K / 32 K / 131 K / 300 K / 600 K of them: 2.56 / 2.37 / 2.31 / 2.30 / 2.29 ms per 4 M reads at k = 25)
    const unsigned blocks = (unsigned)std::min<uint64_t>(n_tiles, (uint64_t)256 * 32 * 16);
    const bool aligned = (((uintptr_t)b) & 15) == 0;
    const uint4 *pages = reinterpret_cast<const uint4 *>(db->d_dir);
    const uint32_t cbase = (uint32_t)db->n_mslots, bshift = 30u - db->bloom_bits;
    const bool bl = db->d_bloom && !db->expect_hits;
#define SS_LAUNCH_K(A, B) hipLaunchKernelGGL((scan_minik_kernel<A, B>), dim3(blocks), dim3(KT), 0, stream, b, n, n_tiles, db->k, db->d_mkeys, pages, db->n_dir, \
                                             db->d_counts, cbase, db->d_bloom, bshift)
    if (bl) { if (aligned) SS_LAUNCH_K(true, true); else SS_LAUNCH_K(false, true); }
    else    { if (aligned) SS_LAUNCH_K(true, false); else SS_LAUNCH_K(false, false); }
#undef SS_LAUNCH_K
    SS_HIP(hipGetLastError());
    return SS_OK;
}

template <int LB>
static void launch_lb(bool aligned, bool comb, unsigned blocks, hipStream_t stream, const uint8_t *bases, uint64_t n,
                      uint64_t n_tiles, ss_db *db, bool probe = false)
{
    const uint4 *pages = reinterpret_cast<const uint4 *>(db->d_dir);
    const uint32_t cbase = (uint32_t)db->n_mslots, bshift = 30u - db->bloom_bits;
    constexpr uint32_t swz0 = 1u;
    const uint32_t swz = swz0 | (probe ? 2u : 0u);
    const ScanTabs none = {};
    // (the combining variant needs 79 VGPRs: there is no 8-waves-per-SIMD build of it -- it carried 32 bytes of scratch)
    // k = 31: the instantiation with k a constant; any other k (17..30): k at run time (its queues of 256 runs: five waves per SIMD)
#define SS_LAUNCH(A, B, C_) do {                                                                                                                         \
        if (db->k == 31) hipLaunchKernelGGL((scan_mini_kernel<A, B, C_, (C_ && LB > 6) ? 6 : LB, false, 31>), dim3(blocks), dim3(MT), 0, stream, bases, n,    \
                                            n_tiles, db->d_mkeys, pages, db->n_dir, db->d_counts, cbase, db->d_bloom, bshift, swz, none, 31);               \
        else hipLaunchKernelGGL((scan_mini_kernel<A, B, C_, (LB > 5 ? 5 : LB), false, 0>), dim3(blocks), dim3(MT), 0, stream, bases, n, n_tiles,               \
                                db->d_mkeys, pages, db->n_dir, db->d_counts, cbase, db->d_bloom, bshift, swz, none, db->k);                                 \
    } while (0)
    // a table that expects hits (ss_db_expect_hits) skips its Bloom filter: nearly every minimizer of the reads is in it
    if (comb)             { if (aligned) SS_LAUNCH(true, false, true); else SS_LAUNCH(false, false, true); }
    else if (db->d_bloom && !db->expect_hits) { if (aligned) SS_LAUNCH(true, true, false); else SS_LAUNCH(false, true, false); }
    else                  { if (aligned) SS_LAUNCH(true, false, false); else SS_LAUNCH(false, false, false); }
#undef SS_LAUNCH
}

// one pass of a flat block against up to MULTI_MAX tables of the minimizer layout (all k = 31)
int launch_scan_mini_multi(ss_db *const *dbs, int n_dbs, const void *bases_dev, uint64_t n, hipStream_t stream, bool binned)
{
    if (n_dbs < 1 || n_dbs > MULTI_MAX) return SS_EINVAL;
    ScanTabs tabs = {};
    bool expect = true;
    const int k_all = dbs[0] ? dbs[0]->k : 0;              // (one k for the tables of a pass: the tile's minimizers are made once)
    for (int i = 0; i < n_dbs; i++) {
        ss_db *db = dbs[i];
        if (!db || db->layout != 1 || db->k != k_all) return SS_EINVAL;
        tabs.mkeys[i] = db->d_mkeys;
        tabs.pages[i] = reinterpret_cast<const uint4 *>(db->d_dir);
        tabs.counts[i] = db->d_counts;
        tabs.n_pages[i] = db->n_dir;
        tabs.cbase[i] = (uint32_t)db->n_mslots;
        expect = expect && db->expect_hits;
    }
    tabs.n = n_dbs;
    const bool aligned = (((uintptr_t)bases_dev) & 15) == 0;
    static const int comb_env = [] { const char *e = getenv("SS_COMBINE"); return e ? atoi(e) : -1; }();
    constexpr uint32_t swz = 1u;
    const bool comb = expect && (comb_env < 0 ? binned : comb_env != 0);
    const uint64_t n_tiles = (n + MTILE - 1) / MTILE, units = comb ? (n_tiles + COMB_CH - 1) / COMB_CH : n_tiles;
    unsigned blocks = (unsigned)std::min<uint64_t>(units, (uint64_t)2048 * 256 * (256 / MT));
    blocks = (blocks + 7u) & ~7u;
    const uint8_t *b = (const uint8_t *)bases_dev;
#define SS_LAUNCH_M(A, C_, LB) do {                                                                                                          \
        if (k_all == 31) hipLaunchKernelGGL((scan_mini_kernel<A, false, C_, LB, true, 31>), dim3(blocks), dim3(MT), 0, stream, b, n, n_tiles,      \
                                            tabs.mkeys[0], tabs.pages[0], tabs.n_pages[0], tabs.counts[0], tabs.cbase[0],                        \
                                            (const uint32_t *)nullptr, 0u, swz, tabs, 31);                                                       \
        else hipLaunchKernelGGL((scan_mini_kernel<A, false, C_, (LB > 5 ? 5 : LB), true, 0>), dim3(blocks), dim3(MT), 0, stream, b, n, n_tiles,      \
                                tabs.mkeys[0], tabs.pages[0], tabs.n_pages[0], tabs.counts[0], tabs.cbase[0],                                    \
                                (const uint32_t *)nullptr, 0u, swz, tabs, k_all);                                                                \
    } while (0)
    if (comb) { if (aligned) SS_LAUNCH_M(true, true, 6); else SS_LAUNCH_M(false, true, 6); }
    else      { if (aligned) SS_LAUNCH_M(true, false, 8); else SS_LAUNCH_M(false, false, 8); }
#undef SS_LAUNCH_M
    SS_HIP(hipGetLastError());
    for (int i = 0; i < n_dbs; i++) dbs[i]->launches++;
    return SS_OK;
}

#ifdef SS_COMB_STATS
extern "C" int ss_debug_comb_stats(unsigned long long *out8 /* [16] */, int reset)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out8, HIP_SYMBOL(ss_comb_stats), 128);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(ss_comb_stats), z, 128); }
    return 0;
}
#endif
#ifdef SS_TIMING
extern "C" int ss_debug_timing(unsigned long long *out32, int reset)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out32, HIP_SYMBOL(ss_timing), 256);
    if (reset) { unsigned long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(ss_timing), z, 256); }
    return 0;
}
#endif

// Which kernel scans a BINNED block (ss_reorder.hip: the reads of a locus lie together) against this table: the plain one,
// whose every hit is a global atomic, or the combining one (COMB: the hits of four consecutive tiles are added up in LDS
// first).  With few hits the plain kernel is the faster (tree tables at their usual 5 % of the read k-mers: 3.8 vs 4.5 ms
// per 20 M reads on sampled node sets, 2.7 vs 3.2 on contiguous ones); with many, the same-address atomics of a locus'
// reads queue up and the combining one wins by up to 3x (half of the read k-mers in the table: 17.1 -> 10.6 ms sampled,
// 14.8 -> 4.9 contiguous; file order 14.1 / 10.2; profiles/r05_ab_log.md 1).  The caller's flag (ss_db_expect_hits: the table of
// an identified cluster) is a hint that saves the probe; every other table is asked: its first PROBE_TILES tiles run
// through the plain kernel with the probe bit set -- they count for real, the main launch starts behind them -- and
// report their found runs (minimizers of the reads that have a bucket in the table); at PROBE_RUNS_PER_TILE and above
// the rest of the block, and every later block of the same read set, goes through the combining kernel.  One probe per
// (read set, table): ~40 us and one stream synchronisation on the first scan of a sample against a table.
constexpr uint64_t PROBE_TILES = 8192;          // one wave per tile: one round of the chip's 8192 wave slots
constexpr double PROBE_RUNS_PER_TILE = 8.0;     // measured crossover: 4 (plain wins by 15 %) ... 8-10 (even) ... 16 (COMB by 30 %)
static std::mutex probe_mu;                     // the probe's device words are one set per process

// (the probe's verdict is read and written under probe_mu: two threads may scan the same table)
static bool probe_known(ss_db *db, uint64_t set_id, bool *comb)
{
    std::lock_guard<std::mutex> g(probe_mu);
    if (db->probe_set != set_id) return false;
    *comb = db->probe_comb != 0;
    return true;
}

static int launch_plain_or_comb(ss_db *db, bool comb, bool probe, const uint8_t *b, uint64_t n, uint64_t n_tiles, hipStream_t stream)
{
    const bool aligned = (((uintptr_t)b) & 15) == 0;
    // grid-stride over tiles with MANY more blocks than fit the chip: short blocks start at scattered times, so
    // the waves sharing a SIMD stop marching through their ALU and memory phases in step.  Measured with 8 waves
    // per SIMD resident (20 M reads = 3.04 M tiles; blocks = x * 1024): x = 8 (one round of resident blocks)
    // 4.03 ms, 32: 3.76, 128: 3.59, 512: 3.55, 2048 (1.5 tiles per block): 3.50, 4096 (one tile each): 3.51
    const uint64_t units = comb ? (n_tiles + COMB_CH - 1) / COMB_CH : n_tiles;
    unsigned blocks = (unsigned)std::min<uint64_t>(units, (uint64_t)2048 * 256 * (256 / MT));
    blocks = (blocks + 7u) & ~7u;                           // a multiple of 8: the same number of workgroups on every XCD
    // (the combining variant needs 71 VGPRs: at 8 waves per SIMD it would spill four of them to scratch)
    if (comb) launch_lb<6>(aligned, comb, blocks, stream, b, n, n_tiles, db, probe);
    else launch_lb<8>(aligned, comb, blocks, stream, b, n, n_tiles, db, probe);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int launch_scan_mini(ss_db *db, const void *bases_dev, uint64_t n, hipStream_t stream, unsigned /*blocks*/,
                     uint64_t n_tiles, bool binned, uint64_t set_id)
{
    // Which kernel.  k = 31: scan_mini_kernel with k a constant (everything below).  17 <= k <= 30: scan_mini_kernel with k at run time
    // (KK = 0, queues of 256 runs) -- measured against the one-lane-per-position kernel scan_minik_kernel (profiles/r06_k_index.json):
    // a table with few hits, 4 M reads, k = 29 / 25 / 21 / 20 / 19 / 17: 1.38 / 1.51 / 1.73 / 1.82 / 2.63 / 3.46 ms against 2.24 / 2.31 /
    // 2.75 / 2.95 / 3.18 / 3.64 (k = 31: 1.34); a cluster table under binned reads (the combining variant), 8 M reads, k = 29 / 25 / 21 /
    // 20: 3.3 / 4.1 / 7.3 / 9.1 against 8.9 / 9.9 / 15.3 / 16.6.  The exception: tables that expect hits at k <= 19, where a tile's
    // ~330+ runs overflow the queues and the per-position kernel is as fast or faster (k = 19: 18.5 binned / 18.9 in file order
    // against 17.5 / 22.2).
    // ss_test_hook 4 (tests: the kernels held to each other on one index): 1 = tables of k = 31 through the per-position kernel,
    // 2 = tables of every k through it, 3 = tables of every k through scan_mini_kernel.
    {
        const long long hk = g_hook_generic_k.load();
        if (db->k == 31 ? (hk == 1 || hk == 2) : (hk != 3 && (hk == 2 || (db->k <= 19 && db->expect_hits))))
            return launch_scan_minik(db, (const uint8_t *)bases_dev, n, stream);
        if (db->k != 31) {                                  // (no probe: the flag decides)
            const uint64_t nt = (n + MTILE - 1) / MTILE;
            return launch_plain_or_comb(db, binned && db->expect_hits, false, (const uint8_t *)bases_dev, n, nt, stream);
        }
    }
    n_tiles = (n + MTILE - 1) / MTILE;                      // this kernel's tile is 62 x 16 positions
    const uint8_t *b = (const uint8_t *)bases_dev;
    // SS_COMBINE (A/B runs and tests): 0 never, 1 every scan of a table that expects hits -- binned or not --, 2 every binned scan
    static const int comb_env = [] { const char *e = getenv("SS_COMBINE"); return e ? atoi(e) : -1; }();
    bool comb = false;
    if (comb_env >= 0) comb = comb_env == 2 ? binned : (db->expect_hits && comb_env != 0);
    else if (!binned) comb = false;
    else if (db->expect_hits) comb = true;
    else if (set_id && probe_known(db, set_id, &comb)) {}
    else if (n_tiles >= 4 * PROBE_TILES) {
        // The first scan of a (read set, table) pair makes ONE stream synchronisation here (~40 us of probe tiles, a memset and a
        // 256-byte copy back): ss_scan_reads is asynchronous from the second scan of the pair on, and this first launch cannot
        // be captured into a hipGraph (include/strainscan_hip.h says so at ss_scan_reads).
        std::lock_guard<std::mutex> g(probe_mu);
        uint32_t runs[64];
        void *sym = nullptr;
        SS_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(ss_probe_runs)));
        SS_HIP(hipMemsetAsync(sym, 0, sizeof(runs), stream));
        int rc = launch_plain_or_comb(db, false, true, b, n, PROBE_TILES, stream);      // (n: the last tile's k-mers reach beyond it)
        if (rc) return rc;
        SS_HIP(hipMemcpyAsync(runs, sym, sizeof(runs), hipMemcpyDeviceToHost, stream));
        SS_HIP(hipStreamSynchronize(stream));
        uint64_t total = 0;
        for (uint32_t r : runs) total += r;
        db->probe_runs_per_tile = (double)total / (double)PROBE_TILES;
        comb = db->probe_runs_per_tile >= PROBE_RUNS_PER_TILE;
        db->probe_comb = comb;
        db->probe_set = set_id;
        b += PROBE_TILES * (uint64_t)MTILE;                 // (a multiple of 16 bytes: the alignment of the block is kept)
        n -= PROBE_TILES * (uint64_t)MTILE;
        n_tiles -= PROBE_TILES;
    }
    return launch_plain_or_comb(db, comb, false, b, n, n_tiles, stream);
}

}  // namespace ss

// ---------------------------------------------------------------------------------------------
// Index image on disk: the built minimizer index (device arrays) dumped verbatim, so that a
// database is indexed once, not at every run (device image cache).
// ---------------------------------------------------------------------------------------------
namespace {
// An imported image is checked before it is used: every index that the scan or gather kernels will follow must stay
// inside its array (a truncated-and-padded or overwritten cache file must fail here, not read out of bounds later).
__global__ void validate_image_kernel(const uint32_t *__restrict__ slot_of_row, uint64_t n_rows, uint64_t n_slots,
                                      const uint8_t *__restrict__ pages, uint64_t n_pages, const uint64_t *__restrict__ mkeys,
                                      uint64_t n_mslots, uint32_t e_max /* k - 15 */, uint32_t *__restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows) {
        const uint32_t sl = slot_of_row[i];
        if (sl != SS_NO_SLOT && sl >= n_slots) atomicAdd(bad, 1u);
    }
    if (i < n_pages * 8) {
        const uint8_t *pp = pages + (i >> 3) * 64;
        const uint32_t sl = (uint32_t)(i & 7), hi8 = pp[8 + sl];
        if (hi8 & 0x80u) {                       // bucket reference: header + candidates inside d_mkeys
            uint32_t lo;
            memcpy(&lo, pp + 16 + 4 * sl, 4);
            const uint64_t start = lo & ss::START_MASK;
            if (start + 1 >= n_mslots) atomicAdd(bad, 1u);
            else {
                const uint64_t cnt = mkeys[start] >> 32;          // header at start, k-mers at start + 1 .. start + cnt
                if (cnt < 1 || cnt > n_mslots || start + cnt >= n_mslots) atomicAdd(bad, 1u);
            }
        } else if (hi8 != ss::PG_EMPTY_HI && (hi8 & 31u) > e_max) atomicAdd(bad, 1u);
    }
}

struct ImageHeader {
    char magic[8];          // "SSIDX10\0" (10: PG_SOLID flags in the bucket references)
    int32_t k, layout;
    uint64_t n_rows, n_distinct, n_slots, n_buckets, n_mslots, n_inline;
    uint32_t n_dir, bloom_bits, n_dir_alloc, reserved;
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
    memcpy(h.magic, "SSIDX10", 8);
    h.k = db->k; h.layout = db->layout;
    h.n_rows = db->n_rows; h.n_distinct = db->n_distinct; h.n_slots = db->n_slots; h.n_buckets = db->n_buckets;
    h.n_mslots = db->n_mslots; h.n_inline = db->n_inline;
    h.n_dir = db->n_dir; h.n_dir_alloc = db->n_dir_alloc;
    h.bloom_bits = db->d_bloom ? db->bloom_bits : 0;
    const uint64_t nr = std::max<uint64_t>(1, db->n_rows);
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && write_dev(f, db->d_mkeys, db->n_mslots * 8) &&
              write_dev(f, db->d_dir, (uint64_t)db->n_dir_alloc * 64) && write_dev(f, db->d_slot_of_row, nr * 4) &&
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
    if (fstat(fd, &st) != 0 || pread(fd, &h, sizeof(h), 0) != (ssize_t)sizeof(h) || memcmp(h.magic, "SSIDX10", 8) != 0 ||
        h.layout != 1 || h.k < ss::MINI_K_MIN || h.k > 31 || h.n_mslots == 0 || h.n_dir < ss::PG_MIN_PAGES || h.n_dir_alloc != h.n_dir + h.n_dir / 1024 || h.n_slots != h.n_mslots + (uint64_t)h.n_dir_alloc * 8 ||
        h.n_slots >= 0xFFFFFFF0ull || h.n_mslots >= (uint64_t)ss::START_MASK || (h.bloom_bits && (h.bloom_bits < 10 || h.bloom_bits > 30))) {
        close(fd);
        return SS_EINVAL;
    }
    const uint64_t nr = std::max<uint64_t>(1, h.n_rows);
    const uint64_t sizes[5] = {h.n_mslots * 8, (uint64_t)h.n_dir_alloc * 64, nr * 4, nr, h.bloom_bits ? (1ull << h.bloom_bits) / 8 : 0};
    uint64_t offs[6] = {sizeof(h), 0, 0, 0, 0, 0};
    for (int i = 0; i < 5; i++) offs[i + 1] = offs[i] + sizes[i];
    if ((uint64_t)st.st_size != offs[5]) { close(fd); return SS_EIO; }     // the file must be exactly the image
    ss_db *db = new (std::nothrow) ss_db();
    if (!db) { close(fd); return SS_ENOMEM; }
    db->k = h.k; db->layout = 1;
    db->n_rows = h.n_rows; db->n_distinct = h.n_distinct; db->n_slots = h.n_slots; db->capacity = h.n_slots;
    db->n_mslots = h.n_mslots; db->n_inline = h.n_inline;
    db->n_buckets = h.n_buckets; db->n_dir = h.n_dir; db->n_dir_alloc = h.n_dir_alloc;
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
    if (ok) {
        uint32_t *d_bad = nullptr, bad = 1;
        const uint64_t nchk = std::max<uint64_t>(h.n_rows, (uint64_t)h.n_dir_alloc * 8);
        ok = hipMalloc((void **)&d_bad, 4) == hipSuccess && hipMemset(d_bad, 0, 4) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(validate_image_kernel, dim3((unsigned)((nchk + 255) / 256)), dim3(256), 0, 0, db->d_slot_of_row, h.n_rows,
                               h.n_slots, (const uint8_t *)db->d_dir, (uint64_t)h.n_dir_alloc, db->d_mkeys, h.n_mslots, (uint32_t)(h.k - ss::MINI_M), d_bad);
            ok = hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost) == hipSuccess && bad == 0;
        }
        hipFree(d_bad);
    }
    if (ok && h.bloom_bits) db->bloom_bits = h.bloom_bits;
    if (!ok) { ss_db_destroy(db); return SS_EIO; }
    db->device_bytes = db->n_mslots * 8 + db->n_slots * 4 + (uint64_t)db->n_dir_alloc * 64 + nr * 5 + sizes[4];
    *out = db;
    return SS_OK;
}

}  // extern "C"
