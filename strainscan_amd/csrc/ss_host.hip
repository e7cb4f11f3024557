// ss_host.hip -- host side of the C ABI: error text, device helpers, k-mer FASTA rows,
// FASTA/FASTQ -> flat base blocks (what jellyfish's sequence parser hands its counter), revcomp.
#include "ss_common.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <zlib.h>
#include <hipcub/hipcub.hpp>

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <immintrin.h>
#define SS_HOST_X86 1
#endif

#include <algorithm>
#include <atomic>
#include <memory>
#include <string>
#include <condition_variable>
#include <chrono>
#include <thread>
#include <vector>

namespace ss {
static thread_local char g_last_error[512] = "";
void set_last_error(const char *what, const char *file, int line, hipError_t e)
{
    snprintf(g_last_error, sizeof(g_last_error), "%s failed at %s:%d: %s", what, file, line, hipGetErrorString(e));
}
}  // namespace ss

namespace {

// IUPAC complement of library/seqpy.c:5-22 stated as a rule (letters only, case kept; '`' -> '@').
__host__ __device__ inline unsigned char comp_base(unsigned char c)
{
    const char *up = "TVGHEFCDIJMLKNOPQYSAABWXRZ";
    if (c >= 'A' && c <= 'Z') return (unsigned char)up[c - 'A'];
    if (c >= 'a' && c <= 'z') return (unsigned char)(up[c - 'a'] + 32);
    if (c == 0x60) return 0x40;
    return c;
}

__global__ void revcomp_kernel(const char *__restrict__ in, char *__restrict__ out, uint64_t seq_len, uint64_t n_seq)
{
    const uint64_t total = seq_len * n_seq;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t s = i / seq_len, p = i - s * seq_len;
        out[s * seq_len + (seq_len - 1 - p)] = (char)comp_base((unsigned char)in[i]);
    }
}

bool read_whole_file(const char *path, std::string &out)
{
    gzFile f = gzopen(path, "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    std::vector<char> buf(8 << 20);
    for (;;) {
        int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) { gzclose(f); return false; }
        if (n == 0) break;
        out.append(buf.data(), (size_t)n);
    }
    gzclose(f);
    return true;
}

// rows of a k-mer FASTA exactly like f.readlines()[2*i+1].rstrip()  (identify.py:92-94)
struct RowSpan { uint64_t s, e; };

void split_rows(const char *t, uint64_t len, std::vector<RowSpan> &rows)
{
    uint64_t pos = 0, line = 0;
    while (pos < len) {
        uint64_t s = pos;
        const void *nl = memchr(t + pos, '\n', len - pos);
        uint64_t e = nl ? (uint64_t)((const char *)nl - t) : len;
        pos = nl ? e + 1 : len;
        if (line & 1) {
            while (e > s && (t[e - 1] == ' ' || t[e - 1] == '\r' || t[e - 1] == '\t' || t[e - 1] == '\v' ||
                             t[e - 1] == '\f'))
                e--;
            rows.push_back({s, e});
        }
        line++;
    }
}

void encode_rows(const char *t, const RowSpan *rows, uint64_t n, int k, uint64_t *keys, uint8_t *flags)
{
    for (uint64_t i = 0; i < n; i++) {
        keys[i] = 0;
        flags[i] = 0;
        if (rows[i].e - rows[i].s != (uint64_t)k) continue;
        uint64_t key = 0;
        uint8_t f = SS_ROW_VALID;
        for (int j = 0; j < k; j++) {
            unsigned char c = (unsigned char)t[rows[i].s + j];
            int code = ss::base_code(c);
            if (code < 0) { f = 0; break; }
            if (c >= 'a') f |= SS_ROW_LOWER;
            key |= (uint64_t)code << (2 * j);
        }
        if (f) { keys[i] = key; flags[i] = f; }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// FASTA/FASTQ reader: the record grammar jellyfish 2.3.0 accepts [probed, tests/golden/f1]:
//   '>' header, sequence lines up to the next '>' line;
//   '@' header, sequence lines up to a '+' line, then quality lines until as many quality
//   characters as sequence characters were consumed; blank lines between records are skipped;
//   the sequence lines of one record are concatenated.
// Output: sequence bytes, one '\n' after each record.  A record that does not fit the caller's
// buffer is cut, the continuation re-emitting the last `overlap` bases so that no k-mer
// (k <= overlap + 1) is lost or counted twice.
// ---------------------------------------------------------------------------------------------
struct ss_reader {
    std::vector<std::string> paths;
    size_t file_idx = 0;
    gzFile f = nullptr;
    std::vector<char> in;
    size_t in_pos = 0, in_len = 0;
    bool eof_file = true;
    enum State { START, FA_HDR, FA_SEQ_BOL, FA_SEQ, FQ_HDR, FQ_SEQ_BOL, FQ_SEQ, FQ_PLUS, FQ_QUAL } st = START;
    uint64_t seqlen = 0, qlen = 0;
    int overlap = 30;
    std::string tail;      // last `overlap` bases of the current record (for cut records)
    bool need_tail = false;
};

namespace {

bool reader_fill(ss_reader *r)
{
    for (;;) {
        if (r->f) {
            int n = gzread(r->f, r->in.data(), (unsigned)r->in.size());
            if (n > 0) { r->in_pos = 0; r->in_len = (size_t)n; return true; }
            gzclose(r->f);
            r->f = nullptr;
            // end of file ends the record in flight
            return false;
        }
        return false;
    }
}

}  // namespace

extern "C" {

int ss_version(void) { return 100; }

const char *ss_strerror(int code)
{
    switch (code) {
    case SS_OK: return "ok";
    case SS_EINVAL: return "invalid argument";
    case SS_ENOMEM: return "out of memory";
    case SS_EIO: return "I/O error";
    case SS_EHIP: return "HIP runtime error";
    case SS_ENODEV: return "no usable GPU";
    case SS_EKEY: return "k-mer without an owning row (KeyError in the reference)";
    case SS_ERANGE: return "value out of supported range";
    case SS_EAGAIN: return "the device path declined a .gz input under the strict policy (ss_gz_set_policy)";
    default: return "unknown error";
    }
}

const char *ss_last_error(void) { return ss::g_last_error; }

int ss_device_count(int *n)
{
    if (!n) return SS_EINVAL;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; ss::set_last_error("hipGetDeviceCount", __FILE__, __LINE__, e); return SS_ENODEV; }
    *n = c;
    return SS_OK;
}

int ss_set_device(int dev) { SS_HIP(hipSetDevice(dev)); return SS_OK; }
int ss_device_sync(void) { SS_HIP(hipDeviceSynchronize()); return SS_OK; }
int ss_stream_sync(void *stream) { SS_HIP(hipStreamSynchronize(ss::as_stream(stream))); return SS_OK; }

}  // extern "C"
void ss::pool_keep_at_least(uint64_t bytes)
{
    static std::mutex mu;
    static uint64_t now = 0;
    std::lock_guard<std::mutex> g(mu);
    if (bytes <= now) return;
    int dev = 0;
    hipMemPool_t pool = nullptr;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess &&
        hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &bytes) == hipSuccess)
        now = bytes;
}
namespace {
struct BigBlock { void *p; uint64_t cap; int dev; };      // dev: the device the block lives on -- a block is only ever served there
std::mutex g_big_mu;
std::vector<BigBlock> g_big;
uint64_t g_big_served = 0;
constexpr uint64_t BIG_MIN = 256ull << 20, BIG_TOTAL = 24ull << 30;
constexpr size_t BIG_N = 3;
}
void *ss::big_take(uint64_t bytes, uint64_t *cap)
{
    if (bytes < BIG_MIN) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(g_big_mu);
    size_t best = g_big.size();
    for (size_t i = 0; i < g_big.size(); i++)
        if (g_big[i].dev == dev && g_big[i].cap >= bytes && g_big[i].cap <= bytes / 2 * 5 && (best == g_big.size() || g_big[i].cap < g_big[best].cap)) best = i;
    if (best == g_big.size()) return nullptr;
    void *p = g_big[best].p;
    if (cap) *cap = g_big[best].cap;
    g_big.erase(g_big.begin() + (long)best);
    g_big_served++;
    return p;
}
void ss::big_put(void *p, uint64_t cap)
{
    if (!p) return;
    if (cap >= BIG_MIN) {
        // The caller's contract (ss_common.h): nothing in flight still touches the block -- hipFree would have waited for the
        // device, keeping a block does not, so the callers that cannot know (ss_reads_destroy) synchronise first.
        hipPointerAttribute_t at;
        int dev = -1;
        if (hipPointerGetAttributes(&at, p) == hipSuccess) dev = at.device; else (void)hipGetLastError();
        std::lock_guard<std::mutex> g(g_big_mu);
        uint64_t held = 0;
        for (const auto &b : g_big) held += b.cap;
        if (dev >= 0 && g_big.size() < BIG_N && held + cap <= BIG_TOTAL) { g_big.push_back({p, cap, dev}); return; }
    }
    hipFree(p);
}
void ss::big_release()
{
    std::vector<BigBlock> all;
    {
        std::lock_guard<std::mutex> g(g_big_mu);
        all.swap(g_big);
    }
    for (const auto &b : all) hipFree(b.p);
}
hipError_t ss::big_malloc(void **p, uint64_t bytes, uint64_t *got)
{
    if (got) *got = bytes;
    if ((*p = big_take(bytes, got)) != nullptr) return hipSuccess;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        big_release();
        e = hipMalloc(p, bytes);
    }
    return e;
}
extern "C" int ss_dev_big_blocks(uint64_t out[3])
{
    if (!out) return SS_EINVAL;
    std::lock_guard<std::mutex> g(g_big_mu);
    out[0] = g_big.size(); out[1] = 0; out[2] = g_big_served;
    for (const auto &b : g_big) out[1] += b.cap;
    return SS_OK;
}
extern "C" int ss_dev_big_release(void) { ss::big_release(); return SS_OK; }
hipError_t ss::l2s::dmalloc(void **p, size_t n)
{
    static std::once_flag once;
    std::call_once(once, [] { ss::pool_keep_at_least(2ull << 30); });      // freed temporaries stay in the pool instead of going back at every synchronisation
    return hipMallocAsync(p, n ? n : 1, stream());
}
extern "C" {
/* stream-ordered twins of ss_dev_alloc / ss_dev_free on the calling thread's stream (layer 2's temporaries) */
int ss_dev_alloc_async(void **dptr, uint64_t bytes)
{
    if (!dptr) return SS_EINVAL;
    SS_HIP(ss::l2s::dmalloc(dptr, bytes));
    SS_HIP(ss::l2s::sync());               // the allocation has happened: the memory may be used from any stream
    return SS_OK;
}
int ss_dev_free_async(void *dptr) { SS_HIP(ss::l2s::dfree(dptr)); return SS_OK; }

int ss_dev_alloc(void **dptr, uint64_t bytes)
{
    if (!dptr) return SS_EINVAL;
    SS_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return SS_OK;
}
int ss_dev_free(void *dptr) { SS_HIP(hipFree(dptr)); return SS_OK; }
int ss_memcpy_h2d(void *dst, const void *src, uint64_t bytes, void *stream)
{
    SS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ss::as_stream(stream)));
    SS_HIP(hipStreamSynchronize(ss::as_stream(stream)));
    return SS_OK;
}
int ss_memcpy_d2h(void *dst, const void *src, uint64_t bytes, void *stream)
{
    SS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ss::as_stream(stream)));
    SS_HIP(hipStreamSynchronize(ss::as_stream(stream)));
    return SS_OK;
}
int ss_memset_dev(void *dst, int byte, uint64_t bytes, void *stream)
{
    SS_HIP(hipMemsetAsync(dst, byte, bytes, ss::as_stream(stream)));
    return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// ShuffleSplit(n_splits, test_size, random_state=seed) of scikit-learn 0.23 (model_selection/_split.py):
// for every split `rng.permutation(n)[:n_test]` is the test set, rng = numpy.random.RandomState(seed).
// Restated: MT19937 seeded by init_genrand(seed) (numpy _legacy_seeding for an integer), permutation(n) =
// Fisher-Yates from the top -- for i = n-1 .. 1: j = random_interval(i); swap(x[i], x[j]) -- with
// random_interval(max) = 32-bit draws masked to the next power of two minus one, redrawn while > max
// (numpy/random/src/distributions: random_interval; mtrand.pyx: _shuffle_raw).
// What the splits share is ONE word stream: where a split's words begin depends on how many draws the splits before it had
// to repeat.  So one thread walks the stream and only COUNTS (64 words per step on AVX-512 hosts, see reject_avx512; a
// compare and an add-with-carry per word elsewhere), and hands every split's worker a snapshot of the generator (22 KB: its
// state and the block it is in) at the split's first word; the worker goes through the same words again, this time keeping the accepted draws -- a few
// thousand at a time, in its L1 -- and applies the swaps (the cache misses of a split are its own).  Round 3: one thread did
// everything, 165 ms for 5 M rows x 20 splits; round 4: generator thread -> rejection thread -> swap threads with 20 MB of swap
// partners per split between them, 114 ms (bound first by the compare chain, then by the generator's lines crossing between
// two CCDs and 400 MB of partners going through memory); now 24 ms for the walk + one worker's 12 ms behind it on the
// GPU boxes' EPYC 9575F (16 CPUs granted).
// Pinned against numpy itself in tests/test_abi_and_host.py and against sklearn's ShuffleSplit golden.
// ---------------------------------------------------------------------------------------------
namespace {
// MT19937 as a LINEAR recurrence over a buffer: x[0..623] = the state, x[624 + n] = x[n + 397] ^ twist(x[n], x[n + 1]) -- the
// in-place generator unrolled; output word n of the stream = temper(x[624 + n]).  A word depends on words 227 and more
// places behind it, so the loop vectorises; with AVX2 it runs 2.4 x as fast as with the baseline SSE2.
__attribute__((always_inline)) inline uint32_t mt_temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
__attribute__((always_inline)) inline void mt_extend_body(uint32_t *x, int n_new)
{
#pragma clang loop vectorize(assume_safety)
    for (int n = 0; n < n_new; n++) {
        const uint32_t y = (x[n] & 0x80000000u) | (x[n + 1] & 0x7fffffffu);
        x[624 + n] = x[n + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    }
}
__attribute__((always_inline)) inline void mt_temper_body(uint32_t *x, int n)
{
    for (int i = 0; i < n; i++) x[i] = mt_temper(x[i]);
}
#ifdef SS_HOST_X86
__attribute__((target("avx2"))) void mt_extend_avx2(uint32_t *x, int n_new)
{
    // (vector width 8 <= 227: no lane reads what another lane of the same step writes)
    for (int n0 = 0; n0 < n_new; n0 += 224) mt_extend_body(x + n0, std::min(224, n_new - n0));
}
__attribute__((target("avx2"))) void mt_temper_avx2(uint32_t *x, int n) { mt_temper_body(x, n); }
// 16 words per step: three loads, a bit select, a shift, a test and two xors (one masked).  A step reads x[n .. n + 412],
// all of it written at least 14 steps earlier.
__attribute__((target("avx512f"))) void mt_extend_avx512(uint32_t *x, int n_new)
{
    const __m512i up = _mm512_set1_epi32((int)0x80000000u), mag = _mm512_set1_epi32((int)0x9908b0dfu), one = _mm512_set1_epi32(1);
    int n = 0;
    for (; n + 16 <= n_new; n += 16) {
        const __m512i a = _mm512_loadu_si512((const void *)(x + n)), b = _mm512_loadu_si512((const void *)(x + n + 1));
        const __m512i m = _mm512_loadu_si512((const void *)(x + n + 397));
        const __m512i y = _mm512_ternarylogic_epi32(up, a, b, 0xCA);              // up ? a : b
        const __m512i r = _mm512_xor_si512(m, _mm512_srli_epi32(y, 1));
        _mm512_storeu_si512((void *)(x + 624 + n), _mm512_mask_xor_epi32(r, _mm512_test_epi32_mask(y, one), r, mag));
    }
    if (n < n_new) mt_extend_body(x + n, n_new - n);           // (fewer than 16 words: far below the distance of 227)
}
#endif
void mt_extend_base(uint32_t *x, int n_new)
{
    for (int n0 = 0; n0 < n_new; n0 += 224) mt_extend_body(x + n0, std::min(224, n_new - n0));
}
void mt_temper_base(uint32_t *x, int n) { mt_temper_body(x, n); }
bool host_simd_allowed() { static const bool on = !(getenv("SS_SPLIT_SIMD") && !atoi(getenv("SS_SPLIT_SIMD"))); return on; }      // (A/B)
void mt_extend(uint32_t *x, int n_new)
{
#ifdef SS_HOST_X86
    static const bool avx512 = __builtin_cpu_supports("avx512f") && host_simd_allowed();
    static const bool avx2 = __builtin_cpu_supports("avx2") && host_simd_allowed();
    if (avx512) return mt_extend_avx512(x, n_new);
    if (avx2) return mt_extend_avx2(x, n_new);
#endif
    mt_extend_base(x, n_new);
}
void mt_temper_all(uint32_t *x, int n)
{
#ifdef SS_HOST_X86
    static const bool avx2 = __builtin_cpu_supports("avx2") && host_simd_allowed();
    if (avx2) return mt_temper_avx2(x, n);
#endif
    mt_temper_base(x, n);
}

// Word buffers of the splits (swap partners, the permutation itself), kept between splits AND between calls: a fresh 20 MB
// vector per split is zero-filled and page-faulted in by whoever touches it first -- 40 such buffers per call were a third of the
// generator thread's time.  At most 24 buffers / 512 MB stay pooled.
struct WordPool {
    std::mutex mu;
    std::vector<std::pair<uint32_t *, uint64_t>> free_;
    uint32_t *get(uint64_t n)
    {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].second >= n) { uint32_t *p = free_[i].first; free_.erase(free_.begin() + (long)i); return p; }
        }
        return static_cast<uint32_t *>(malloc(std::max<uint64_t>(n, 1) * sizeof(uint32_t)));
    }
    void put(uint32_t *p, uint64_t n)
    {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(mu);
            uint64_t held = 0;
            for (auto &f : free_) held += f.second;
            if (free_.size() < 24 && (held + n) * sizeof(uint32_t) <= (512ull << 20)) { free_.emplace_back(p, n); return; }
        }
        free(p);
    }
};
WordPool g_words;

// The word stream, generated where it is consumed: 624 x 8 words at a time behind a copy of the 624 words before them, in a
// buffer that stays in the consumer's L1/L2.  `raw`: the words are handed out UNTEMPERED (the 64-words-per-step consumer tempers
// them in its vector registers).  Plain data: a copy is a snapshot from which the same words come again.
struct MTWords {
    static constexpr int BLK = 624 * 8;
    uint32_t hist[624];                       // the last 624 words generated, untempered (at first: init_genrand(seed))
    uint32_t x[624 + BLK];
    int pos = BLK;                            // next unread word of x + 624 (BLK: the block is used up)
    bool raw;
    MTWords(uint32_t seed, bool raw_) : raw(raw_)
    {
        hist[0] = seed;
        for (int i = 1; i < 624; i++) hist[i] = 1812433253u * (hist[i - 1] ^ (hist[i - 1] >> 30)) + (uint32_t)i;
    }
    // the unread words of the current block (a new block when none is left): at least one, at most BLK
    const uint32_t *words(int *avail)
    {
        if (pos == BLK) {
            memcpy(x, hist, sizeof(hist));
            mt_extend(x, BLK);
            memcpy(hist, x + BLK, sizeof(hist));
            if (!raw) mt_temper_all(x + 624, BLK);
            pos = 0;
        }
        *avail = BLK - pos;
        return x + 624 + pos;
    }
};
}  // namespace

// The rejection rule over a stretch of the word stream, 64 words per step (AVX-512).  i is the row the next accepted draw
// belongs to; a draw v is accepted when v <= i.  Within a step i falls by at most 63, so v <= i - 63 is accepted and v > i
// rejected whatever happened to the words before it -- the dependent chain is ONE count per 64 words instead of a compare and
// an add-with-carry per word.  A draw inside that band of 63 values (one word in 2^k / 64, k >= 14 here) sends the step to
// the scalar rule.  The words arrive untempered and are tempered here.  STORE: the accepted draws are written in stream
// order to out[0 ..] (up to 15 words of slack behind them); otherwise they are only counted.
// Returns the words consumed (a multiple of the step, <= n_words); *i_io falls by the draws accepted.
extern "C++" {
#ifdef SS_HOST_X86
template <bool STORE, int NV>            // NV vectors of 16 words per step: 4 (a band of 63 values)
__attribute__((target("avx512f,popcnt"))) static int reject_avx512(const uint32_t *w, int n_words, uint32_t mask, uint32_t *i_io, uint32_t *out)
{
    constexpr int STEP = 16 * NV;
    const uint32_t i0 = *i_io;
    uint32_t c = 0;                                            // draws accepted so far
    const __m512i vmask = _mm512_set1_epi32((int)mask), t7 = _mm512_set1_epi32((int)0x9d2c5680u), t15 = _mm512_set1_epi32((int)0xefc60000u);
    int k = 0;
    for (; k + STEP <= n_words; k += STEP) {
        const uint32_t i = i0 - c;
        const __m512i hi = _mm512_set1_epi32((int)i), lo = _mm512_set1_epi32((int)(i - (uint32_t)(STEP - 1)));
        __m512i v[NV];
        __mmask16 a[NV];
        unsigned unsure = 0;
#pragma unroll
        for (int q = 0; q < NV; q++) {
            __m512i y = _mm512_loadu_si512((const void *)(w + k + 16 * q));
            y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 11));
            y = _mm512_ternarylogic_epi32(y, _mm512_slli_epi32(y, 7), t7, 0x78);       // a ^ (b & c)
            y = _mm512_ternarylogic_epi32(y, _mm512_slli_epi32(y, 15), t15, 0x78);
            y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 18));
            v[q] = _mm512_and_si512(y, vmask);
            a[q] = _mm512_cmple_epu32_mask(v[q], lo);
            unsure |= (unsigned)(a[q] ^ _mm512_cmple_epu32_mask(v[q], hi));
        }
        if (__builtin_expect(unsure != 0, 0)) {
            uint32_t ii = i;
            for (int q = 0; q < STEP; q++) {
                const uint32_t x = mt_temper(w[k + q]) & mask;
                if (STORE) out[i0 - ii] = x;
                ii -= (uint32_t)(x <= ii);
            }
            c = i0 - ii;
            continue;
        }
#pragma unroll
        for (int q = 0; q < NV; q++) {
            if (STORE) _mm512_storeu_si512((void *)(out + c), _mm512_maskz_compress_epi32(a[q], v[q]));
            c += (uint32_t)__builtin_popcount((unsigned)a[q]);
        }
    }
    *i_io = i0 - c;
    return k;
}
#endif

namespace {
// One split's walk over the word stream: permutation(n) = Fisher-Yates from the top, row i = n - 1 .. 1 takes the first draw
// (masked to the next power of two above i, minus one) that is <= i as its partner.  STORE: the partners go to
// sink(partners, count) in stream order, a few thousand at a time (`buf`: CH + BLK + 16 words); otherwise the stream is only
// moved past the split's words -- that is all the splits have in common, the rest of a split's work is its own.
constexpr uint32_t SPLIT_CH = 2048;
// `stop` (>= 1): the walk ends once row `stop` has its partner -- a worker needs the rows from n - 1 down to n_test only (see
// split_worker); the walker of the shared stream passes 1 and goes through all of them.
// DIRECT (with STORE): `buf` is the whole output -- room for every partner of the rows walked + SPLIT_CH + BLK + 16 words --, nothing
// is handed to the sink before the end (the caller reads the count there).  `start`: the first row (default n - 1).
template <bool STORE, bool SIMD, bool DIRECT = false, class Sink>
void walk_split(MTWords &rng, uint64_t n, uint32_t *buf, Sink &&sink, uint64_t stop = 1, uint64_t start = ~0ull)
{
    uint64_t fill = 0;
    for (uint64_t hi = start == ~0ull ? n - 1 : start; hi >= stop;) {
        // all i in (mask >> 1, hi] share the mask: the next power of two above i, minus one
        uint32_t mask = (uint32_t)hi;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        const uint64_t lo = std::max<uint64_t>((uint64_t)(mask >> 1) + 1, stop);
        for (uint64_t i = hi; i >= lo;) {
            // at most `room` words can be consumed before i drops below lo: no bound check on i inside
            const uint64_t room = i - lo + 1;
            int avail = 0;
            const uint32_t *ob = rng.words(&avail);
            avail = (int)std::min<uint64_t>((uint64_t)avail, room);
            const int taken = avail;
            uint32_t ii = (uint32_t)i;
#ifdef SS_HOST_X86
            // (from rows of 8192 on: below, more than a third of the steps meet a draw inside the band and fall back to the scalar
            //  rule anyway; with the levels up to 131072 walked word by word, as at first, those 2.6 % of the rows were 8 ms of a
            //  19 ms walk -- tempering on the scalar unit is 8 cycles a word)
            // (steps of 128 words -- NV = 8 -- for the large levels: the walk 24.1 instead of 18.0 ms on the boxes' EPYC 9575F, slower
            //  on a Xeon too: not the chain through the count but the work per word is what a step costs)
            if (SIMD && lo >= 8192 && avail >= 64) {
                const int used = reject_avx512<STORE, 4>(ob, avail, mask, &ii, buf + fill + ((uint32_t)i - ii));
                ob += used;
                avail -= used;
            }
            if (SIMD && STORE) fill += (uint64_t)((uint32_t)i - ii);
#endif
            // branch-free rejection: every draw is written to its row's place; the row moves on only when the draw is accepted
            const uint32_t i1 = ii;
            for (int k = 0; k < avail; k++) {
                const uint32_t v = (SIMD ? mt_temper(ob[k]) : ob[k]) & mask;
                if (STORE) buf[fill + (i1 - ii)] = v;
                ii -= (uint32_t)(v <= ii);
            }
            if (STORE) {
                fill += i1 - ii;
                if (!DIRECT && fill >= SPLIT_CH) { sink((const uint32_t *)buf, (uint32_t)fill); fill = 0; }
            }
            rng.pos += taken;
            i = ii;
            if (ii < (uint32_t)lo) break;           // (only when lo > 0: i is unsigned)
        }
        hi = lo - 1;
    }
    if (STORE && fill) sink((const uint32_t *)buf, (uint32_t)fill);
}

// the swaps of one split, on a thread of its own, from a snapshot of the stream at the split's first word
template <bool SIMD>
int split_worker(std::unique_ptr<MTWords> rng, uint64_t n, uint64_t n_test, uint64_t *bitmap)
{
    uint32_t *xp = g_words.get(n);
    std::unique_ptr<uint32_t[]> buf(new (std::nothrow) uint32_t[SPLIT_CH + MTWords::BLK + 16]);
    if (!xp || !buf) { g_words.put(xp, n); return SS_ENOMEM; }
    for (uint64_t i = 0; i < n; i++) xp[i] = (uint32_t)i;
    uint64_t i = n - 1;
    // Fisher-Yates from the top: the swap of row i touches row i and a row below it, so once i has gone below n_test the swaps
    // only move elements about INSIDE the first n_test places -- the test SET, all the caller asks for, is complete when row
    // n_test has been placed.  The worker stops its walk and its swaps there (round 5: half of both; only the walker of the
    // shared stream has to pass the remaining words, to know where the next split begins).
    walk_split<true, SIMD>(*rng, n, buf.get(), [&](const uint32_t *p, uint32_t cnt) {
        // (the partners are known ahead: their cache lines are requested early -- 20 MB of x do not fit L2)
        // (... 64 swaps early: 24 / 48 / 64 / 96 / 128 ahead gave workers of 10 / 8 / 8 / 8 / 7 ms and the whole call 29.9 / 27.2 / 27.4 /
        //  28.3 / 27.8 ms on the boxes' EPYC 9575F)
        constexpr uint32_t PD = 64u;
        for (uint32_t k = 0; k < std::min(PD, cnt); k++) __builtin_prefetch(&xp[p[k]], 1, 1);
        for (uint32_t k = 0; k < cnt; k++, i--) {
            if (k + PD < cnt) __builtin_prefetch(&xp[p[k + PD]], 1, 1);
            const uint32_t j = p[k], a = xp[i];
            xp[i] = xp[j];
            xp[j] = a;
        }
    }, std::max<uint64_t>(1, n_test));
    // the test set as a bitmap of the split's own (n / 8 bytes: it stays in this core's L2; atomic ORs into the shared
    // 32-bit words -- half of the rows, from every worker at once -- were as long as the swaps)
    memset(bitmap, 0, ((n + 63) / 64) * sizeof(uint64_t));
    for (uint64_t q = 0; q < n_test; q++) bitmap[xp[q] >> 6] |= 1ull << (xp[q] & 63u);
    g_words.put(xp, n);
    return SS_OK;
}
// rows [64 g0, 64 g1) of the result from the splits' bitmaps: bit f of bits[r] = bit r of bitmap f
#ifdef SS_HOST_X86
__attribute__((target("avx512f"))) void merge_bitmaps_avx512(const uint64_t *bm, uint64_t words, int n_splits, uint64_t g0, uint64_t g1, uint64_t n, uint32_t *bits)
{
    for (uint64_t g = g0; g < g1; g++) {
        __m512i acc[4] = {_mm512_setzero_si512(), _mm512_setzero_si512(), _mm512_setzero_si512(), _mm512_setzero_si512()};
        for (int f = 0; f < n_splits; f++) {
            const uint64_t w = bm[(uint64_t)f * words + g];
            const __m512i bit = _mm512_set1_epi32((int)(1u << f));
            for (int q = 0; q < 4; q++) acc[q] = _mm512_mask_or_epi32(acc[q], (__mmask16)(w >> (16 * q)), acc[q], bit);
        }
        if (g * 64 + 64 <= n) {
            for (int q = 0; q < 4; q++) _mm512_storeu_si512((void *)(bits + g * 64 + 16 * q), acc[q]);
        } else {
            alignas(64) uint32_t t[64];
            for (int q = 0; q < 4; q++) _mm512_store_si512((void *)(t + 16 * q), acc[q]);
            for (uint64_t r = g * 64; r < n; r++) bits[r] = t[r - g * 64];
        }
    }
}
#endif
void merge_bitmaps_base(const uint64_t *bm, uint64_t words, int n_splits, uint64_t g0, uint64_t g1, uint64_t n, uint32_t *bits)
{
    for (uint64_t g = g0; g < g1; g++) {
        uint32_t t[64] = {0};
        for (int f = 0; f < n_splits; f++)
            for (uint64_t w = bm[(uint64_t)f * words + g]; w; w &= w - 1) t[__builtin_ctzll(w)] |= 1u << f;
        for (uint64_t r = g * 64; r < std::min(n, g * 64 + 64); r++) bits[r] = t[r - g * 64];
    }
}
}  // namespace
}  // extern "C++"

namespace {
// Host cores shared between the ShuffleSplit calls of one process (round 5: the clusters of a sample are solved at once, each
// call = one walker of the shared stream + a swap thread per split): a thread takes a core for as long as it computes, so that
// four calls do not put 4 x 15 threads on 16 cores and slow each other's walkers -- the critical path -- down.
struct CoreSlots {
    std::mutex mu;
    std::condition_variable cv;
    int free_ = (int)std::max(2u, ss::host_cpus());
    void take() { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return free_ > 0; }); free_--; }
    void give() { { std::lock_guard<std::mutex> g(mu); free_++; } cv.notify_one(); }
};
CoreSlots g_cores;
struct CoreGuard { CoreGuard() { g_cores.take(); } ~CoreGuard() { g_cores.give(); } };
}  // namespace

int ss_shuffle_split_bits(uint64_t n, int n_splits, uint64_t n_test, uint32_t seed, uint32_t *bits)
{
    if (n_splits < 1 || n_splits > 31 || n_test > n || n > 0xFFFFFFFFull || (n && !bits)) return SS_EINVAL;
    if (n < 2) {
        if (n) bits[0] = n_test ? (1u << n_splits) - 1u : 0u;                         // permutation(1) = [0]
        return SS_OK;
    }
    // swaps of several splits at once: one core each, two stay free
    unsigned in_flight = std::max(1u, std::min(20u, ss::host_cpus() > 3 ? ss::host_cpus() - 2 : 1u));
    static const bool trace = getenv("SS_SPLIT_TRACE") != nullptr;
#ifdef SS_HOST_X86
    static const bool simd = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("popcnt") && host_simd_allowed();
#else
    static const bool simd = false;
#endif
    const auto t_begin = std::chrono::steady_clock::now();
    double t_wait = 0;
    std::unique_ptr<MTWords> rng(new (std::nothrow) MTWords(seed, simd));
    const uint64_t bm_words = (n + 63) / 64;
    uint64_t *bm = reinterpret_cast<uint64_t *>(g_words.get(2 * bm_words * (uint64_t)n_splits));      // (every worker clears its own)
    if (!rng || !bm) { g_words.put(reinterpret_cast<uint32_t *>(bm), 2 * bm_words * (uint64_t)n_splits); return SS_ENOMEM; }
    std::vector<std::thread> pool((size_t)n_splits);
    std::vector<std::pair<double, double>> t_worker((size_t)n_splits);             // (trace: when every worker began and ended)
    std::atomic<int> err(SS_OK);
    std::unique_ptr<CoreGuard> walker_core(new CoreGuard());                          // the walker's own core, until the stream is walked
    for (int f = 0; f < n_splits; f++) {
        if (f >= (int)in_flight) {                                                   // bounds the memory: in_flight x 4 n bytes
            const auto t0 = std::chrono::steady_clock::now();
            walker_core.reset();                                                     // (a thread that waits holds no core)
            pool[(size_t)f - in_flight].join();
            walker_core.reset(new CoreGuard());
            t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        // the split's worker starts from a snapshot of the stream at the split's first word and goes through the split's draws
        // itself; this thread only moves the stream past them (the next split begins where this one's last accepted draw was)
        MTWords *snap = new (std::nothrow) MTWords(*rng);
        if (!snap) { err = SS_ENOMEM; pool[(size_t)f] = std::thread([] {}); break; }
        uint64_t *my_bm = bm + (uint64_t)f * bm_words;
        pool[(size_t)f] = std::thread([snap, n, n_test, f, my_bm, &err, &t_worker, t_begin] {
            CoreGuard core;
            const double w0 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
            const int rc = simd ? split_worker<true>(std::unique_ptr<MTWords>(snap), n, n_test, my_bm) : split_worker<false>(std::unique_ptr<MTWords>(snap), n, n_test, my_bm);
            if (rc != SS_OK) err = rc;
            t_worker[(size_t)f] = {w0, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count()};
        });
        if (f + 1 < n_splits) {
            if (simd) walk_split<false, true>(*rng, n, nullptr, [](const uint32_t *, uint32_t) {});
            else walk_split<false, false>(*rng, n, nullptr, [](const uint32_t *, uint32_t) {});
        }
    }
    const double t_gen = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    walker_core.reset();
    for (auto &th : pool) if (th.joinable()) th.join();
    const double t_workers = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    if (err == SS_OK) {
        // bits[r] = the r-th bits of the splits' bitmaps, row ranges in parallel
        const unsigned parts = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(in_flight, bm_words / 1024));
        std::vector<std::thread> mt;
        for (unsigned q = 0; q < parts; q++) {
            const uint64_t g0 = bm_words * q / parts, g1 = bm_words * (q + 1) / parts;
            mt.emplace_back([=] {
#ifdef SS_HOST_X86
                if (simd) return merge_bitmaps_avx512(bm, bm_words, n_splits, g0, g1, n, bits);
#endif
                merge_bitmaps_base(bm, bm_words, n_splits, g0, g1, n, bits);
            });
        }
        for (auto &th : mt) th.join();
    }
    g_words.put(reinterpret_cast<uint32_t *>(bm), 2 * bm_words * (uint64_t)n_splits);
    if (trace)
        fprintf(stderr, "[shuffle-split] n = %llu: stream walked at %.1f ms (%.1f ms of it waiting for swap threads, %u in flight, %s), workers done at %.1f ms, all at %.1f ms\n",
                (unsigned long long)n, t_gen * 1e3, t_wait * 1e3, in_flight, simd ? "64 words per step" : "word by word", t_workers * 1e3,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() * 1e3);
    if (trace && n >= 1000000) {
        fprintf(stderr, "[shuffle-split] workers (begin + duration, ms):");
        for (auto &w : t_worker) fprintf(stderr, " %.0f+%.0f", w.first * 1e3, (w.second - w.first) * 1e3);
        fprintf(stderr, "\n");
    }
    return err;
}

// ---------------------------------------------------------------------------------------------------------------------------
// ShuffleSplit with the swaps on the device (round 5).  The host walks the word stream -- that chain is the specification's:
// one sequential MT19937 stream through 20 permutations -- and writes down, for each split, the partners j of the rows
// i = n - 1 .. n_test (the test SET is complete there, see split_worker).  The swaps themselves, a dependent chain through a
// 20 MB array per split that took a host core 5-10 ms per split (20 splits: the host's cores were what four clusters solved
// at once waited for), are not replayed at all.  Step k (row i_k = n - 1 - k, partner j_k) moves into place i_k, for good, the
// element that lay at j_k: the one the most recent earlier step t < k with j_t = j_k left there -- which is what lay at ITS
// row i_t just before, and so on back -- or j_k itself when no step touched the place before.  "Most recent earlier step with
// the same partner" is the neighbour after a stable sort of the steps by partner; a place is the partner of ln 2 steps on
// average, so the walk back is a step or two.  Per split: one radix sort of n / 2 (partner, step) pairs, three small kernels,
// one atomicOr per row of the training half.  Output: bit f of train[e] = row e is in the TRAINING half of split f.
// ---------------------------------------------------------------------------------------------------------------------------
extern "C++" {
namespace {
constexpr uint32_t SD_NONE = 0xFFFFFFFFu;
constexpr int SD_NB = 3;                      // partner buffers in flight between the walker and the device

// link[k] = the step that had partner j[k] last before this kernel's thread k came by (any order): a list per place
__global__ __launch_bounds__(256) void sd_link_kernel(const uint32_t *__restrict__ j, uint32_t m, uint32_t *__restrict__ head, uint32_t *__restrict__ link)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < m) link[k] = atomicExch(&head[j[k]], k);
}
__global__ __launch_bounds__(256) void sd_clear_kernel(const uint32_t *__restrict__ j, uint32_t m, uint32_t *__restrict__ head)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < m) head[j[k]] = SD_NONE;
}
// the most recent step before `before` among the steps whose partner is place p (its list: a member or two)
__device__ __forceinline__ uint32_t sd_latest_before(const uint32_t *__restrict__ head, const uint32_t *__restrict__ link, uint32_t p, uint32_t before)
{
    uint32_t best = SD_NONE;
    for (uint32_t c = head[p]; c != SD_NONE; c = link[c])
        if (c < before && (best == SD_NONE || c > best)) best = c;
    return best;
}
__global__ __launch_bounds__(256) void sd_resolve_kernel(const uint32_t *__restrict__ j, const uint32_t *__restrict__ head, const uint32_t *__restrict__ link,
                                                         uint32_t m, uint32_t n, uint32_t bit, uint32_t *__restrict__ train)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= m) return;
    uint32_t e = j[k];
    // what lay at j[k] when step k took it: what the most recent earlier step with that partner left there -- the old content of ITS
    // row, that is what the most recent step before IT with that row as partner left there -- ... -- or the place's own number
    for (uint32_t t = sd_latest_before(head, link, e, k); t != SD_NONE;) {
        e = n - 1u - t;
        t = sd_latest_before(head, link, e, t);
    }
    atomicOr(&train[e], bit);
}
}  // namespace

struct ss_split {
    uint64_t n = 0, n_test = 0, m = 0;
    int n_splits = 0;
    uint32_t seed = 0;
    std::thread th;
    std::atomic<int> rc{SS_OK};
    std::atomic<bool> cancel{false};      // ss_split_dev_free before the walk is over (the pre-scan found one strain: no regression)
    hipStream_t stream = nullptr;
    uint32_t *d_train = nullptr, *d_j[SD_NB] = {nullptr, nullptr, nullptr}, *h_j[SD_NB] = {nullptr, nullptr, nullptr};
    hipEvent_t copied[SD_NB] = {nullptr, nullptr, nullptr};
    uint32_t *d_head = nullptr, *d_link = nullptr;
    double walk_ms = 0, total_ms = 0;
    int device = 0;
    std::thread::id user;                 // the thread that took the result (ss_split_dev_wait with a pointer): its stream reads d_train
    bool used = false;
};

namespace {
// pinned partner buffers are kept between calls (hipHostMalloc of 10 MB takes a millisecond; a solve uses three)
struct PinnedPool {
    std::mutex mu;
    std::vector<std::pair<uint32_t *, uint64_t>> free_;
    uint32_t *get(uint64_t words)
    {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].second >= words) { uint32_t *p = free_[i].first; free_.erase(free_.begin() + (long)i); return p; }
        }
        void *p = nullptr;
        return hipHostMalloc(&p, words * 4, hipHostMallocDefault) == hipSuccess ? (uint32_t *)p : nullptr;
    }
    void put(uint32_t *p, uint64_t words)
    {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(mu);
            uint64_t held = 0;
            for (auto &f : free_) held += f.second;
            if (free_.size() < 12 && (held + words) * 4 <= (256ull << 20)) { free_.emplace_back(p, words); return; }
        }
        hipHostFree(p);
    }
};
PinnedPool g_pinned;

void split_dev_free_buffers(ss_split *s)
{
    const uint64_t cap = s->m + SPLIT_CH + MTWords::BLK + 64;
    for (int b = 0; b < SD_NB; b++) {
        g_pinned.put(s->h_j[b], cap);
        s->h_j[b] = nullptr;
        if (s->d_j[b]) hipFreeAsync(s->d_j[b], s->stream);      // (stream-ordered, from the pool: no device-wide synchronisation beside
        s->d_j[b] = nullptr;                                    //  the other clusters being solved)
        if (s->copied[b]) hipEventDestroy(s->copied[b]);
        s->copied[b] = nullptr;
    }
    if (s->d_head) hipFreeAsync(s->d_head, s->stream);
    if (s->d_link) hipFreeAsync(s->d_link, s->stream);
    s->d_head = s->d_link = nullptr;
}

// Two host threads per call (round 5, last step): the WALKER only moves the word stream past every split -- the chain that is
// sequential by specification, 0.9 ms per split of 5 M rows -- and leaves a snapshot of the generator at each split's first word;
// the FEEDER takes the snapshots in order, walks the rows n - 1 .. n_test of each split again WITH the partners written into a
// pinned buffer (0.6 ms), and hands them to the device.  One thread doing both took 1.1 ms per split.
template <bool SIMD>
void split_dev_walk(ss_split *s)
{
    const auto t_begin = std::chrono::steady_clock::now();
    if (hipSetDevice(s->device) != hipSuccess) { s->rc = SS_EHIP; return; }
    std::unique_ptr<MTWords> rng(new (std::nothrow) MTWords(s->seed, SIMD));
    if (!rng) { s->rc = SS_ENOMEM; return; }
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::unique_ptr<MTWords>> snaps((size_t)s->n_splits);
    int produced = 0;                                     // snapshots [0, produced) are there
    bool walker_done = false;
    std::thread feeder([&] {
        if (hipSetDevice(s->device) != hipSuccess) { s->rc = SS_EHIP; return; }
        const uint32_t m = (uint32_t)s->m, n = (uint32_t)s->n;
        const unsigned grid = (m + 255u) / 256u;
        bool used[SD_NB] = {false, false, false};
        for (int f = 0; f < s->n_splits && s->rc == SS_OK && !s->cancel; f++) {
            std::unique_ptr<MTWords> snap;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return produced > f || walker_done; });
                if (produced <= f) break;                 // (the walker gave up)
                snap = std::move(snaps[(size_t)f]);
            }
            const int b = f % SD_NB;
            if (used[b] && hipEventSynchronize(s->copied[b]) != hipSuccess) { s->rc = SS_EHIP; break; }      // the copy that read this pinned buffer three splits ago
            uint64_t got = 0;
            {
                CoreGuard core;
                walk_split<true, SIMD, true>(*snap, s->n, s->h_j[b], [&](const uint32_t *, uint32_t cnt) { got = cnt; }, s->n_test);
            }
            if (got != s->m) { s->rc = SS_ERANGE; break; }
            hipError_t e = hipMemcpyAsync(s->d_j[b], s->h_j[b], (uint64_t)m * 4, hipMemcpyHostToDevice, s->stream);
            if (e == hipSuccess) e = hipEventRecord(s->copied[b], s->stream);
            used[b] = true;
            if (e == hipSuccess) {
                hipLaunchKernelGGL(sd_link_kernel, dim3(grid), dim3(256), 0, s->stream, s->d_j[b], m, s->d_head, s->d_link);
                hipLaunchKernelGGL(sd_resolve_kernel, dim3(grid), dim3(256), 0, s->stream, s->d_j[b], s->d_head, s->d_link, m, n, 1u << f, s->d_train);
                hipLaunchKernelGGL(sd_clear_kernel, dim3(grid), dim3(256), 0, s->stream, s->d_j[b], m, s->d_head);
                e = hipGetLastError();
            }
            if (e != hipSuccess) { ss::set_last_error("ss_split_dev", __FILE__, __LINE__, e); s->rc = SS_EHIP; break; }
        }
    });
    {
        CoreGuard core;
        for (int f = 0; f < s->n_splits && s->rc == SS_OK && !s->cancel; f++) {
            std::unique_ptr<MTWords> snap(new (std::nothrow) MTWords(*rng));
            if (!snap) { s->rc = SS_ENOMEM; break; }
            {
                std::lock_guard<std::mutex> g(mu);
                snaps[(size_t)f] = std::move(snap);
                produced = f + 1;
            }
            cv.notify_all();
            // past the split's words: only to know where the next split's begin
            if (f + 1 < s->n_splits) walk_split<false, SIMD>(*rng, s->n, nullptr, [](const uint32_t *, uint32_t) {});
        }
    }
    { std::lock_guard<std::mutex> g(mu); walker_done = true; }
    cv.notify_all();
    s->walk_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() * 1e3;
    feeder.join();
}
}  // namespace
}  // extern "C++"

int ss_split_dev_start(uint64_t n, int n_splits, uint64_t n_test, uint32_t seed, ss_split **out)
{
    if (!out || n_splits < 1 || n_splits > 31 || n_test < 1 || n_test >= n || n > 0x7FFFFFFFull) return SS_EINVAL;
    ss_split *s = new (std::nothrow) ss_split();
    if (!s) return SS_ENOMEM;
    s->n = n; s->n_test = n_test; s->m = n - n_test; s->n_splits = n_splits; s->seed = seed;
    hipGetDevice(&s->device);
    const uint64_t cap = s->m + SPLIT_CH + MTWords::BLK + 64;
    // the LOWEST stream priority: the splits are needed only when the pre-scan is over, and the pre-scan's many small kernels
    // (and those of the other clusters being solved) should not queue behind 2.5 M-thread resolve launches
    int pr_least = 0, pr_greatest = 0;
    hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
    bool ok = hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, pr_least) == hipSuccess;
    for (int b = 0; b < SD_NB && ok; b++) {
        s->h_j[b] = g_pinned.get(cap);
        ok = s->h_j[b] && hipMallocAsync((void **)&s->d_j[b], s->m * 4, s->stream) == hipSuccess &&
             hipEventCreateWithFlags(&s->copied[b], hipEventDisableTiming) == hipSuccess;
    }
    if (ok) ss::pool_keep_at_least(2ull << 30);
    ok = ok && hipMallocAsync((void **)&s->d_train, n * 4, s->stream) == hipSuccess && hipMallocAsync((void **)&s->d_head, n * 4, s->stream) == hipSuccess &&
         hipMallocAsync((void **)&s->d_link, s->m * 4, s->stream) == hipSuccess;
    if (ok) ok = hipMemsetAsync(s->d_train, 0, n * 4, s->stream) == hipSuccess && hipMemsetAsync(s->d_head, 0xFF, n * 4, s->stream) == hipSuccess;
    if (!ok) {
        if (s->stream) {
            split_dev_free_buffers(s);
            if (s->d_train) hipFreeAsync(s->d_train, s->stream);
            hipStreamSynchronize(s->stream);
            hipStreamDestroy(s->stream);
        }
        delete s;
        return SS_ENOMEM;
    }
#ifdef SS_HOST_X86
    const bool simd = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("popcnt") && host_simd_allowed();
#else
    const bool simd = false;
#endif
    s->th = std::thread([=] { if (simd) split_dev_walk<true>(s); else split_dev_walk<false>(s); });
    *out = s;
    return SS_OK;
}

int ss_split_dev_wait(ss_split *s, const uint32_t **train_bits_dev, double *walk_ms)
{
    if (!s) return SS_EINVAL;
    if (s->th.joinable()) s->th.join();
    if (s->stream && hipStreamSynchronize(s->stream) != hipSuccess && s->rc == SS_OK) s->rc = SS_EHIP;
    split_dev_free_buffers(s);                          // only the result stays
    if (train_bits_dev) { *train_bits_dev = s->d_train; s->user = std::this_thread::get_id(); s->used = true; }      // (the thread whose stream will read the bits)
    if (walk_ms) *walk_ms = s->walk_ms;
    return s->rc;
}

int ss_split_dev_free(ss_split *s)
{
    if (!s) return SS_OK;
    s->cancel = true;
    ss_split_dev_wait(s, nullptr, nullptr);
    // The training bits were last read on the per-thread stream of the thread that asked for them (ss_l2_fold_train): freed in
    // that stream's order when this IS that thread; from any other thread (a garbage collector's) the per-thread stream is a
    // different one and orders nothing, so the free waits for the device instead.
    if (s->d_train) {
        if (!s->used || s->user == std::this_thread::get_id()) hipFreeAsync(s->d_train, hipStreamPerThread);
        else hipFree(s->d_train);
    }
    if (s->stream) { hipStreamSynchronize(s->stream); hipStreamDestroy(s->stream); }
    delete s;
    return SS_OK;
}

int ss_revcomp(const char *in, char *out, uint64_t n)
{
    if (n && (!in || !out)) return SS_EINVAL;
    for (uint64_t i = 0; i < n; i++) out[n - 1 - i] = (char)comp_base((unsigned char)in[i]);
    return SS_OK;
}

int ss_revcomp_dev(const char *in_dev, char *out_dev, uint64_t seq_len, uint64_t n_seq, void *stream)
{
    if (!seq_len || !n_seq) return SS_OK;
    if (!in_dev || !out_dev) return SS_EINVAL;
    const uint64_t total = seq_len * n_seq;
    const unsigned blocks = (unsigned)std::min<uint64_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(revcomp_kernel, dim3(blocks), dim3(256), 0, ss::as_stream(stream), in_dev, out_dev, seq_len,
                       n_seq);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int ss_encode_kmer(const char *kmer, int k, uint64_t *key)
{
    if (!kmer || !key || k < 1 || k > 31) return SS_EINVAL;
    uint64_t v = 0;
    for (int j = 0; j < k; j++) {
        int code = ss::base_code((unsigned char)kmer[j]);
        if (code < 0) return SS_EINVAL;
        v |= (uint64_t)code << (2 * j);
    }
    *key = v;
    return SS_OK;
}

int ss_host_cpus(void) { return (int)ss::host_cpus(); }

}  // extern "C"

namespace {

// A plain (not gzip) file mapped read-only; ok() is false for gzip input or when it cannot be mapped
// (the callers then take the gzread path).
struct PlainMap {
    const char *p = nullptr;
    uint64_t n = 0;
    bool mapped = false;
    explicit PlainMap(const char *path)
    {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 2) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                p = (const char *)m;
                n = (uint64_t)st.st_size;
                mapped = true;
                if ((unsigned char)p[0] == 0x1f && (unsigned char)p[1] == 0x8b) release();
            }
        }
        close(fd);
    }
    void release() { if (mapped) munmap((void *)p, n); mapped = false; p = nullptr; n = 0; }
    ~PlainMap() { release(); }
    bool ok() const { return mapped; }
};

unsigned host_threads(uint64_t bytes)
{
    if (const char *e = getenv("SS_HOST_THREADS")) return (unsigned)std::max(1, atoi(e));   // tests: many chunks of a small file
    if (bytes < (8u << 20)) return 1;
    return std::min<unsigned>(ss::host_cpus(), 32u);
}

// '\n' count of [a, b) per chunk of a text cut into `T` equal byte ranges
std::vector<uint64_t> newline_counts(const char *t, uint64_t n, unsigned T)
{
    std::vector<uint64_t> cnt(T, 0);
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < T; w++)
        pool.emplace_back([&, w] {
            const uint64_t a = n * w / T, b = n * (w + 1) / T;
            uint64_t c = 0;
            for (uint64_t i = a; i < b; i++) c += (t[i] == '\n');
            cnt[w] = c;
        });
    for (auto &th : pool) th.join();
    return cnt;
}

// rows (the odd lines) of a mapped k-mer FASTA encoded by T threads; returns the number of rows
uint64_t encode_mapped(const char *t, uint64_t n, int k, uint64_t n_rows, uint64_t *keys, uint8_t *flags)
{
    const unsigned T = host_threads(n);
    const std::vector<uint64_t> cnt = newline_counts(t, n, T);
    std::vector<uint64_t> before(T + 1, 0);                    // '\n's in front of chunk w = index of the line cut by its start
    for (unsigned w = 0; w < T; w++) before[w + 1] = before[w] + cnt[w];
    const uint64_t lines = before[T] + ((n && t[n - 1] != '\n') ? 1 : 0);
    if (lines / 2 != n_rows) return lines / 2;
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < T; w++)
        pool.emplace_back([&, w] {
            const uint64_t a = n * w / T, b = n * (w + 1) / T;
            // first line that STARTS in [a, b)
            uint64_t pos = a, line = before[w];
            if (a > 0 && t[a - 1] != '\n') {
                const void *nl = memchr(t + a, '\n', n - a);
                if (!nl) return;
                pos = (uint64_t)((const char *)nl - t) + 1;
                line++;
            }
            for (; pos < b && pos < n; line++) {
                const void *nl = memchr(t + pos, '\n', n - pos);
                uint64_t e = nl ? (uint64_t)((const char *)nl - t) : n;
                const uint64_t next = nl ? e + 1 : n;
                if (line & 1) {
                    while (e > pos && (t[e - 1] == ' ' || t[e - 1] == '\r' || t[e - 1] == '\t' || t[e - 1] == '\v' || t[e - 1] == '\f')) e--;
                    const RowSpan r{pos, e};
                    if (line / 2 < n_rows) encode_rows(t, &r, 1, k, keys + line / 2, flags + line / 2);
                }
                pos = next;
            }
        });
    for (auto &th : pool) th.join();
    return n_rows;
}

}  // namespace

extern "C" {

// <Tree_database>/kmers/<id>: ONE line of space separated row numbers of kmer.fa (Build_tree.py:686-698; identify.py:116-118
// reads `f.readline().split()`).  1645 such files hold 25 M tokens for an E. coli tree: 0.7 s of numpy, 40 ms here on the
// host's threads.  Two calls: counts[i] = tokens of file i; then rows (offsets[i] = where file i's tokens go).  Anything
// that is not a decimal number below n_rows: SS_EINVAL and *bad = the file's index (the caller parses that file itself
// and raises what numpy raises).
static int node_file_tokens(const char *dir, long long id, uint64_t n_rows, uint64_t *count, uint32_t *out)
{
    char path[4096];
    if (snprintf(path, sizeof path, "%s/%lld", dir, id) >= (int)sizeof path) return SS_EINVAL;
    std::string t;
    if (!read_whole_file(path, t)) return SS_EIO;
    uint64_t n = 0;
    const size_t end = std::min(t.size(), t.find('\n'));          // the first line only
    for (size_t i = 0; i < end;) {
        const unsigned char c = (unsigned char)t[i];
        if (c == ' ' || c == '\t' || c == '\r' || c == '\f' || c == '\v') { i++; continue; }
        uint64_t v = 0;
        size_t j = i;
        for (; j < end && t[j] >= '0' && t[j] <= '9' && j - i < 12; j++) v = v * 10 + (uint64_t)(t[j] - '0');
        if (j == i || (j < end && !(t[j] == ' ' || t[j] == '\t' || t[j] == '\r' || t[j] == '\f' || t[j] == '\v')) || v >= std::max<uint64_t>(1, n_rows))
            return SS_EINVAL;
        if (out) out[n] = (uint32_t)v;
        n++;
        i = j;
    }
    *count = n;
    return SS_OK;
}

int ss_node_lists_parse(const char *kmers_dir, const long long *ids, uint32_t n_ids, uint64_t n_rows, uint64_t *counts,
                        const uint64_t *offsets, uint32_t *rows, uint32_t *bad)
{
    if (!kmers_dir || (n_ids && (!ids || !counts)) || (rows && !offsets)) return SS_EINVAL;
    std::atomic<uint32_t> next(0);
    std::atomic<int> err(SS_OK);
    std::atomic<uint32_t> first_bad(0xFFFFFFFFu);
    const unsigned nt = std::max(1u, std::min(32u, ss::host_cpus()));
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < nt; w++)
        pool.emplace_back([&] {
            for (uint32_t i; (i = next.fetch_add(1)) < n_ids;) {
                uint64_t c = 0;
                const int rc = node_file_tokens(kmers_dir, ids[i], n_rows, &c, rows ? rows + offsets[i] : nullptr);
                if (rc != SS_OK) {
                    err = rc;
                    uint32_t cur = first_bad.load();
                    while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
                } else if (rows && c != counts[i]) {
                    err = SS_EIO;                                   // the file changed between the two calls
                } else {
                    counts[i] = c;
                }
            }
        });
    for (auto &th : pool) th.join();
    if (bad) *bad = first_bad.load();
    return err;
}

int ss_kmerfa_count_rows(const char *path, uint64_t *n_rows)
{
    if (!path || !n_rows) return SS_EINVAL;
    {
        PlainMap m(path);
        if (m.ok()) {
            uint64_t nl = 0;
            for (uint64_t c : newline_counts(m.p, m.n, host_threads(m.n))) nl += c;
            if (m.n && m.p[m.n - 1] != '\n') nl++;
            *n_rows = nl / 2;
            return SS_OK;
        }
    }
    std::string t;
    if (!read_whole_file(path, t)) return SS_EIO;
    uint64_t nl = 0;
    for (char c : t) nl += (c == '\n');
    if (!t.empty() && t.back() != '\n') nl++;
    *n_rows = nl / 2;
    return SS_OK;
}

int ss_kmerfa_encode_mem(const char *text, uint64_t len, int k, uint64_t n_rows, uint64_t *keys, uint8_t *flags)
{
    if ((len && !text) || (n_rows && (!keys || !flags))) return SS_EINVAL;
    if (k < 1 || k > 31) return SS_ERANGE;
    std::vector<RowSpan> rows;
    split_rows(text, len, rows);
    if (rows.size() != n_rows) return SS_EINVAL;
    encode_rows(text, rows.data(), n_rows, k, keys, flags);
    return SS_OK;
}

int ss_kmerfa_encode(const char *path, int k, uint64_t n_rows, uint64_t *keys, uint8_t *flags, int threads)
{
    if (!path || (n_rows && (!keys || !flags))) return SS_EINVAL;
    if (k < 1 || k > 31) return SS_ERANGE;
    if (threads <= 0) {                         // plain file: mapped, cut at arbitrary offsets, every thread finds its lines
        PlainMap m(path);
        if (m.ok()) return encode_mapped(m.p, m.n, k, n_rows, keys, flags) == n_rows ? SS_OK : SS_EINVAL;
    }
    std::string t;
    if (!read_whole_file(path, t)) return SS_EIO;
    std::vector<RowSpan> rows;
    split_rows(t.data(), t.size(), rows);
    if (rows.size() != n_rows) return SS_EINVAL;
    if (threads <= 0) threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    if (n_rows < 100000) threads = 1;
    std::vector<std::thread> pool;
    const uint64_t per = (n_rows + threads - 1) / threads;
    for (int w = 0; w < threads; w++) {
        const uint64_t lo = std::min<uint64_t>(n_rows, per * w), hi = std::min<uint64_t>(n_rows, lo + per);
        if (lo >= hi) break;
        pool.emplace_back([&, lo, hi] { encode_rows(t.data(), rows.data() + lo, hi - lo, k, keys + lo, flags + lo); });
    }
    for (auto &th : pool) th.join();
    return SS_OK;
}

int ss_reader_open(const char *const *paths, int n_paths, ss_reader **out)
{
    if (!paths || n_paths < 1 || !out) return SS_EINVAL;
    ss_reader *r = new (std::nothrow) ss_reader();
    if (!r) return SS_ENOMEM;
    for (int i = 0; i < n_paths; i++) {
        if (!paths[i]) { delete r; return SS_EINVAL; }
        if (paths[i][0]) r->paths.emplace_back(paths[i]);  // '' = no second file (StrainScan.py:182)
    }
    r->in.resize(4 << 20);
    for (const auto &p : r->paths) {  // fail early on unreadable inputs
        gzFile f = gzopen(p.c_str(), "rb");
        if (!f) { delete r; return SS_EIO; }
        gzclose(f);
    }
    *out = r;
    return SS_OK;
}

int ss_reader_set_overlap(ss_reader *r, int overlap)
{
    if (!r || overlap < 0 || overlap > 62) return SS_EINVAL;
    r->overlap = overlap;
    return SS_OK;
}

int ss_reader_close(ss_reader *r)
{
    if (!r) return SS_OK;
    if (r->f) gzclose(r->f);
    delete r;
    return SS_OK;
}

int ss_reader_next(ss_reader *r, char *out, uint64_t cap, uint64_t *out_len, uint64_t *n_records)
{
    if (!r || !out || !out_len || cap < 4096) return SS_EINVAL;
    uint64_t o = 0, recs = 0;
    const uint64_t ov = (uint64_t)r->overlap;
    auto in_seq = [&] { return r->st == ss_reader::FA_SEQ_BOL || r->st == ss_reader::FA_SEQ ||
                               r->st == ss_reader::FQ_SEQ_BOL || r->st == ss_reader::FQ_SEQ; };
    if (r->need_tail) {  // continuation of a record cut at the previous buffer end
        memcpy(out, r->tail.data(), r->tail.size());
        o = r->tail.size();
        r->need_tail = false;
    }
    auto end_record = [&] {
        out[o++] = '\n';
        recs++;
        r->tail.clear();
    };
    for (;;) {
        if (r->in_pos >= r->in_len) {
            if (!r->f) {
                // finish a record that the previous file left open, then open the next file
                if (r->st != ss_reader::START) {
                    if (in_seq()) end_record();
                    r->st = ss_reader::START;
                }
                if (r->file_idx >= r->paths.size()) break;
                r->f = gzopen(r->paths[r->file_idx++].c_str(), "rb");
                if (!r->f) return SS_EIO;
                gzbuffer(r->f, 1 << 20);
            }
            if (!reader_fill(r)) continue;  // file ended: loop closes the record / opens the next
        }
        // leave room for the longest thing one step can append: a line chunk is bounded below
        if (o + 2 >= cap) {
            if (in_seq()) {  // cut inside a sequence: remember the last `overlap` bases
                uint64_t s = o;
                while (s > 0 && out[s - 1] != '\n' && o - s < ov) s--;
                r->tail.assign(out + s, out + o);
                r->need_tail = true;
            }
            break;
        }
        const char *p = r->in.data() + r->in_pos;
        const size_t avail = r->in_len - r->in_pos;
        switch (r->st) {
        case ss_reader::START: {
            const char c = *p;
            if (c == '\n') { r->in_pos++; }
            else if (c == '>') { r->st = ss_reader::FA_HDR; }
            else if (c == '@') { r->st = ss_reader::FQ_HDR; r->seqlen = 0; r->qlen = 0; }
            else { r->st = ss_reader::FQ_PLUS; r->seqlen = 0; r->qlen = 0; }  // stray line: skip it
            break;
        }
        case ss_reader::FA_HDR:
        case ss_reader::FQ_HDR:
        case ss_reader::FQ_PLUS: {
            const void *nl = memchr(p, '\n', avail);
            if (!nl) { r->in_pos = r->in_len; break; }
            r->in_pos += (size_t)((const char *)nl - p) + 1;
            r->st = (r->st == ss_reader::FA_HDR)   ? ss_reader::FA_SEQ_BOL
                    : (r->st == ss_reader::FQ_HDR) ? ss_reader::FQ_SEQ_BOL
                    : (r->qlen < r->seqlen)        ? ss_reader::FQ_QUAL
                                                   : ss_reader::START;
            break;
        }
        case ss_reader::FA_SEQ_BOL:
            if (*p == '>') { end_record(); r->st = ss_reader::FA_HDR; }
            else r->st = ss_reader::FA_SEQ;
            break;
        case ss_reader::FQ_SEQ_BOL:
            if (*p == '+') { end_record(); r->st = ss_reader::FQ_PLUS; }
            else r->st = ss_reader::FQ_SEQ;
            break;
        case ss_reader::FA_SEQ:
        case ss_reader::FQ_SEQ: {
            const void *nl = memchr(p, '\n', avail);
            size_t n = nl ? (size_t)((const char *)nl - p) : avail;
            const uint64_t room = cap - 2 - o;
            bool whole = true;
            if (n > room) { n = (size_t)room; whole = false; }
            memcpy(out + o, p, n);
            o += n;
            r->seqlen += n;
            r->in_pos += n;
            if (whole && nl) {
                r->in_pos++;
                r->st = (r->st == ss_reader::FA_SEQ) ? ss_reader::FA_SEQ_BOL : ss_reader::FQ_SEQ_BOL;
            }
            break;
        }
        case ss_reader::FQ_QUAL: {
            const void *nl = memchr(p, '\n', avail);
            size_t n = nl ? (size_t)((const char *)nl - p) : avail;
            r->qlen += n;
            r->in_pos += n + (nl ? 1 : 0);
            if (nl && r->qlen >= r->seqlen) r->st = ss_reader::START;
            break;
        }
        }
    }
    *out_len = o;
    if (n_records) *n_records = recs;
    return SS_OK;
}

int ss_fastx_to_flat(const char *text, uint64_t len, char *out, uint64_t *out_len, uint64_t *n_records)
{
    if ((len && !text) || !out || !out_len) return SS_EINVAL;
    // same grammar as ss_reader_next, on an in-memory text (single pass, no cuts)
    uint64_t i = 0, o = 0, recs = 0;
    const char *t = text;
    while (i < len) {
        if (t[i] == '\n') { i++; continue; }
        if (t[i] == '>') {
            const void *nl = memchr(t + i, '\n', len - i);
            i = nl ? (uint64_t)((const char *)nl - t) + 1 : len;
            while (i < len && t[i] != '>') {
                nl = memchr(t + i, '\n', len - i);
                uint64_t e = nl ? (uint64_t)((const char *)nl - t) : len;
                memcpy(out + o, t + i, e - i);
                o += e - i;
                i = nl ? e + 1 : len;
            }
            out[o++] = '\n';
            recs++;
        } else if (t[i] == '@') {
            uint64_t seqlen = 0, qlen = 0;
            const void *nl = memchr(t + i, '\n', len - i);
            i = nl ? (uint64_t)((const char *)nl - t) + 1 : len;
            while (i < len && t[i] != '+') {
                nl = memchr(t + i, '\n', len - i);
                uint64_t e = nl ? (uint64_t)((const char *)nl - t) : len;
                memcpy(out + o, t + i, e - i);
                o += e - i;
                seqlen += e - i;
                i = nl ? e + 1 : len;
            }
            out[o++] = '\n';
            recs++;
            nl = (i < len) ? memchr(t + i, '\n', len - i) : nullptr;
            i = nl ? (uint64_t)((const char *)nl - t) + 1 : len;
            while (i < len && qlen < seqlen) {
                nl = memchr(t + i, '\n', len - i);
                uint64_t e = nl ? (uint64_t)((const char *)nl - t) : len;
                qlen += e - i;
                i = nl ? e + 1 : len;
            }
        } else {
            const void *nl = memchr(t + i, '\n', len - i);
            i = nl ? (uint64_t)((const char *)nl - t) + 1 : len;
        }
    }
    *out_len = o;
    if (n_records) *n_records = recs;
    return SS_OK;
}

}  // extern "C"
