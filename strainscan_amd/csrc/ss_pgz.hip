// ss_pgz.hip -- a single gzip member inflated by many threads (host code only).
//
// The reference pipes `zcat` into jellyfish (library/identify.py:81-84): one inflate thread per file,
// which is also what zlib and libdeflate give (a deflate stream has no entry points: every block may
// copy from the 32 KB before it).  This is the two-pass scheme of Kerbiriou & Chikhi ("Parallel
// decompression of gzip-compressed files and random access to DNA sequences", 2019), restated:
//   A  every thread takes a byte range of the compressed stream and searches it, bit by bit, for the
//      first position that parses as the header of a dynamic-Huffman block (complete code-length
//      code, valid run lengths, complete literal/length code with an end-of-block symbol) and from
//      which a few blocks decode without an error;
//   B  every thread inflates from its position to the position of the next thread, not knowing the
//      32 KB in front of it: the output is 16-bit symbols, a byte or "byte w of the unknown window"
//      (a copy of a copy keeps the window index: one lookup resolves any symbol);
//   C  in stream order, the last 32 KB of every chunk are resolved and become the window of the next;
//   D  every thread turns its symbols into bytes at the chunk's place in the text.
// The result is accepted only if every chunk stopped exactly where the next one began, the stream
// ended where the 8-byte gzip trailer begins, and CRC-32 and ISIZE of the trailer match; otherwise
// the caller inflates the file with libdeflate or zlib.  Stored and fixed-Huffman blocks are decoded
// but never used as entry points (their headers are too easy to mistake).
// No HIP in this file: tests/pgz_fuzz.cpp compiles it with g++ -fsanitize=address,undefined as well.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t WSIZE = 32768;
constexpr uint16_t UNRESOLVED = 0x8000;     // symbol = UNRESOLVED | index into the 32 KB in front of the chunk

struct Bits {
    const uint8_t *in;
    uint64_t n;          // bytes
    uint64_t pos;        // next byte to load
    uint64_t buf = 0;
    int cnt = 0;
    bool over = false;   // read past the end

    Bits(const uint8_t *p, uint64_t len, uint64_t bitpos) : in(p), n(len), pos(bitpos >> 3)
    {
        const int skip = (int)(bitpos & 7);
        if (skip) { need(8); buf >>= skip; cnt -= skip; }
    }
    inline void need(int k)      // k <= 32
    {
        while (cnt < k) {
            if (pos + 4 <= n && cnt <= 32) {
                uint32_t w;
                memcpy(&w, in + pos, 4);
                buf |= (uint64_t)w << cnt;
                pos += 4;
                cnt += 32;
            } else if (pos < n) {
                buf |= (uint64_t)in[pos++] << cnt;
                cnt += 8;
            } else {
                over = true;     // zeros
                cnt += 8;
            }
        }
    }
    // at least 56 bits in the buffer (one unaligned 8-byte load) unless the input is about to end
    inline void refill()
    {
        if (pos + 8 <= n) {
            uint64_t w;
            memcpy(&w, in + pos, 8);
            buf |= w << cnt;
            const int adv = (63 - cnt) >> 3;
            pos += (uint64_t)adv;
            cnt += adv * 8;
        } else {
            need(32);
        }
    }
    inline uint32_t peek(int k) { need(k); return (uint32_t)(buf & ((1ull << k) - 1)); }
    inline void drop(int k) { buf >>= k; cnt -= k; }
    inline uint32_t get(int k) { const uint32_t v = peek(k); drop(k); return v; }
    uint64_t bitpos() const { return pos * 8 - (uint64_t)cnt; }
    void align() { drop(cnt & 7); }
};

// canonical Huffman code, decoded with a PB-bit table and a bit-by-bit walk for longer codes
template <int PB, int MAXSYM>
struct Huff {
    uint16_t tsym[1 << PB];
    uint8_t tlen[1 << PB];
    uint16_t count[16], sorted[MAXSYM];
    int maxlen = 0;

    // 0 ok, 1 incomplete (left > 0), -1 over-subscribed / unusable
    int build(const uint8_t *lens, int n)
    {
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++) count[lens[i]]++;
        maxlen = 15;
        while (maxlen > 0 && count[maxlen] == 0) maxlen--;
        int left = 1;
        for (int l = 1; l <= 15; l++) {
            left <<= 1;
            left -= count[l];
            if (left < 0) return -1;
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; i++) if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
        memset(tlen, 0, sizeof(tlen));
        // table: every PB-bit pattern whose low bits are the bit-reversed code
        uint32_t code = 0;
        int idx = 0;
        for (int l = 1; l <= 15; l++) {
            for (int k = 0; k < count[l]; k++, idx++, code++) {
                if (l > PB) continue;
                uint32_t r = 0;
                for (int b = 0; b < l; b++) r |= ((code >> b) & 1u) << (l - 1 - b);
                for (uint32_t e = r; e < (1u << PB); e += 1u << l) { tsym[e] = sorted[idx]; tlen[e] = (uint8_t)l; }
            }
            code <<= 1;
        }
        return left > 0 ? 1 : 0;
    }
    // -1 on an invalid code
    inline int decode(Bits &b) const
    {
        const uint32_t v = b.peek(15);
        const int l = tlen[v & ((1u << PB) - 1)];
        if (l) { b.drop(l); return tsym[v & ((1u << PB) - 1)]; }
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++) {
            code |= (int)((v >> (len - 1)) & 1u);
            const int c = count[len];
            if (code - c < first) { b.drop(len); return sorted[index + (code - first)]; }
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

using LitCode = Huff<11, 288>;
using DistCode = Huff<8, 32>;
using ClCode = Huff<7, 19>;

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Codes {
    LitCode lit;
    DistCode dist;
    bool dist_usable = true;
};

// header of a dynamic block (the three header bits already consumed); false = not a valid header
bool read_dynamic(Bits &b, Codes &c)
{
    const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; i++) cl[CL_ORDER[i]] = (uint8_t)b.get(3);
    ClCode clc;
    if (clc.build(cl, 19) != 0) return false;                      // zlib: the code-length code must be complete
    uint8_t lens[286 + 30];
    int i = 0;
    while (i < hlit + hdist) {
        const int s = clc.decode(b);
        if (s < 0 || b.over) return false;
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        int rep, val = 0;
        if (s == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (int)b.get(2); }
        else if (s == 17) rep = 3 + (int)b.get(3);
        else rep = 11 + (int)b.get(7);
        if (i + rep > hlit + hdist) return false;
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (lens[256] == 0) return false;                                // no end-of-block code
    const int rl = c.lit.build(lens, hlit);
    if (rl < 0 || (rl > 0 && c.lit.maxlen != 1)) return false;       // zlib: incomplete only as a single 1-bit code
    const int rd = c.dist.build(lens + hlit, hdist);
    if (rd < 0 || (rd > 0 && c.dist.maxlen > 1)) return false;
    c.dist_usable = c.dist.maxlen > 0;
    return !b.over;
}

const Codes &fixed_codes()
{
    static Codes c;
    static std::once_flag once;
    std::call_once(once, [] {
        uint8_t l[288];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        c.lit.build(l, 288);
        uint8_t d[30];
        for (int i = 0; i < 30; i++) d[i] = 5;
        c.dist.build(d, 30);        // incomplete by two codes (30, 31 invalid): the table still decodes 0..29
    });
    return c;
}

struct alignas(128) Out {           // one per thread, written on every symbol: its own cache lines
    uint16_t *p = nullptr;
    uint64_t n = 0, cap = 0;
    bool count_only = false;        // the search only needs to know that decoding works
    ~Out() { if (p) munmap(p, cap * 2); }
    // anonymous mappings with transparent huge pages asked for: thirty-two threads touching 4 KB pages of
    // fresh memory for the first time spend their time in page faults, not in inflate
    bool room(uint64_t k)
    {
        if (count_only || n + k <= cap) return true;
        uint64_t ncap = std::max<uint64_t>(cap + cap / 2, n + k + (1u << 20));
        ncap = (ncap * 2 + (2u << 20) - 1) / (2u << 20) * (2u << 20) / 2;        // whole 2 MB pages
        void *q;
        if (!p) {
            q = mmap(nullptr, ncap * 2, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q == MAP_FAILED) return false;
        } else {
            q = mremap(p, cap * 2, ncap * 2, MREMAP_MAYMOVE);
            if (q == MAP_FAILED) return false;
        }
        madvise(q, ncap * 2, MADV_HUGEPAGE);
        p = (uint16_t *)q;
        cap = ncap;
        return true;
    }
};

// One block from the current position (BFINAL and BTYPE not yet read).  0 = block done, 1 = final block done,
// <0 = error.  known_window: references in front of the output are an error (chunk 0 and libdeflate agree).
int inflate_block(Bits &b, Out &o, bool known_window)
{
    const uint32_t bfinal = b.get(1), btype = b.get(2);
    if (btype == 3) return -1;
    if (btype == 0) {
        b.align();
        const uint32_t len = b.get(16), nlen = b.get(16);
        if (b.over || (len ^ 0xFFFFu) != nlen) return -2;
        if (!o.room(len)) return -9;
        for (uint32_t i = 0; i < len; i++) {
            const uint32_t v = b.get(8);
            if (!o.count_only) o.p[o.n] = (uint16_t)v;
            o.n++;
        }
        return b.over ? -3 : (int)bfinal;
    }
    Codes dyn;
    const Codes *c = &dyn;
    if (btype == 1) c = &fixed_codes();
    else if (!read_dynamic(b, dyn)) return -4;
    const LitCode &lit = c->lit;
    const DistCode &dc = c->dist;
    const bool counting = o.count_only;
    uint64_t n = o.n;                                  // kept in registers; written back on every exit
    uint16_t *out = o.p;
#define PGZ_RET(v) do { o.n = n; return (v); } while (0)
    for (;;) {
        if (!counting && n + 264 > o.cap) {
            o.n = n;
            if (!o.room(264)) return -9;
            out = o.p;
        }
        b.refill();
        int s = lit.decode(b);
        if (s >= 0 && s < 256) {                      // up to three literals per refill (3 x 15 bits <= 56)
            if (!counting) out[n] = (uint16_t)s;
            n++;
            s = lit.decode(b);
            if (s >= 0 && s < 256) {
                if (!counting) out[n] = (uint16_t)s;
                n++;
                s = lit.decode(b);
                if (s >= 0 && s < 256) {
                    if (!counting) out[n] = (uint16_t)s;
                    n++;
                    if (b.over) PGZ_RET(-5);
                    continue;
                }
            }
        }
        if (s < 0 || b.over) PGZ_RET(-5);
        if (s == 256) PGZ_RET((int)bfinal);
        if (s > 285) PGZ_RET(-6);
        b.refill();                                   // length extra 5 + distance 15 + distance extra 13 <= 33 bits
        const int li = s - 257;
        const uint32_t len = LEN_BASE[li] + b.get(LEN_EXTRA[li]);
        if (!c->dist_usable) PGZ_RET(-7);
        const int ds = dc.decode(b);
        if (ds < 0 || ds > 29) PGZ_RET(-7);
        const uint32_t dist = DIST_BASE[ds] + b.get(DIST_EXTRA[ds]);
        if (b.over) PGZ_RET(-5);
        if (counting) {
            if (known_window && dist > n) PGZ_RET(-8);
            n += len;
            continue;
        }
        uint16_t *dst = out + n;
        if (dist <= n) {
            const uint16_t *src = dst - dist;
            if (dist >= 8) {
                // blocks of 8 symbols (one 16-byte move each, no call): block k reads symbols that are final already
                // because dist > 7, also when the match overlaps itself; up to 7 symbols of overrun land in the
                // 264 symbols of slack and are overwritten by what follows
                for (uint32_t i = 0; i < len; i += 8) memcpy(dst + i, src + i, 16);
            } else {
                for (uint32_t i = 0; i < len; i++) dst[i] = src[i];          // short period: element by element
            }
        } else {
            if (known_window) PGZ_RET(-8);
            // the first (dist - n) elements come from the unknown window, the rest (if any) from the output
            const uint64_t from_w = std::min<uint64_t>(len, dist - n);
            const uint32_t w0 = WSIZE - (uint32_t)(dist - n);
            for (uint64_t i = 0; i < from_w; i++) dst[i] = (uint16_t)(UNRESOLVED | (w0 + i));
            for (uint64_t i = from_w; i < len; i++) dst[i] = out[n + i - dist];
        }
        n += len;
    }
#undef PGZ_RET
}

// gzip member header -> offset of the deflate data, 0 = not a header this code handles
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
// where the text goes: by default 2 MB aligned heap memory released with free(); ss_gz_inflate_to_file maps a file
struct TextAlloc {
    void *(*alloc)(uint64_t cap, void *ctx) = nullptr;
    void (*release)(void *p, uint64_t cap, void *ctx) = nullptr;
    void *ctx = nullptr;
};
bool parallel_gunzip(const uint8_t *in, uint64_t in_n, unsigned threads, uint64_t budget, char **text, uint64_t *len,
                     const TextAlloc *ta = nullptr);

// CRC-32 of a buffer, continuing `crc` (zlib's convention).  zlib's own does ~2 GB/s per thread; the ingest
// installs libdeflate's carry-less-multiply version (10+ GB/s) when that library is loaded.
static uint32_t (*g_crc32)(uint32_t, const void *, size_t) = nullptr;
void pgz_set_crc32(uint32_t (*fn)(uint32_t, const void *, size_t)) { g_crc32 = fn; }
static inline uint32_t crc_of(const uint8_t *p, uint64_t n)
{
    if (g_crc32) return g_crc32(0, p, (size_t)n);
    uint32_t k = (uint32_t)crc32(0L, Z_NULL, 0);
    for (uint64_t a = 0; a < n; a += 1u << 30) k = (uint32_t)crc32(k, p + a, (uInt)std::min<uint64_t>(1u << 30, n - a));
    return k;
}

// symbols -> bytes: a plain narrowing pass the compiler vectorises, and a second look only at the (rare) groups
// that hold a window symbol
static inline void resolve(const uint16_t *sy, uint64_t n, const uint8_t *w, uint8_t *dst)
{
    uint64_t i = 0;
    for (; i + 64 <= n; i += 64) {
        uint16_t any = 0;
        for (int k = 0; k < 64; k++) { const uint16_t v = sy[i + k]; any |= v; dst[i + k] = (uint8_t)v; }
        if (any & UNRESOLVED)
            for (int k = 0; k < 64; k++) { const uint16_t v = sy[i + k]; if (v & UNRESOLVED) dst[i + k] = w[v & (WSIZE - 1)]; }
    }
    for (; i < n; i++) { const uint16_t v = sy[i]; dst[i] = (v & UNRESOLVED) ? w[v & (WSIZE - 1)] : (uint8_t)v; }
}

// bgzip (BGZF) files: every member says how long it is (extra subfield 'B','C': BSIZE = member bytes - 1) and holds
// at most 64 KB of text, so the members are independent work items with known places in the text.  zlib's raw
// inflate per member, CRC-32 and ISIZE of every member checked.
static bool bgzf_gunzip(const uint8_t *in, uint64_t in_n, unsigned threads, uint64_t budget, char **text, uint64_t *len)
{
    struct Mem { uint64_t off, hdr, size, out_off; uint32_t isize, crc; };
    std::vector<Mem> ms;
    uint64_t pos = 0, total = 0;
    while (pos < in_n) {
        if (in_n - pos < 28) return false;
        const uint8_t *h = in + pos;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
        const uint64_t xlen = (uint64_t)h[10] | (uint64_t)h[11] << 8;
        if (12 + xlen > in_n - pos) return false;
        uint64_t bsize = 0;
        for (uint64_t x = 12; x + 4 <= 12 + xlen;) {          // subfields: SI1 SI2 SLEN(2) data
            const uint64_t slen = (uint64_t)h[x + 2] | (uint64_t)h[x + 3] << 8;
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = ((uint64_t)h[x + 4] | (uint64_t)h[x + 5] << 8) + 1;
            x += 4 + slen;
        }
        const uint64_t hl = gzip_header_len(h, in_n - pos);
        if (!bsize || !hl || bsize < hl + 8 || bsize > in_n - pos) return false;
        const uint8_t *t = h + bsize - 8;
        Mem m;
        m.off = pos; m.hdr = hl; m.size = bsize; m.out_off = total;
        m.crc = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
        m.isize = (uint32_t)t[4] | (uint32_t)t[5] << 8 | (uint32_t)t[6] << 16 | (uint32_t)t[7] << 24;
        if (m.isize > (1u << 16)) return false;                 // BGZF blocks hold at most 64 KB
        total += m.isize;
        ms.push_back(m);
        pos += bsize;
    }
    if (ms.empty() || total > budget) return false;
    void *mem = nullptr;
    const uint64_t cap = (std::max<uint64_t>(1, total) + (2u << 20) - 1) & ~(uint64_t)((2u << 20) - 1);
    if (posix_memalign(&mem, 2u << 20, cap) != 0 || !mem) return false;
    madvise(mem, cap, MADV_HUGEPAGE);
    std::atomic<uint64_t> next(0);
    std::atomic<bool> bad(false);
    auto worker = [&] {
        z_stream z;
        memset(&z, 0, sizeof(z));
        if (inflateInit2(&z, -15) != Z_OK) { bad = true; return; }
        for (uint64_t i0; !bad && (i0 = next.fetch_add(64)) < ms.size();)
            for (uint64_t i = i0; i < std::min<uint64_t>(ms.size(), i0 + 64); i++) {
                const Mem &m = ms[i];
                uint8_t *dst = (uint8_t *)mem + m.out_off;
                inflateReset(&z);
                z.next_in = const_cast<Bytef *>(in + m.off + m.hdr);
                z.avail_in = (uInt)(m.size - m.hdr - 8);
                z.next_out = dst;
                z.avail_out = m.isize;
                const int r = inflate(&z, Z_FINISH);
                if ((r != Z_STREAM_END) || z.avail_out != 0 || z.avail_in != 0 || crc_of(dst, m.isize) != m.crc) { bad = true; break; }
            }
        inflateEnd(&z);
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < std::max(1u, threads); t++) pool.emplace_back(worker);
    for (auto &th : pool) th.join();
    if (bad) { free(mem); return false; }
    *text = (char *)mem;
    *len = total;
    return true;
}

// Inflate the gzip file image `in` with `threads` threads.  On success *text is a buffer of *len bytes to be
// released with free().  false = not applicable or not verified: the caller uses another inflater.
//
// Streaming form of the scheme above: the deflate data is cut into many small chunks (1-4 MB), handed out IN ORDER
// to the threads.  A thread finds its chunk's entry point, inflates to the next chunk's entry point into ITS symbol
// buffer (reused for every chunk it takes: a few MB that stay in cache, instead of 2 bytes of fresh memory per byte
// of text), then -- as soon as the chunk before it has published them -- takes its place in the text and its window,
// passes both on to its successor, and writes its bytes.  The text is the only large allocation.
//
// Several members (lanes concatenated with `cat a.gz b.gz`): where a member ends is only known when its final
// block has been decoded.  The chunk that meets it cancels the chunks after it (they were decoding the next
// member's data ahead of time), the member's trailer is checked, and the pipeline starts again behind it.
// bgzip (BGZF) files, thousands of <= 64 KB members with their size in the header, take bgzf_gunzip above.
bool parallel_gunzip(const uint8_t *in, uint64_t in_n, unsigned threads, uint64_t budget, char **text, uint64_t *len,
                     const TextAlloc *ta)
{
    static const bool trace = getenv("SS_INGEST_TRACE") != nullptr;
    auto say = [&](const char *what, uint64_t a = 0, uint64_t b = 0) {
        if (trace) fprintf(stderr, "[pgz] %s (%llu, %llu)\n", what, (unsigned long long)a, (unsigned long long)b);
        return false;
    };
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[pgz]   %-12s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    if (!gzip_header_len(in, in_n) || threads < 2) return say("no gzip header / one thread", 0, threads);
    if (in_n < (uint64_t)threads * (1u << 20) + 64) return say("too small", in_n, threads);
    const uint64_t lim = in_n - 8;                       // no deflate data at or beyond this byte (the last trailer)
    // the text: ISIZE of the LAST member + k * 2^32 for a file of one member (the smallest such length that is not
    // below the file's own size is right for anything that deflates at all); several members: room for 6 x the file
    // (address space only; pages are touched as the text is written).  Too small a guess is caught: the text would
    // not fit, and the file goes to the caller's other inflaters.
    uint64_t guess = (uint32_t)in[lim + 4] | (uint32_t)in[lim + 5] << 8 | (uint32_t)in[lim + 6] << 16 | (uint64_t)in[lim + 7] << 24;
    while (guess < in_n) guess += 1ull << 32;
    const uint64_t sym_bytes = (uint64_t)threads * (64ull << 20);
    if (std::max<uint64_t>(guess, 4 * in_n) + sym_bytes > budget) return say("over the memory budget", guess, budget);
    void *mem = nullptr;
    const uint64_t text_cap = (std::max<uint64_t>(guess, 6 * in_n) + (2u << 20) - 1) & ~(uint64_t)((2u << 20) - 1);
    if (ta && ta->alloc) mem = ta->alloc(text_cap, ta->ctx);
    else if (posix_memalign(&mem, 2u << 20, text_cap) != 0) mem = nullptr;
    if (!mem) return say("no memory for the text", text_cap);
    madvise(mem, text_cap, MADV_HUGEPAGE);
    uint8_t *out = (uint8_t *)mem;
    auto drop_text = [&] { if (ta && ta->release) ta->release(mem, text_cap, ta->ctx); else free(mem); };

    constexpr uint64_t UNKNOWN = ~0ull, NONE = ~0ull - 1;
    struct alignas(64) Chunk {
        std::atomic<uint64_t> entry{~0ull};        // bit position, NONE, or UNKNOWN
        std::atomic<int> entry_state{0};            // 0 nobody looked, 1 being searched, 2 known
        std::atomic<bool> off_ready{false}, win_ready{false};
        uint64_t off = 0, n = 0;                    // place and length of the chunk's text
        uint32_t crc = 0;
        bool exists = false;
        uint64_t end_byte = 0;                      // where the member's trailer starts, if this chunk met the final block
        uint8_t *window = nullptr;                  // WSIZE bytes in front of the chunk (owned by the chunk)
    };
    std::atomic<uint64_t> t_entry(0), t_decode(0), t_wait(0), t_resolve(0), t_crc(0);     // microseconds, all threads (trace)
    uint64_t text_off = 0, mstart = 0, members = 0;

    while (mstart < in_n) {
        const uint64_t hl = gzip_header_len(in + mstart, in_n - mstart);
        if (!hl) { drop_text(); return say("what follows a member is not a gzip header", mstart, members); }
        {   // bgzip: an extra field with a 'B','C' subfield; thousands of 64 KB members are not for this pipeline
            const uint8_t *h = in + mstart;
            if ((h[3] & 4) && hl >= 18 && h[12] == 'B' && h[13] == 'C') {
                drop_text();
                if (mstart == 0 && !ta && bgzf_gunzip(in, in_n, threads, budget, text, len)) { say("ok, bgzf", *len); return true; }
                return say("bgzf, not taken", mstart);
            }
        }
        const uint64_t ds = mstart + hl;                 // first byte of deflate data of this member
        if (lim - ds < (2u << 20) && members > 0) { drop_text(); return say("small trailing member", lim - ds, members); }
        const uint64_t CH = std::min<uint64_t>(4u << 20, std::max<uint64_t>(1u << 20, (lim - ds) / ((uint64_t)threads * 8)));
        const uint64_t nch = std::max<uint64_t>(1, (lim - ds + CH - 1) / CH);
        std::vector<Chunk> ch(nch);
        std::atomic<bool> bad(false);
        std::atomic<uint64_t> next(0), total(UNKNOWN), cut(UNKNOWN);

        // entry point of chunk k: the first position in its byte range that parses as a dynamic block and decodes
        auto find_entry = [&](uint64_t k) -> uint64_t {
            if (k == 0) return ds * 8;
            const uint64_t lo = (ds + CH * k) * 8, hi = std::min<uint64_t>(lim, ds + CH * (k + 1)) * 8;
            for (uint64_t bp = lo; bp < hi; bp++) {
                const uint64_t byte = bp >> 3;      // BFINAL = 0, BTYPE = 10 (LSB first: 0, 0, 1)
                const uint32_t three = (((uint32_t)in[byte] | (uint32_t)in[byte + 1] << 8) >> (bp & 7)) & 7u;
                if (three != 4u) continue;
                Bits b(in, in_n, bp + 3);
                Codes c;
                if (!read_dynamic(b, c)) continue;
                Bits b2(in, in_n, bp);                // this block and the next two must come through without an error
                Out o;
                o.count_only = true;
                int r = 0;
                for (int q = 0; q < 3 && r == 0; q++) r = inflate_block(b2, o, false);
                if (r < 0) continue;
                return bp;
            }
            return NONE;
        };
        auto ensure_entry = [&](uint64_t k) -> uint64_t {
            Chunk &c = ch[k];
            int st = 0;
            if (c.entry_state.compare_exchange_strong(st, 1)) {
                c.entry.store(find_entry(k), std::memory_order_relaxed);
                c.entry_state.store(2, std::memory_order_release);
            } else {
                for (unsigned spin = 0; c.entry_state.load(std::memory_order_acquire) != 2; spin++) {
                    if (bad) return NONE;
                    if (spin < 256) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
                }
            }
            return c.entry.load(std::memory_order_relaxed);
        };
        // false: give up (an error somewhere, or chunk j lies behind the member's end)
        auto wait_for = [&](std::atomic<bool> &flag, uint64_t j) {
            for (unsigned spin = 0; !flag.load(std::memory_order_acquire); spin++) {
                if (bad || j > cut.load(std::memory_order_acquire)) return false;
                if (spin < 256) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            return true;
        };

        ch[0].window = (uint8_t *)calloc(WSIZE, 1);
        if (!ch[0].window) { drop_text(); return false; }
        ch[0].off = text_off;
        ch[0].off_ready = true;
        ch[0].win_ready = true;

        auto worker = [&] {
            auto now = [] { return std::chrono::steady_clock::now(); };
            auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
            Out o;                                           // this thread's symbols, reused chunk after chunk
            for (uint64_t j; !bad && (j = next.fetch_add(1)) < nch && j <= cut.load(std::memory_order_acquire);) {
                auto t0 = now();
                const uint64_t e = ensure_entry(j);
                if (e == NONE) continue;                     // no entry point in this range: the chunk before runs through it
                Chunk &c = ch[j];
                // the next chunk that has an entry point is where this one stops
                uint64_t s = j + 1, stop = UNKNOWN;
                for (; s < nch; s++) {
                    const uint64_t es = ensure_entry(s);
                    if (bad) return;
                    if (es != NONE) { stop = es; break; }
                }
                auto t1 = now();
                o.n = 0;
                Bits b(in, in_n, e);
                bool final_seen = false;
                for (;;) {
                    const int r = inflate_block(b, o, j == 0);
                    if (j > cut.load(std::memory_order_acquire)) break;     // this was the next member's data
                    if (r < 0 || bad) { bad = true; return; }
                    const uint64_t bp = b.bitpos();
                    if (r == 1) {                                    // the member's final block: its trailer follows
                        const uint64_t eb = (bp + 7) >> 3;
                        if (eb > lim) { bad = true; return; }
                        c.end_byte = eb;
                        // chunks behind this one are not part of the member.  A chunk that was decoding the NEXT member
                        // ahead of time may get here first with that member's final block: the smallest index wins
                        uint64_t cur = cut.load(std::memory_order_acquire);
                        while (j < cur && !cut.compare_exchange_weak(cur, j, std::memory_order_acq_rel)) {}
                        final_seen = true;
                        break;
                    }
                    if (bp == stop) break;
                    if (bp > stop) { bad = true; return; }           // ran over the next entry point: it was not a block start
                }
                if (j > cut.load(std::memory_order_acquire)) continue;
                c.exists = true;
                c.n = o.n;
                auto t2 = now();
                // place in the text, handed on to the successor at once
                if (!wait_for(c.off_ready, j)) return;
                if (c.off + c.n > text_cap) { bad = true; return; }
                if (final_seen) total.store(c.off + c.n, std::memory_order_release);
                else if (s < nch) { ch[s].off = c.off + c.n; ch[s].off_ready.store(true, std::memory_order_release); }
                // window of the successor = the last WSIZE bytes of the text up to here
                if (!wait_for(c.win_ready, j)) return;
                auto t3 = now();
                const uint8_t *w = c.window;
                if (!final_seen && s < nch) {
                    uint8_t *nw = (uint8_t *)malloc(WSIZE);
                    if (!nw) { bad = true; return; }
                    const uint64_t take = std::min<uint64_t>(WSIZE, o.n);
                    memcpy(nw, w + take, WSIZE - take);
                    resolve(o.p + (o.n - take), take, w, nw + (WSIZE - take));
                    ch[s].window = nw;
                    ch[s].win_ready.store(true, std::memory_order_release);
                }
                // the chunk's bytes and their CRC
                uint8_t *dst = out + c.off;
                resolve(o.p, o.n, w, dst);
                auto t4 = now();
                c.crc = crc_of(dst, o.n);
                if (trace) { auto t5 = now(); t_entry += us(t0, t1); t_decode += us(t1, t2); t_wait += us(t2, t3); t_resolve += us(t3, t4); t_crc += us(t4, t5); }
            }
        };
        {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < threads; t++) pool.emplace_back(worker);
            for (auto &th : pool) th.join();
        }
        const uint64_t last = cut.load(), tot = total.load();
        const uint64_t eb = last != UNKNOWN ? ch[last].end_byte : UNKNOWN;
        uint32_t all = (uint32_t)crc32(0L, Z_NULL, 0);
        for (uint64_t k = 0; k < nch; k++) {
            free(ch[k].window);
            if (ch[k].exists && k <= last && !bad) all = (uint32_t)crc32_combine(all, ch[k].crc, (z_off_t)ch[k].n);
        }
        if (bad || last == UNKNOWN || tot == UNKNOWN || !ch[last].exists) {
            drop_text();
            return say("a chunk failed, overran the next entry point, or no final block was found", members);
        }
        const uint32_t want_crc = (uint32_t)in[eb] | (uint32_t)in[eb + 1] << 8 | (uint32_t)in[eb + 2] << 16 | (uint32_t)in[eb + 3] << 24;
        const uint32_t want_size = (uint32_t)in[eb + 4] | (uint32_t)in[eb + 5] << 8 | (uint32_t)in[eb + 6] << 16 | (uint32_t)in[eb + 7] << 24;
        if ((uint32_t)(tot - text_off) != want_size) { drop_text(); return say("length differs from ISIZE", tot - text_off, want_size); }
        if (all != want_crc) { drop_text(); return say("CRC-32 differs", all, want_crc); }
        members++;
        text_off = tot;
        mstart = eb + 8;
        lap("member done");
        if (mstart < in_n && (eb - ds) < (2u << 20)) { drop_text(); return say("small members", eb - ds, members); }
    }
    if (trace) fprintf(stderr, "[pgz]   thread-seconds: entry search %.3f, decode %.3f, waiting %.3f, resolve %.3f, crc %.3f (%u threads)\n",
                       t_entry / 1e6, t_decode / 1e6, t_wait / 1e6, t_resolve / 1e6, t_crc / 1e6, threads);
    say("ok", text_off, members);
    *text = (char *)mem;
    *len = text_off;
    return true;
}

}  // namespace ss
