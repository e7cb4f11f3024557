// ss_ingest.hip -- multi-threaded FASTA/FASTQ ingest feeding the scan.
//
// The reference hands the read files to `jellyfish count -t 8` (library/identify.py:82-86), which
// parses them with its own threads.  Here a plain (uncompressed) file is mmap'ed and cut into
// chunks at RECORD boundaries found from arbitrary byte offsets:
//   FASTA : a line starting with '>' (sequence lines cannot start with '>')
//   FASTQ : a line starting with '@' whose second-next line starts with '+' and whose next and
//           third-next lines have equal length (a quality line may start with '@' or '+', but
//           then the line two below it is a header or a sequence, never a '+' line of that shape)
// Worker threads turn their chunk into a flat base block (same grammar as ss_fastx_to_flat) in a
// pinned buffer and enqueue H2D copy + scan kernel on their own HIP stream, so parsing, PCIe
// copies and kernels of different chunks overlap.  Counting is order independent (integer
// atomics), so the result equals the sequential path bit for bit.  Anything else (gzip input,
// multi-line FASTQ, files the boundary rule cannot segment) takes the sequential reader.
#include "ss_common.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// bytes of text per work item.  4, 8, 12 and 24 MB parse at the same steady rate
// (230-270 M reads/s with 20 threads); smaller chunks mean less pinned and private buffer memory to set up and
// touch for the first time in a process (20 x 27 MB at 24 MB chunks)
const uint64_t CHUNK = 8ull << 20;

struct Line { uint64_t s, e; };   // [s, e) without the '\n'; s == npos when absent

inline bool next_line(const char *t, uint64_t n, uint64_t pos, Line &ln)
{
    if (pos >= n) return false;
    const void *nl = memchr(t + pos, '\n', n - pos);
    ln.s = pos;
    ln.e = nl ? (uint64_t)((const char *)nl - t) : n;
    return true;
}

// first record start at or after `from` (from must be 0 or follow a '\n' search), or n
uint64_t sync_record(const char *t, uint64_t n, uint64_t from, bool fastq)
{
    uint64_t pos = from;
    if (pos > 0) {   // move to the start of the next line
        const void *nl = memchr(t + pos - 1, '\n', n - pos + 1);
        if (!nl) return n;
        pos = (uint64_t)((const char *)nl - t) + 1;
    }
    while (pos < n) {
        Line l0;
        if (!next_line(t, n, pos, l0)) return n;
        if (!fastq) {
            if (t[pos] == '>') return pos;
        } else if (t[pos] == '@') {
            Line l1, l2, l3;
            if (next_line(t, n, l0.e + 1, l1) && next_line(t, n, l1.e + 1, l2) && l2.e > l2.s && t[l2.s] == '+' &&
                next_line(t, n, l2.e + 1, l3) && (l3.e - l3.s) == (l1.e - l1.s))
                return pos;
        }
        pos = l0.e + 1;
    }
    return n;
}

// the head of the file must be plain 4-line FASTQ (or FASTA) for the chunked path
bool head_is_simple(const char *t, uint64_t n, bool &fastq)
{
    if (n == 0) return false;
    if (t[0] == '>') { fastq = false; return true; }
    if (t[0] != '@') return false;
    fastq = true;
    uint64_t pos = 0;
    for (int r = 0; r < 256 && pos < n; r++) {
        Line h, s, p, q;
        if (!next_line(t, n, pos, h) || t[h.s] != '@') return false;
        if (!next_line(t, n, h.e + 1, s) || !next_line(t, n, s.e + 1, p) || p.e == p.s || t[p.s] != '+') return false;
        if (!next_line(t, n, p.e + 1, q) || (q.e - q.s) != (s.e - s.s)) return false;
        pos = q.e + 1;
    }
    return true;
}

}  // namespace

namespace ss {
unsigned host_cpus()
{
    static const unsigned cached = [] {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        if (const char *e = getenv("SS_HOST_CPUS")) return (unsigned)std::max(1, atoi(e));
        long long quota = -1, period = 100000;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            if (fscanf(f, "%31s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(g, "%lld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
        }
        if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        return n;
    }();
    return cached;
}

// Parse threads of ONE process.  Measured on the MI355X host: ~21 M reads/s per thread up to ~20 threads, falling beyond 24.
// Several ranks on one node (torchrun sets LOCAL_WORLD_SIZE) share the node's CPUs -- and its cgroup quota, host_cpus() --
// so each takes its share: 8 ranks on 16 CPUs of quota run 2 parse threads each, not 8 x 20 threads that the scheduler
// throttles as a group.  SS_INGEST_THREADS overrides (per process).
unsigned ingest_threads()
{
    if (const char *e = getenv("SS_INGEST_THREADS")) return (unsigned)std::max(1, atoi(e));
    unsigned n = std::max(1u, std::thread::hardware_concurrency());      // (one process: as measured, the quota notwithstanding --
    if (const char *lw = getenv("LOCAL_WORLD_SIZE")) {                   //  parse threads spend half their time in pread and copies)
        const int w = atoi(lw);
        if (w > 1) n = std::max(2u, host_cpus() / (unsigned)w);
    }
    return std::min(n, 20u);
}
}  // namespace ss

extern "C" int ss_ingest_threads(int *n)
{
    if (!n) return SS_EINVAL;
    *n = (int)ss::ingest_threads();
    return SS_OK;
}

namespace ss {

// The parse threads share a few process-wide streams: creating a stream costs ~13 ms on this stack (a
// hardware queue each: 20 of them were 0.26 s of a 0.33 s first load), the runtime multiplexes streams
// onto four hardware queues anyway, and a worker waits for ITS work through an event, not a stream.
constexpr int N_INGEST_STREAMS = 4;
hipStream_t ingest_stream(unsigned i)
{
    static hipStream_t pool[N_INGEST_STREAMS];
    static std::once_flag once;
    std::call_once(once, [] {
        int device = 0;
        hipGetDevice(&device);
        std::vector<std::thread> th;                      // ~13 ms each: at the same time rather than one after another
        for (auto &st : pool)
            th.emplace_back([&st, device] {
                hipSetDevice(device);
                if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
            });
        for (auto &t : th) t.join();
    });
    return pool[i % N_INGEST_STREAMS];
}

// ---------------------------------------------------------------------------------------------
// gzip input.  zlib inflates ~350 MB/s of FASTQ text per thread and a gzip member has no entry points,
// so a .gz sample is normally bound by one inflate thread per file (the reference pipes `zcat` into
// jellyfish, identify.py:81-84: the same bound at a lower rate).  Here a .gz input is inflated WHOLE
// into memory, all .gz inputs of a call concurrently, and the text is then parsed by the chunked
// parser above like a plain file:
//   1. ss_pgz.hip: one member, up to 32 threads (entry points found inside the deflate stream, chunks
//      decoded against an unknown window, verified against the member's CRC-32 and length): 5-10 GB/s
//      of text for the decode itself; a sample of two 300 MB files is ready in ~0.35 s, most of it
//      the kernel handing out ~3 bytes of fresh memory per byte of text (huge pages asked for);
//   2. libdeflate when it is on the machine (dlopen: the image ships the runtime library, not its
//      header): one thread per file, ~550 MB/s (files of several members, small files);
//   3. otherwise, or when the text would not fit the memory budget (SS_INFLATE_MAX_GB, default a
//      quarter of the memory the process may use), the zlib reader streams the file as before.
// ---------------------------------------------------------------------------------------------
void pgz_set_crc32(uint32_t (*fn)(uint32_t, const void *, size_t));      // ss_pgz.hip
struct TextAlloc {                                                          // ss_pgz.hip
    void *(*alloc)(uint64_t cap, void *ctx) = nullptr;
    void (*release)(void *p, uint64_t cap, void *ctx) = nullptr;
    void *ctx = nullptr;
};
bool parallel_gunzip(const uint8_t *in, uint64_t in_n, unsigned threads, uint64_t budget, char **text, uint64_t *len,
                     const TextAlloc *ta);
namespace {
struct Deflate {
    void *(*alloc)() = nullptr;
    int (*gunzip)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    bool ok = false;
};
const Deflate &deflate_lib()
{
    static Deflate d;
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("SS_NO_LIBDEFLATE")) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        d.alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        d.gunzip = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(h, "libdeflate_gzip_decompress_ex");
        d.release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        d.ok = d.alloc && d.gunzip && d.release;
        if (void *c = dlsym(h, "libdeflate_crc32")) pgz_set_crc32((uint32_t (*)(uint32_t, const void *, size_t))c);
    });
    return d;
}
}  // namespace

// a text of several hundred MB is released on a background thread (returning its pages takes tens of ms)
void free_later(char *p)
{
    if (p) std::thread([p] { free(p); }).detach();
}

uint64_t inflate_budget_bytes()
{
    if (const char *e = getenv("SS_INFLATE_MAX_GB")) return (uint64_t)(atof(e) * 1e9);
    const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
    uint64_t mem = pages > 0 && psz > 0 ? (uint64_t)pages * (uint64_t)psz : (32ull << 30);
    // a container may be allowed far less than the machine has (cgroup v2, then v1)
    for (const char *f : {"/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"}) {
        FILE *fp = fopen(f, "r");
        if (!fp) continue;
        unsigned long long lim = 0;
        if (fscanf(fp, "%llu", &lim) == 1 && lim > 0 && lim < mem) mem = lim;      // "max" does not parse: no limit
        fclose(fp);
    }
    return mem / 4;
}

// path -> malloc'ed text of all its gzip members, or false (not gzip, no libdeflate, damaged, over `budget`)
// mode: 0 = parallel inflater when the file is large enough, else libdeflate; 1 = parallel only; 2 = libdeflate only
bool inflate_whole(const char *path, uint64_t budget, char **text, uint64_t *len, int mode, unsigned threads)
{
    const Deflate &L = deflate_lib();
    if (!L.ok && mode == 2) return false;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) { close(fd); return false; }
    const uint64_t in_n = (uint64_t)st.st_size;
    const unsigned char *in = (const unsigned char *)mmap(nullptr, in_n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (in == MAP_FAILED) return false;
    bool ok = in[0] == 0x1f && in[1] == 0x8b;
    if (ok && mode != 2 && !getenv("SS_NO_PGZ")) {
        // many threads on ONE member (ss_pgz.hip); verified against the trailer's CRC-32 and length
        if (!threads) threads = std::min<unsigned>(host_cpus(), 32u);
        if (const char *e = getenv("SS_PGZ_THREADS")) threads = (unsigned)std::max(1, atoi(e));
        threads = (unsigned)std::min<uint64_t>(threads, std::max<uint64_t>(1, in_n >> 21));      // >= 2 MB of input each
        if (parallel_gunzip(in, in_n, threads, budget, text, len, nullptr)) { munmap((void *)in, in_n); return true; }
    }
    if (mode == 1 || !L.ok) { munmap((void *)in, in_n); return false; }
    char *out = nullptr;
    uint64_t cap = 0, opos = 0;
    if (ok) {
        // ISIZE (last four bytes) = length of the LAST member mod 2^32: exact for the usual single-member file
        // below 4 GB of text; k * 2^32 is added while the text would be shorter than the file; a file of many
        // members (bgzip) gets four times its size; the buffer grows when a member does not fit
        uint64_t guess = (uint64_t)in[in_n - 4] | (uint64_t)in[in_n - 3] << 8 | (uint64_t)in[in_n - 2] << 16 | (uint64_t)in[in_n - 1] << 24;
        while (guess < in_n) guess += 1ull << 32;
        cap = std::max<uint64_t>(guess, std::min<uint64_t>(4 * in_n, guess + (1ull << 32))) + (64 << 10);
        if (guess <= (1ull << 20)) cap = 4 * in_n + (64 << 10);       // small last member: many members
        ok = cap <= budget && (out = (char *)malloc(cap)) != nullptr;
    }
    void *dec = ok ? L.alloc() : nullptr;
    ok = ok && dec;
    uint64_t ipos = 0;
    while (ok && ipos + 18 <= in_n && in[ipos] == 0x1f && in[ipos + 1] == 0x8b) {
        size_t ain = 0, aout = 0;
        const int r = L.gunzip(dec, in + ipos, in_n - ipos, out + opos, cap - opos, &ain, &aout);
        if (r == 0) { ipos += ain; opos += aout; continue; }
        if (r != 3) { ok = false; break; }                             // damaged data: let the zlib reader report it
        const uint64_t ncap = cap + std::max<uint64_t>(cap / 2, 1ull << 32);   // LIBDEFLATE_INSUFFICIENT_SPACE
        char *no = ncap <= budget ? (char *)realloc(out, ncap) : nullptr;
        if (!no) { ok = false; break; }
        out = no;
        cap = ncap;
    }
    if (dec) L.release(dec);
    munmap((void *)in, in_n);
    if (!ok || ipos == 0) { free(out); return false; }
    *text = out;
    *len = opos;
    return true;
}

}  // namespace ss

extern "C" {
int ss_gz_inflate(const char *path, int threads, int mode, char **text, uint64_t *len)
{
    if (!path || !text || !len || mode < 0 || mode > 2 || threads < 0) return SS_EINVAL;
    *text = nullptr;
    *len = 0;
    return ss::inflate_whole(path, ss::inflate_budget_bytes(), text, len, mode, (unsigned)threads) ? SS_OK : SS_ERANGE;
}
void ss_gz_free(char *text) { free(text); }

// The text of a .gz file written to `out_path` (a file on tmpfs, so that the ranks of one node share ONE inflate:
// strainscan_amd/dist.py).  The threaded inflater writes straight into a shared mapping of the file; otherwise
// libdeflate inflates to memory and the text is written out.
int ss_gz_inflate_to_file(const char *path, const char *out_path, int threads, uint64_t *len)
{
    if (!path || !out_path || !len || threads < 0) return SS_EINVAL;
    *len = 0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return SS_EIO;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) { close(fd); return SS_ERANGE; }
    const uint64_t in_n = (uint64_t)st.st_size;
    const uint8_t *in = (const uint8_t *)mmap(nullptr, in_n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (in == MAP_FAILED) return SS_EIO;
    struct Ctx { const char *out_path; int fd; } ctx{out_path, -1};
    ss::TextAlloc ta;
    ta.ctx = &ctx;
    ta.alloc = [](uint64_t cap, void *c) -> void * {
        Ctx *x = (Ctx *)c;
        x->fd = open(x->out_path, O_RDWR | O_CREAT | O_TRUNC, 0600);
        if (x->fd < 0 || ftruncate(x->fd, (off_t)cap) != 0) return nullptr;          // sparse: pages come as they are written
        void *m = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, x->fd, 0);
        return m == MAP_FAILED ? nullptr : m;
    };
    ta.release = [](void *p, uint64_t cap, void *c) {
        Ctx *x = (Ctx *)c;
        munmap(p, cap);
        if (x->fd >= 0) { close(x->fd); x->fd = -1; }
        unlink(x->out_path);
    };
    unsigned t = threads ? (unsigned)threads : std::min<unsigned>(ss::host_cpus(), 32u);
    t = (unsigned)std::min<uint64_t>(t, std::max<uint64_t>(1, in_n >> 21));
    char *text = nullptr;
    uint64_t n = 0;
    int rc = SS_ERANGE;
    if (!getenv("SS_NO_PGZ") && ss::parallel_gunzip(in, in_n, t, ss::inflate_budget_bytes(), &text, &n, &ta)) {
        // the mapping's capacity was an upper bound: cut the file to the text
        struct stat so;
        const uint64_t cap = fstat(ctx.fd, &so) == 0 ? (uint64_t)so.st_size : n;
        munmap(text, cap);
        rc = ftruncate(ctx.fd, (off_t)n) == 0 ? SS_OK : SS_EIO;
        close(ctx.fd);
        if (rc != SS_OK) unlink(out_path);
    } else {
        if (ctx.fd >= 0) { close(ctx.fd); unlink(out_path); }
        if (ss::inflate_whole(path, ss::inflate_budget_bytes(), &text, &n, 2, 0)) {
            FILE *f = fopen(out_path, "wb");
            rc = (f && fwrite(text, 1, n, f) == n) ? SS_OK : SS_EIO;
            if (f && fclose(f) != 0) rc = SS_EIO;
            if (rc != SS_OK) unlink(out_path);
            free(text);
        }
    }
    munmap((void *)in, in_n);
    if (rc == SS_OK) *len = n;
    return rc;
}
}

namespace ss {

// All gzip inputs of a call inflated concurrently (one thread per file); entry i stays empty when path i is not
// gzip or cannot be inflated here.  The caller frees the texts.
std::vector<InflatedText> inflate_gz_inputs(const char *const *paths, int n_paths)
{
    std::vector<InflatedText> out((size_t)std::max(0, n_paths));
    const uint64_t budget = inflate_budget_bytes() / (uint64_t)std::max(1, n_paths);
    std::vector<int> gz;
    for (int i = 0; i < n_paths; i++) {
        if (!paths[i] || !paths[i][0]) continue;
        unsigned char magic[2] = {0, 0};
        FILE *f = fopen(paths[i], "rb");
        if (!f) continue;
        if (fread(magic, 1, 2, f) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) gz.push_back(i);
        fclose(f);
    }
    if (gz.empty()) return out;
    // the files inflate concurrently and share the CPUs the process may use (not the machine's hardware threads)
    const unsigned per_file = std::max(1u, std::min(32u, host_cpus() / (unsigned)gz.size()));
    std::vector<std::thread> pool;
    for (int i : gz)
        pool.emplace_back([&out, paths, i, budget, per_file] { if (!inflate_whole(paths[i], budget, &out[i].p, &out[i].n, 0, per_file)) out[i].p = nullptr; });
    for (auto &th : pool) th.join();
    return out;
}

int gz_inputs_on_device(const char *const *paths, int n_paths, int shard_rank, int shard_world,
                        const std::function<int(int, char *, uint64_t, uint64_t, uint64_t)> &flat, std::vector<InflatedText> &texts,
                        std::vector<char> &done)
{
    texts.assign((size_t)std::max(0, n_paths), InflatedText());
    done.assign((size_t)std::max(0, n_paths), 0);
    if (!gz_on_gpu()) return SS_OK;
    std::vector<int> gz;
    for (int i = 0; i < n_paths; i++) {
        if (!paths[i] || !paths[i][0]) continue;
        unsigned char magic[2] = {0, 0};
        FILE *f = fopen(paths[i], "rb");
        if (!f) continue;
        if (fread(magic, 1, 2, f) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) gz.push_back(i);
        fclose(f);
    }
    if (gz.empty()) return SS_OK;
    if (gz_range_active() && gz_policy() == 1) {
        // the ranks share every file's inflation (ss_gz_set_range); a rank keeps all records that begin in its slices.  The
        // chain of messages between the ranks is per file and the files use it in the order of the paths (gz_range_ticket), but
        // the files are in flight together: while one file's messages travel, the next one's slices are uploaded and inflated
        int device = 0;
        hipGetDevice(&device);
        std::mutex mu;
        int rc = SS_OK;
        std::vector<uint64_t> ticket;
        for (size_t q = 0; q < gz.size(); q++) ticket.push_back(gz_range_ticket());
        constexpr bool serial = false;
        std::vector<std::thread> pool;
        for (size_t q = 0; q < gz.size(); q++) {
            auto one = [&, q] {
                const int i = gz[q];
                hipSetDevice(device);
                const int r = gz_fastq_pieces_dev(paths[i], ticket[q], [&, i](char *d, uint64_t len, uint64_t cap, uint64_t nrec) { return flat(i, d, len, cap, nrec); });
                std::lock_guard<std::mutex> g(mu);
                if (r == 0) done[i] = 1;
                else if (r == 1) rc = SS_EAGAIN;      // (the chain of the remaining files is still served: every rank goes through all of them)
                else if (rc == SS_OK) rc = r;
            };
            if (serial) one();
            else pool.emplace_back(one);
            if (!serial && pool.size() == 2 && q + 1 < gz.size()) { pool.front().join(); pool.erase(pool.begin()); }      // two files in flight: a pair
        }
        for (auto &th : pool) th.join();
        return rc;
    }
    int device = 0;
    hipGetDevice(&device);
    std::atomic<int> err(SS_OK);
    std::vector<std::thread> pool;
    for (int i : gz)
        pool.emplace_back([&, i] {
            hipSetDevice(device);
            char *d = nullptr;
            uint64_t len = 0, cap = 0, nrec = 0;
            const int r = gz_fastq_to_flat_dev(paths[i], shard_rank, shard_world, &d, &len, &cap, &nrec, &texts[i].p, &texts[i].n);
            if (r == 0) {
                const int rc = flat(i, d, len, cap, nrec);
                if (rc != SS_OK) err = rc;
                done[i] = 1;
            } else if (r == 1 && gz_policy() == 1) {
                err = SS_EAGAIN;          // strict policy: a declined input is the caller's to settle with the other ranks
            }
        });
    for (auto &th : pool) th.join();
    return err;
}

// returns SS_OK and *handled = true when the file was scanned here; *handled = false => caller
// must use the sequential reader for this file
int scan_file_parallel(ss_db *db, const char *path, uint64_t *n_records, uint64_t *n_bases, bool *handled, int shard_rank,
                       int shard_world)
{
    // SS_INGEST_ZEROCOPY=1: the kernel streams the flat block straight out of the pinned host buffer
    // (every base is read once, with 16-byte loads) instead of waiting for a DMA copy of it
    constexpr bool zero_copy = false;
    return parse_file_parallel(db->workers, path, shard_rank, shard_world, n_records, n_bases, handled,
                               [db](const char *h_buf, char *d_buf, uint64_t len, hipStream_t stream) {
                                   return ss_scan_flat_dev(db, zero_copy ? h_buf : d_buf, len, stream);
                               }, !zero_copy);
}

int scan_text_parallel(ss_db *db, const char *text, uint64_t n, uint64_t *n_records, uint64_t *n_bases, bool *handled,
                       int shard_rank, int shard_world)
{
    return parse_text_parallel(db->workers, text, n, nullptr, shard_rank, shard_world, n_records, n_bases, handled,
                               [db](const char *, char *d_buf, uint64_t len, hipStream_t stream) {
                                   return ss_scan_flat_dev(db, d_buf, len, stream);
                               }, true);
}

// Parse `path` with worker threads; each flat block (already copied to the worker's device buffer
// on `stream`) is handed to `sink`.  Chunks c with c % shard_world != shard_rank are skipped
// (multi-GPU read sharding without parsing the other ranks' share).
int parse_file_parallel(ss_db::Worker *workers, const char *path, int shard_rank, int shard_world,
                        uint64_t *n_records, uint64_t *n_bases, bool *handled, const BlockSink &sink, bool copy)
{
    *handled = false;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return SS_EIO;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < (off_t)(4 << 20)) { close(fd); return SS_OK; }
    const uint64_t n = (uint64_t)st.st_size;
    const char *t = (const char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (t == MAP_FAILED) return SS_OK;
    if ((unsigned char)t[0] == 0x1f) { munmap((void *)t, n); return SS_OK; }
    madvise((void *)t, n, MADV_SEQUENTIAL);
    const int rc = parse_text_parallel(workers, t, n, path, shard_rank, shard_world, n_records, n_bases, handled, sink, copy);
    munmap((void *)t, n);
    return rc;
}

// The chunked parse of a text that is in memory: a mapped file (`path` then names it: chunks are pread() from it)
// or a buffer (path == nullptr: an inflated .gz, parsed where it lies).
int parse_text_parallel(ss_db::Worker *workers, const char *t, uint64_t n, const char *path, int shard_rank,
                        int shard_world, uint64_t *n_records, uint64_t *n_bases, bool *handled, const BlockSink &sink,
                        bool copy)
{
    *handled = false;
    bool fastq = true;
    if (n < (4u << 20) || !head_is_simple(t, n, fastq)) return SS_OK;

    // chunk starts at record boundaries
    std::vector<uint64_t> starts;
    starts.push_back(0);
    for (uint64_t off = CHUNK; off < n; off += CHUNK) {
        const uint64_t s = sync_record(t, n, off, fastq);
        if (s >= n) break;
        if (s > starts.back()) starts.push_back(s);
    }
    starts.push_back(n);
    const size_t n_chunks = starts.size() - 1;
    uint64_t max_chunk = 0;
    for (size_t c = 0; c < n_chunks; c++) max_chunk = std::max(max_chunk, starts[c + 1] - starts[c]);
    if (max_chunk > 8 * CHUNK) return SS_OK;   // a giant record: sequential path

    // parse threads: each owns a text buffer, a pinned buffer, a device buffer and a stream.  Measured on
    // the MI355X host (scripts/bench_e2e.py): 21 M reads/s per thread up to ~20 threads (256 M reads/s =
    // 79 GB/s of FASTQ text, 39 GB/s over PCIe), falling again beyond 24.  SS_INGEST_THREADS overrides.
    unsigned nthreads = ss::ingest_threads();
    nthreads = std::min<unsigned>(nthreads, (unsigned)ss_db::MAX_WORKERS);
    nthreads = (unsigned)std::min<size_t>(nthreads, n_chunks);
    std::atomic<size_t> next(0);
    std::atomic<uint64_t> recs(0), bases(0);
    std::atomic<int> err(SS_OK);
    // chunks are pread() into a private buffer before parsing: parsing the mapping directly takes a minor page
    // fault every 4 KB (1.2 M faults per 5 GB) and runs at half the rate.
    const bool use_pread = path != nullptr;
    const int fd2 = use_pread ? open(path, O_RDONLY) : -1;
    if (use_pread && fd2 < 0) return SS_EIO;
    int device = 0;
    hipGetDevice(&device);
    static const bool trace = getenv("SS_INGEST_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    auto worker = [&](unsigned wid) {
        hipSetDevice(device);
        const double t_dev = since();
        ss_db::Worker &W = workers[wid];
        if (W.cap < max_chunk + 64 || (copy && !W.d_buf)) {   // first use (or a larger chunk than ever before)
            if (W.h_buf) hipHostFree(W.h_buf);
            if (W.d_buf) hipFree(W.d_buf);
            W.h_buf = W.d_buf = nullptr;
            W.cap = 0;
            const uint64_t cap = std::max<uint64_t>(max_chunk + 64, CHUNK + CHUNK / 8);
            if (hipHostMalloc((void **)&W.h_buf, cap, hipHostMallocDefault) != hipSuccess ||
                (copy && hipMalloc((void **)&W.d_buf, cap) != hipSuccess))     // a sink that places the block itself needs none
                err = SS_ENOMEM;
            else
                W.cap = cap;
        }
        if (!W.done && hipEventCreateWithFlags(&W.done, hipEventDisableTiming) != hipSuccess) err = SS_EHIP;
        char *h_buf = W.h_buf, *d_buf = W.d_buf;
        hipStream_t stream = ingest_stream(wid);
        if (!stream) err = SS_EHIP;
        if (trace) fprintf(stderr, "[ingest] worker %u: device %.4f s, ready %.4f s\n", wid, t_dev, since());
        for (size_t c; err == SS_OK && (c = next.fetch_add(1)) < n_chunks;) {
            if ((int)(c % (size_t)shard_world) != shard_rank) continue;
            uint64_t out_len = 0, nr = 0;
            const char *src = t + starts[c];
            const uint64_t clen = starts[c + 1] - starts[c];
            if (use_pread) {
                if (W.t_cap < clen) {
                    free(W.t_buf);
                    W.t_cap = std::max<uint64_t>(max_chunk, CHUNK + CHUNK / 8);
                    W.t_buf = (char *)malloc(W.t_cap);
                    if (!W.t_buf) { W.t_cap = 0; err = SS_ENOMEM; break; }
                }
                uint64_t got = 0;
                while (got < clen) {
                    const ssize_t r = pread(fd2, W.t_buf + got, clen - got, (off_t)(starts[c] + got));
                    if (r <= 0) break;
                    got += (uint64_t)r;
                }
                if (got != clen) { err = SS_EIO; break; }
                src = W.t_buf;
            }
            int rc = ss_fastx_to_flat(src, clen, h_buf, &out_len, &nr);
            if (rc != SS_OK) { err = rc; break; }
            if (copy && hipMemcpyAsync(d_buf, h_buf, out_len, hipMemcpyHostToDevice, stream) != hipSuccess) { err = SS_EHIP; break; }
            rc = sink(h_buf, d_buf, out_len, stream);
            if (rc != SS_OK) { err = rc; break; }
            if (hipEventRecord(W.done, stream) != hipSuccess || hipEventSynchronize(W.done) != hipSuccess) { err = SS_EHIP; break; }   // buffers are reused
            recs += nr;
            bases += out_len;
        }
    };
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < nthreads; w++) pool.emplace_back(worker, w);
    for (auto &th : pool) th.join();
    if (trace) fprintf(stderr, "[ingest] %s: %zu chunks, %u threads, done %.4f s\n", path ? path : "(inflated text)", n_chunks, nthreads, since());
    if (fd2 >= 0) close(fd2);
    if (err != SS_OK) return err;
    *n_records += recs;
    *n_bases += bases;
    *handled = true;
    return SS_OK;
}

}  // namespace ss

void ss_db::free_workers(Worker *w, int n)
{
    for (int i = 0; i < n; i++) {
        if (w[i].h_buf) hipHostFree(w[i].h_buf);
        if (w[i].d_buf) hipFree(w[i].d_buf);
        if (w[i].done) hipEventDestroy(w[i].done);
        free(w[i].t_buf);
        w[i] = Worker();
    }
}

// ---------------------------------------------------------------------------------------------
// ss_reads: a read set parsed once and kept in HBM as flat base blocks.
// The reference re-reads (and jellyfish re-parses) the whole FASTQ for the tree scan, for every
// identified multi-strain cluster and twice more with -b (identify.py:409, Vote_...:354-372,
// identify_low_depth.py:119,124).  With 288 GB of HBM the 2-byte-per-base text is parsed and
// shipped over PCIe once; every later scan is a 10 ms kernel over resident blocks.
// ---------------------------------------------------------------------------------------------
namespace {
// pinned buffers and streams of the parse threads, shared by every ss_reads_load of the process (pinning
// 20 x 27 MB costs more than parsing a small sample); one load at a time
ss_db::Worker g_read_workers[ss_db::MAX_WORKERS];
std::mutex g_read_workers_mu;
}  // namespace

extern "C" {

// What the first ss_reads_load of a process pays before it parses anything: the parse threads' pinned buffers (20 x 9 MB:
// ~0.1 s the first time), their text buffers, events and the ingest streams.  A command-line process calls this on a worker
// thread while the interpreter is still importing modules (strainscan_amd/_lib.py warm_up): 0.1 s off `strainscan`'s 0.65 s.
int ss_ingest_warm_up(void)
{
    std::lock_guard<std::mutex> pool_lock(g_read_workers_mu);
    unsigned nthreads = ss::ingest_threads();
    nthreads = std::min<unsigned>(nthreads, (unsigned)ss_db::MAX_WORKERS);
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return SS_ENODEV;
    const uint64_t cap = CHUNK + CHUNK / 8;
    std::atomic<int> err(SS_OK);
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < nthreads; w++)
        pool.emplace_back([&, w] {
            hipSetDevice(device);
            ss_db::Worker &W = g_read_workers[w];
            if (!W.h_buf) {
                if (hipHostMalloc((void **)&W.h_buf, cap, hipHostMallocDefault) == hipSuccess) W.cap = cap;
                else { W.h_buf = nullptr; err = SS_ENOMEM; }
            }
            if (!W.t_buf) {
                W.t_buf = (char *)malloc(cap);
                W.t_cap = W.t_buf ? cap : 0;
                if (W.t_buf) for (uint64_t i = 0; i < cap; i += 4096) W.t_buf[i] = 0;      // first touch here, not under the parser
            }
            if (!W.done && hipEventCreateWithFlags(&W.done, hipEventDisableTiming) != hipSuccess) err = SS_EHIP;
            if (!ss::ingest_stream(w)) err = SS_EHIP;
        });
    for (auto &th : pool) th.join();
    return err;
}

int ss_reads_load(const char *const *paths, int n_paths, int shard_rank, int shard_world, ss_reads **out)
{
    if (!paths || n_paths < 1 || !out || shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world) return SS_EINVAL;
    ss_reads *R = new (std::nothrow) ss_reads();
    if (!R) return SS_ENOMEM;
    uint64_t recs = 0, bases = 0;
    {   // first slab: FASTQ text is a little over 2 bytes per base, this rank's share of it
        uint64_t text = 0;
        for (int i = 0; i < n_paths; i++) {
            struct stat st;
            if (paths[i] && paths[i][0] && stat(paths[i], &st) == 0) text += (uint64_t)st.st_size;
        }
        R->first_slab = text / 2 / (uint64_t)shard_world + text / 50 + (32ull << 20);
    }
    std::lock_guard<std::mutex> pool_lock(g_read_workers_mu);
    const auto t_load = std::chrono::steady_clock::now();
    auto load_since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_load).count(); };
    // the parse thread's pinned block goes straight to its place in a slab
    auto keep = [R](const char *h_buf, char *, uint64_t len, hipStream_t stream) -> int {
        char *dst = R->reserve(len);
        if (!dst) return SS_ENOMEM;
        const uint64_t plen = ss_reads::padded(len);
        memset(const_cast<char *>(h_buf) + len, '\n', plen - len);       // the pinned buffer has 64 bytes of slack
        return hipMemcpyAsync(dst, h_buf, plen, hipMemcpyHostToDevice, stream) == hipSuccess ? SS_OK : SS_EHIP;
    };
    int rc = SS_OK;
    // plain files: chunked worker-thread path; everything else (gzip, multi-line records, small files)
    // goes through the sequential reader -- one thread PER FILE, so the two mates of a paired
    // .fastq.gz sample inflate concurrently (zlib is the limiter there)
    std::vector<int> seq_files;
    // (not SS_GZ_GPU=0) .gz inputs are inflated AND reduced to their sequence lines on the device; the block becomes a
    // slab as it is.  What that path only inflated (not strict four-line FASTQ) arrives as text, the rest goes on below.
    std::vector<ss::InflatedText> texts;
    std::vector<char> on_device;
    std::vector<const char *> rest(paths, paths + n_paths);
    {
        std::atomic<uint64_t> drecs(0), dbases(0);
        rc = ss::gz_inputs_on_device(paths, n_paths, shard_rank, shard_world, [&](int, char *d, uint64_t len, uint64_t cap, uint64_t nrec) {
            R->adopt(d, cap, len);
            drecs += nrec;
            dbases += len;
            return (int)SS_OK;
        }, texts, on_device);
        recs += drecs;
        bases += dbases;
        for (int i = 0; i < n_paths; i++)
            if (on_device[i] || texts[i].p) rest[i] = "";
    }
    if (rc == SS_OK) {
        std::vector<ss::InflatedText> more = ss::inflate_gz_inputs(rest.data(), n_paths);     // .gz inputs, inflated concurrently
        if (texts.empty()) texts = more;
        else for (int i = 0; i < n_paths; i++) if (more[i].p) texts[i] = more[i];
    }
    if (on_device.empty()) on_device.assign((size_t)n_paths, 0);
    if (getenv("SS_INGEST_TRACE")) fprintf(stderr, "[ingest] ss_reads_load: gz inputs inflated at %.4f s\n", load_since());
    for (int i = 0; i < n_paths && rc == SS_OK; i++) {
        if (!paths[i]) { rc = SS_EINVAL; break; }
        if (!paths[i][0] || on_device[i]) continue;
        bool handled = false;
        if (texts[i].p)
            rc = ss::parse_text_parallel(g_read_workers, texts[i].p, texts[i].n, nullptr, shard_rank, shard_world, &recs,
                                         &bases, &handled, keep, false);
        else
            rc = ss::parse_file_parallel(g_read_workers, paths[i], shard_rank, shard_world, &recs, &bases, &handled, keep, false);
        ss::free_later(texts[i].p);
        texts[i].p = nullptr;
        if (getenv("SS_INGEST_TRACE")) fprintf(stderr, "[ingest] ss_reads_load: file %d parsed at %.4f s\n", i, load_since());
        if (rc == SS_OK && !handled) seq_files.push_back(i);
    }
    for (auto &tx : texts) free(tx.p);
    if (rc == SS_OK && !seq_files.empty()) {
        std::atomic<int> err(SS_OK);
        std::atomic<uint64_t> srecs(0), sbases(0);
        int device = 0;
        hipGetDevice(&device);
        auto one_file = [&](int fi) {
            hipSetDevice(device);
            ss_reader *rd = nullptr;
            int r = ss_reader_open(&paths[fi], 1, &rd);
            if (r) { err = r; return; }
            ss_reader_set_overlap(rd, 30);
            const uint64_t cap = 32ull << 20;
            std::vector<char> buf(cap + 32);
            for (uint64_t blk = 0; err == SS_OK; blk++) {
                uint64_t len = 0, nr = 0;
                r = ss_reader_next(rd, buf.data(), cap, &len, &nr);
                if (r) { err = r; break; }
                if (len == 0) break;
                if (buf[len - 1] != '\n') R->has_cut_record = true;
                if ((int)((blk + (uint64_t)fi) % (uint64_t)shard_world) != shard_rank) continue;
                char *dst = R->reserve(len);
                if (!dst) { err = SS_ENOMEM; break; }
                const uint64_t plen = ss_reads::padded(len);
                memset(buf.data() + len, '\n', plen - len);
                if (hipMemcpy(dst, buf.data(), plen, hipMemcpyHostToDevice) != hipSuccess) { err = SS_EHIP; break; }
                srecs += nr;
                sbases += len;
            }
            ss_reader_close(rd);
        };
        std::vector<std::thread> pool;
        for (int fi : seq_files) pool.emplace_back(one_file, fi);
        for (auto &th : pool) th.join();
        rc = err;
        recs += srecs;
        bases += sbases;
    }
    if (rc != SS_OK) { ss_reads_destroy(R); return rc; }
    const double t_parsed = load_since();
    if (hipDeviceSynchronize() != hipSuccess) { ss_reads_destroy(R); return SS_EHIP; }
    if (getenv("SS_INGEST_TRACE")) fprintf(stderr, "[ingest] ss_reads_load: files done %.4f s, device idle %.4f s\n", t_parsed, load_since());
    R->n_records = recs;
    R->n_bases = bases;
    rc = ss::reads_order_for_locality(R);      // unless SS_READS_ORDER=file (ss_reorder.hip)
    if (rc != SS_OK) { ss_reads_destroy(R); return rc; }
    if (getenv("SS_INGEST_TRACE")) fprintf(stderr, "[ingest] ss_reads_load: ordered for locality at %.4f s\n", load_since());
    *out = R;
    return SS_OK;
}

int ss_reads_from_flat_dev(const void *flat_dev, uint64_t n, int order, ss_reads **out)
{
    if (!out || (n && !flat_dev)) return SS_EINVAL;
    ss_reads *R = new (std::nothrow) ss_reads();
    if (!R) return SS_ENOMEM;
    R->first_slab = n + 64;
    R->n_bases = n;
    if (n && order && n >= 64) {
        // binned straight out of the caller's block (ss_reorder.hip): no intermediate copy
        char *d = nullptr;
        uint64_t used = 0, cap = 0;
        const int rc = ss::order_flat_dev(static_cast<const char *>(flat_dev), n, &d, &used, &cap);
        if (rc != SS_OK) { ss_reads_destroy(R); return rc; }
        ss_reads::Slab sl;
        sl.d = d; sl.cap = cap; sl.used = used; sl.binned = true;
        R->slabs.push_back(sl);
        R->device_bytes = cap;
        R->n_blocks = 1;
    } else if (n) {
        char *dst = R->reserve(n);
        if (!dst) { ss_reads_destroy(R); return SS_ENOMEM; }
        const uint64_t plen = ss_reads::padded(n);
        // (both only enqueued on the null stream when they return: the scans may run on any stream)
        if (hipMemcpy(dst, flat_dev, n, hipMemcpyDeviceToDevice) != hipSuccess ||
            hipMemset(dst + n, '\n', plen - n) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { ss_reads_destroy(R); return SS_EHIP; }
    }
    *out = R;
    return SS_OK;
}

int ss_reads_order_timing(double out_ms[3])
{
    if (!out_ms) return SS_EINVAL;
    ss::reorder_timing(out_ms);
    return SS_OK;
}

int ss_reads_order_counters(uint64_t out[2])
{
    if (!out) return SS_EINVAL;
    ss::reorder_counters(out);
    return SS_OK;
}

int ss_reads_read_back(const ss_reads *R, char *host, uint64_t cap, uint64_t *len)
{
    if (!R || !len) return SS_EINVAL;
    uint64_t total = 0;
    for (const auto &sl : R->slabs) total += sl.used;
    *len = total;
    if (!host) return SS_OK;                       // size query
    if (cap < total) return SS_ERANGE;
    uint64_t o = 0;
    for (const auto &sl : R->slabs) {
        if (sl.used) SS_HIP(hipMemcpy(host + o, sl.d, sl.used, hipMemcpyDeviceToHost));
        o += sl.used;
    }
    return SS_OK;
}

int ss_reads_destroy(ss_reads *R)
{
    if (!R) return SS_OK;
    // The large slabs are kept for the next sample of this process (ss::big_put) instead of going back to the driver.  hipFree
    // waited for everything in flight on the device; keeping a block does not, and ss_scan_reads is asynchronous on the CALLER's
    // stream, which this library cannot name -- so the wait is made here, once per read set, exactly where hipFree made it.
    if (!R->slabs.empty()) (void)hipDeviceSynchronize();
    for (auto &sl : R->slabs) ss::big_put(sl.d, sl.cap);
    delete R;
    return SS_OK;
}

int ss_reads_info(const ss_reads *R, uint64_t *n_records, uint64_t *n_bases, uint64_t *n_blocks, uint64_t *device_bytes)
{
    if (!R) return SS_EINVAL;
    if (n_records) *n_records = R->n_records;
    if (n_bases) *n_bases = R->n_bases;
    if (n_blocks) *n_blocks = R->n_blocks;
    if (device_bytes) *device_bytes = R->device_bytes;
    return SS_OK;
}

int ss_scan_reads(ss_db *db, const ss_reads *R, void *stream)
{
    if (!db || !R) return SS_EINVAL;
    int k = 0;
    ss_db_info(db, nullptr, nullptr, nullptr, &k);
    if (R->has_cut_record && k != 31) return SS_ERANGE;   // cut records carry a 30-base overlap
    for (const auto &sl : R->slabs) {
        if (!sl.used) continue;
        int rc = ss::scan_flat_dev(db, sl.d, sl.used, stream, sl.binned, R->serial);
        if (rc) return rc;
    }
    return SS_OK;
}

// The resident reads against SEVERAL tables in one pass (Vote_Strain_L2_Lasso_new_sp.py:295-296 loops over the identified
// clusters, :354-372 re-reads the FASTQ for each): tables of the minimizer layout go four at a time through one kernel that
// makes a tile's codes, minimizers and runs once; anything else is scanned on its own.
int ss_scan_reads_multi(ss_db *const *dbs, int n_dbs, const ss_reads *R, void *stream)
{
    if (!dbs || n_dbs < 1 || !R) return SS_EINVAL;
    std::vector<ss_db *> mini_all;
    for (int i = 0; i < n_dbs; i++) {
        if (!dbs[i]) return SS_EINVAL;
        for (int j = 0; j < i; j++) if (dbs[j] == dbs[i]) return SS_EINVAL;           // a table twice would count twice
        int k = 0;
        ss_db_info(dbs[i], nullptr, nullptr, nullptr, &k);
        if (R->has_cut_record && k != 31) return SS_ERANGE;
        // (several tables per pass: k = 31, and k >= 20 where scan_mini_kernel with k at run time serves every table -- ss_mini.hip launch_scan_mini)
        if (dbs[i]->layout == 1 && k >= 20) mini_all.push_back(dbs[i]);
        else { int rc = ss_scan_reads(dbs[i], R, stream); if (rc) return rc; }
    }
    // the tables of ONE k go through the several-tables kernel together (a tile's minimizers are made once per k)
    std::stable_sort(mini_all.begin(), mini_all.end(), [](const ss_db *a, const ss_db *b) { return a->k < b->k; });
    constexpr int group = 4;
    for (size_t k0 = 0; k0 < mini_all.size();) {
    size_t k1 = k0;
    while (k1 < mini_all.size() && mini_all[k1]->k == mini_all[k0]->k) k1++;
    std::vector<ss_db *> mini(mini_all.begin() + (long)k0, mini_all.begin() + (long)k1);
    k0 = k1;
    for (size_t g = 0; g < mini.size(); g += (size_t)group) {
        const int ng = (int)std::min<size_t>((size_t)group, mini.size() - g);
        for (const auto &sl : R->slabs) {
            if (!sl.used) continue;
            int rc = ng == 1 ? ss::scan_flat_dev(mini[g], sl.d, sl.used, stream, sl.binned, R->serial)
                             : (sl.used < (uint64_t)mini[g]->k ? SS_OK : ss::launch_scan_mini_multi(&mini[g], ng, sl.d, sl.used, ss::as_stream(stream), sl.binned));
            if (rc) return rc;
        }
    }
    }
    return SS_OK;
}

}  // extern "C"
