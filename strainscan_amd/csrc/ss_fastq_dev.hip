// ss_fastq_dev.hip -- a .fastq.gz sample from the file to the flat base block without the text visiting the host.
//
// The reference pipes `zcat` into jellyfish (library/identify.py:81-84).  With the member inflated on the device
// (ss_ginflate.hip) the 307 MB of text per million reads would otherwise cross PCIe to the host parser and half of it
// come back: here the sequence lines are picked out where the text already is.  Only STRICT four-line FASTQ is taken --
// every record exactly "@...", sequence, "+...", quality of the sequence's length, no blank lines -- for which the general
// grammar of ss_fastx_to_flat (ss_host.hip; multi-line records, FASTA) yields exactly "sequence line + '\n'" per record;
// anything else is handed to that host grammar as text.
//   1  newlines per 4 KB tile                      (16 bytes per lane, SWAR newline mask)
//   2  exclusive sum -> every line's start          (one 8-byte entry per line)
//   3  per record: the four checks, length of the sequence line
//   4  exclusive sum -> copy, 16 lanes per record
#include "ss_common.h"

#include <atomic>

#include <hipcub/hipcub.hpp>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <vector>

namespace {

constexpr int TB = 4096;                    // bytes of text per workgroup (256 lanes x 16)

// newline mask of the 16 bytes at i0 (bit i = byte i is '\n'); bytes at or beyond n are no newlines
__device__ __forceinline__ uint32_t nl_mask_at(const char *t, uint64_t n, uint64_t i0)
{
    if (i0 >= n) return 0;
    const uint4 v = *reinterpret_cast<const uint4 *>(t + i0);          // the buffer has slack behind n
    uint32_t m = 0;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t x = w[d] ^ 0x0A0A0A0Au;
        const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;      // 0x80 where the byte was '\n'
        m |= (((z >> 7) | (z >> 14) | (z >> 21) | (z >> 28)) & 0xFu) << (4 * d);
    }
    const uint64_t left = n - i0;
    return left >= 16 ? m : m & ((1u << left) - 1u);
}

// WRITE = false: newlines per tile.  WRITE = true: ls[j + 1] = position behind the j-th newline of the text.
template <bool WRITE>
__global__ __launch_bounds__(256) void fq_lines_kernel(const char *__restrict__ t, uint64_t n, uint64_t *__restrict__ counts,
                                                       const uint64_t *__restrict__ base, uint64_t *__restrict__ ls)
{
    __shared__ uint32_t s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * TB + (uint64_t)tid * 16;
    const uint32_t nl = nl_mask_at(t, n, i0);
    const uint32_t c = (uint32_t)__popc(nl);
    uint32_t incl = c;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (!WRITE) {
        if (tid == 0) counts[blockIdx.x] = (uint64_t)s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        return;
    }
    uint64_t j = base[blockIdx.x] + (incl - c);
    for (int w = 0; w < wave; w++) j += s_wave[w];
    for (uint32_t m = nl; m; m &= m - 1) ls[++j] = i0 + (uint32_t)__ffs(m);            // (ffs is 1-based: the byte BEHIND the newline)
    if (blockIdx.x == 0 && tid == 0) ls[0] = 0;
}

// record r = lines 4r .. 4r + 3.  bad |= 1 when it is not the strict form; len1[r] = sequence line + its '\n' -- or 0 when
// the record belongs to another rank of a sharded run (blocks of SHARD_RECORDS records go round the ranks)
constexpr int SHARD_LOG2 = 12;
__global__ void fq_records_kernel(const char *__restrict__ t, uint64_t n, const uint64_t *__restrict__ ls, uint64_t n_nl,
                                  uint64_t n_rec, uint32_t shard_rank, uint32_t shard_world, uint64_t *__restrict__ len1,
                                  uint32_t *__restrict__ bad)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    const uint64_t s0 = ls[4 * r], s1 = ls[4 * r + 1], s2 = ls[4 * r + 2], s3 = ls[4 * r + 3];
    const uint64_t e3 = 4 * r + 3 < n_nl ? ls[4 * r + 4] - 1 : n;                          // the last line may lack its '\n'
    const uint64_t seq = s2 - 1 - s1;
    const bool ok = t[s0] == '@' && t[s2] == '+' && (seq == 0 || t[s1] != '+') && e3 - s3 == seq;
    if (!ok) atomicOr(bad, 1u);
    len1[r] = (r >> SHARD_LOG2) % shard_world == shard_rank ? seq + 1 : 0;
}

// sequence line of record r (with its '\n') -> dst + off[r]; 16 lanes per record, 16 (unaligned) bytes per lane and round
__global__ __launch_bounds__(256) void fq_copy_kernel(const char *__restrict__ t, const uint64_t *__restrict__ ls,
                                                      const uint64_t *__restrict__ len1, const uint64_t *__restrict__ off,
                                                      uint64_t n_rec, char *__restrict__ dst)
{
    const uint64_t r = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (r >= n_rec) return;
    const uint64_t s = ls[4 * r + 1], l1 = len1[r], o = off[r];
    for (uint64_t c = (uint64_t)(threadIdx.x & 15) * 16; c < l1; c += 256) {
        if (c + 16 <= l1) {
            uint4 v;
            __builtin_memcpy(&v, t + s + c, 16);
            __builtin_memcpy(dst + o + c, &v, 16);
        } else {
            for (uint64_t k = c; k < l1; k++) dst[o + k] = t[s + k];
        }
    }
}

__global__ void fq_pad_kernel(char *dst, uint64_t from, uint64_t to)
{
    const uint64_t i = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < to) dst[i] = '\n';
}

}  // namespace

namespace ss {

// on unless SS_GZ_GPU=0 (the host inflaters of ss_pgz.hip / libdeflate / zlib then take every .gz input, as they take what
// the device path declines)
static std::atomic<int> g_gz_policy{0};      // ss_gz_set_policy: 0 device then host, 1 device or SS_EAGAIN, 2 host
int gz_policy() { return g_gz_policy.load(); }

bool gz_on_gpu()
{
    if (g_gz_policy.load() == 2) return false;
    const char *e = getenv("SS_GZ_GPU");
    return !e || strcmp(e, "0") != 0;
}

// FASTQ text on the device -> a new device buffer with the flat base block (padded like a block of ss_reads: at least one
// '\n' behind it, a multiple of 16 bytes).  0 = done, 1 = not strict four-line FASTQ (nothing returned), < 0 = SS_E*.
int fastq_text_to_flat_dev(const char *d_text, uint64_t n, int shard_rank, int shard_world, char **d_flat, uint64_t *flat_len,
                           uint64_t *flat_cap, uint64_t *n_records)
{
    if (n == 0) return 1;
    hipStream_t st = call_stream_get();
    if (!st) return SS_EHIP;
    const uint64_t n_tiles = (n + TB - 1) / TB;
    uint64_t *d_counts = nullptr, *d_base = nullptr, *d_ls = nullptr, *d_len1 = nullptr, *d_off = nullptr;
    uint32_t *d_bad = nullptr;
    void *d_tmp = nullptr;
    char *flat = nullptr;
    auto done = [&](int r) {
        void *scratch[] = {d_counts, d_base, d_ls, d_len1, d_off, d_bad, d_tmp};
        for (void *q : scratch) if (q) hipFreeAsync(q, st);
        hipStreamSynchronize(st);
        call_stream_put(st);
        if (r != 0 && flat) hipFree(flat);
        return r;
    };
#define FQ(call) do { if ((call) != hipSuccess) return done(SS_EHIP); } while (0)
    FQ(hipMallocAsync((void **)&d_counts, (n_tiles + 1) * 8, st));
    FQ(hipMallocAsync((void **)&d_base, (n_tiles + 1) * 8, st));
    hipLaunchKernelGGL(fq_lines_kernel<false>, dim3((unsigned)n_tiles), dim3(256), 0, st, d_text, n, d_counts, (const uint64_t *)nullptr, (uint64_t *)nullptr);
    FQ(hipMemsetAsync(d_counts + n_tiles, 0, 8, st));
    size_t tmp_bytes = 0;
    FQ(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_counts, d_base, (int)(n_tiles + 1), st));
    size_t tmp_have = tmp_bytes;
    FQ(hipMallocAsync(&d_tmp, std::max<size_t>(tmp_have, 16), st));
    FQ(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_counts, d_base, (int)(n_tiles + 1), st));
    uint64_t n_nl = 0;
    char last = 0;
    FQ(hipMemcpyAsync(&n_nl, d_base + n_tiles, 8, hipMemcpyDeviceToHost, st));
    FQ(hipMemcpyAsync(&last, d_text + n - 1, 1, hipMemcpyDeviceToHost, st));
    FQ(hipStreamSynchronize(st));
    const uint64_t n_lines = n_nl + (last != '\n' ? 1 : 0);
    if (n_lines == 0 || n_lines % 4 != 0 || n_lines / 4 > 0x7FFFFFF0ull) return done(1);
    const uint64_t n_rec = n_lines / 4;
    FQ(hipMallocAsync((void **)&d_ls, (n_nl + 2) * 8, st));
    hipLaunchKernelGGL(fq_lines_kernel<true>, dim3((unsigned)n_tiles), dim3(256), 0, st, d_text, n, (uint64_t *)nullptr, d_base, d_ls);
    FQ(hipMallocAsync((void **)&d_len1, (n_rec + 1) * 8, st));
    FQ(hipMallocAsync((void **)&d_off, (n_rec + 1) * 8, st));
    FQ(hipMallocAsync((void **)&d_bad, 4, st));
    FQ(hipMemsetAsync(d_bad, 0, 4, st));
    FQ(hipMemsetAsync(d_len1 + n_rec, 0, 8, st));
    hipLaunchKernelGGL(fq_records_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, st, d_text, n, d_ls, n_nl, n_rec, (uint32_t)shard_rank,
                       (uint32_t)shard_world, d_len1, d_bad);
    size_t tmp2 = 0;
    FQ(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, d_len1, d_off, (int)(n_rec + 1), st));
    if (tmp2 > tmp_have) {
        FQ(hipFreeAsync(d_tmp, st));
        d_tmp = nullptr;
        FQ(hipMallocAsync(&d_tmp, tmp2, st));
        tmp_have = tmp2;
    }
    FQ(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp2, d_len1, d_off, (int)(n_rec + 1), st));
    uint32_t bad = 0;
    uint64_t total = 0;
    FQ(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
    FQ(hipMemcpyAsync(&total, d_off + n_rec, 8, hipMemcpyDeviceToHost, st));
    FQ(hipStreamSynchronize(st));
    if (bad) return done(1);
    const uint64_t cap = ss_reads::padded(total);
    uint64_t real_cap = cap;                             // (a kept block may be larger)
    if (ss::big_malloc((void **)&flat, cap, &real_cap) != hipSuccess) return done(SS_ENOMEM);
    hipLaunchKernelGGL(fq_copy_kernel, dim3((unsigned)((n_rec + 15) / 16)), dim3(256), 0, st, d_text, d_ls, d_len1, d_off, n_rec, flat);
    hipLaunchKernelGGL(fq_pad_kernel, dim3(1), dim3(64), 0, st, flat, total, cap);
    FQ(hipGetLastError());
    FQ(hipStreamSynchronize(st));
#undef FQ
    *d_flat = flat;
    *flat_len = total;
    *flat_cap = real_cap;
    // this rank's records: the blocks b = rank, rank + world, ... of SHARD_RECORDS records (the last one may be short)
    uint64_t own = 0;
    const uint64_t blk = 1ull << SHARD_LOG2, n_blk = (n_rec + blk - 1) / blk;
    for (uint64_t b = (uint64_t)shard_rank; b < n_blk; b += (uint64_t)shard_world) own += std::min<uint64_t>(blk, n_rec - b * blk);
    *n_records = own;
    return done(0);
}

// One .gz file: inflated on the device (ss_ginflate.hip), the sequence lines extracted there.  In a sharded run every
// rank inflates the whole file on its own GPU (a gzip stream has no entry points to share out; 30-40 ms per 130 MB) and
// keeps the records of its blocks.
//   0  *d_flat (hipMalloc, *flat_cap bytes, padded with '\n') holds the *flat_len bytes of the flat block
//   1  not handled here (not one gzip member, damaged, no room, ...): nothing returned
//   2  inflated, but not strict four-line FASTQ: the text is returned on the host (*text, malloc) for the general grammar
int gz_fastq_to_flat_dev(const char *path, int shard_rank, int shard_world, char **d_flat, uint64_t *flat_len, uint64_t *flat_cap,
                         uint64_t *n_records, char **text, uint64_t *text_len)
{
    static const bool trace = getenv("SS_INGEST_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (g_hook_decline.load()) return 1;                       // test hook (ss_test_hook): this process declines
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return 1;
    struct stat sb;
    // (a small file is inflated on the host before the device path has allocated its buffers)
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < (1 << 20)) { close(fd); return 1; }
    const uint64_t in_n = (uint64_t)sb.st_size;
    // (a plain mapping, copied by one thread: a populated mapping, a read into memory and four copy threads were all
    //  slower -- 0.157 / 0.233 / 0.144 s against 0.129 s for a pair of 290 MB files, scripts/archive/dev/t_gz_input_ab.py)
    const uint8_t *in = (const uint8_t *)mmap(nullptr, in_n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (in == MAP_FAILED) { close(fd); return 1; }
    char *d_text = nullptr;
    uint64_t n = 0;
    void *lease = nullptr;
    const bool ok = in[0] == 0x1f && in[1] == 0x8b && gpu_gunzip(in, in_n, &d_text, &n, &lease, fd);
    close(fd);
    munmap((void *)in, in_n);
    if (!ok) return 1;
    if (trace) fprintf(stderr, "[ingest] %s: %.1f MB of text on the device at %.4f s\n", path, n / 1e6, since());
    int rc = fastq_text_to_flat_dev(d_text, n, shard_rank, shard_world, d_flat, flat_len, flat_cap, n_records);
    if (trace) fprintf(stderr, "[ingest] %s: sequence lines extracted (rc %d) at %.4f s\n", path, rc, since());
    if (rc == 0) { gpu_gunzip_done(lease); return 0; }
    // a failure of this process's own (rc < 0) under the strict policy is a decline, not a reason to parse the text here:
    // the other ranks would keep their blocks of records while this one keeps parse chunks
    if (rc < 0 && gz_policy() == 1) { gpu_gunzip_done(lease); return 1; }
    // the general grammar runs on the host
    char *h = n <= inflate_budget_bytes() ? (char *)malloc(std::max<uint64_t>(n, 1)) : nullptr;
    const bool got = h && (n == 0 || hipMemcpy(h, d_text, n, hipMemcpyDeviceToHost) == hipSuccess);
    gpu_gunzip_done(lease);
    if (!got) { free(h); return 1; }
    *text = h;
    *text_len = n;
    return 2;
}

// The same when several ranks share the file's inflation (range mode, ss_gz_set_range): this rank gets the text of ITS slices
// (gpu_gunzip_range), every piece = the bytes of the record that straddles the cut in front of it + its own complete records;
// the sequence lines of each piece become a flat block handed to `flat`.  0 = done (every record that begins in this rank's
// slices has been handed over), 1 = declined (nothing usable was handed over: the caller reports SS_EAGAIN and all ranks
// settle for another path).
int gz_fastq_pieces_dev(const char *path, uint64_t ticket, const std::function<int(char *, uint64_t, uint64_t, uint64_t)> &flat)
{
    struct Turn { uint64_t t; ~Turn() { gz_range_pass(t); } } turn{ticket};      // (whichever way this file leaves: the next one's chain may start)
    static const bool trace = getenv("SS_INGEST_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    std::vector<GzPiece> pieces;
    char *d_text = nullptr;
    void *lease = nullptr;
    bool ok = false;
    {
        const int fd = open(path, O_RDONLY);
        struct stat sb;
        const uint8_t *in = nullptr;
        uint64_t in_n = 0;
        bool inj = g_hook_decline.load() != 0;                // test hook: this rank declines
        std::vector<uint8_t> head;
        if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size >= 32) {
            in_n = (uint64_t)sb.st_size;
            in = (const uint8_t *)mmap(nullptr, in_n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (in == MAP_FAILED) {
                // this rank cannot map the file (its own trouble: address space, a limit): it declines, but it still serves
                // the chain -- the slices follow from the file's size and the length of the gzip header, which a plain
                // read of the first bytes gives (a header is a few dozen bytes; 70 KB cover the longest legal one)
                in = nullptr;
                head.resize((size_t)std::min<uint64_t>(in_n, 70u << 10));
                if (pread(fd, head.data(), head.size(), 0) == (ssize_t)head.size()) { in = head.data(); inj = true; }
            }
        }
        // (a rank that declines still serves the chain -- gpu_gunzip_range sees to that once it knows the slices, which it
        //  derives from the file's size; a file that cannot be opened at all is the same failure on every rank of a node, and
        //  a rank that is gone altogether ends the others' bounded wait: dist._gz_chain)
        if (in && in[0] == 0x1f && in[1] == 0x8b) ok = gpu_gunzip_range(in, in_n, &d_text, &lease, fd, &pieces, ticket, inj);
        if (in && head.empty()) munmap((void *)in, in_n);
        if (fd >= 0) close(fd);
    }
    if (!ok) return 1;
    if (trace) fprintf(stderr, "[ingest] %s: %zu pieces of this rank on the device at %.4f s\n", path, pieces.size(), since());
    int rc = 0;
    for (const GzPiece &pc : pieces) {
        const uint64_t n = pc.carry.size() + pc.keep;
        if (!n) continue;
        char *buf = nullptr;
        if (hipMalloc((void **)&buf, n + 16) != hipSuccess) { rc = 1; break; }
        bool good = (pc.carry.empty() || hipMemcpy(buf, pc.carry.data(), pc.carry.size(), hipMemcpyHostToDevice) == hipSuccess) &&
                    (!pc.keep || hipMemcpy(buf + pc.carry.size(), d_text + pc.at, pc.keep, hipMemcpyDeviceToDevice) == hipSuccess);
        // (a device-to-device hipMemcpy is only ENQUEUED on the null stream when it returns, and the extraction below runs on
        //  a non-blocking stream of its own: seen as a piece that was "not FASTQ" once in twenty runs on a busy device)
        good = good && hipStreamSynchronize(nullptr) == hipSuccess;
        char *d_flat = nullptr;
        uint64_t flen = 0, fcap = 0, nrec = 0;
        const int r = good ? fastq_text_to_flat_dev(buf, n, 0, 1, &d_flat, &flen, &fcap, &nrec) : SS_EHIP;
        hipFree(buf);
        if (r != 0) { rc = 1; break; }                         // not four-line FASTQ (or a failure of this rank's own): declined
        const int fr = flat(d_flat, flen, fcap, nrec);
        if (fr != SS_OK) { rc = fr; break; }
    }
    gpu_gunzip_done(lease);
    if (trace) fprintf(stderr, "[ingest] %s: sequence lines of the pieces extracted (rc %d) at %.4f s\n", path, rc, since());
    return rc;
}

}  // namespace ss

extern "C" int ss_gz_set_policy(int mode)
{
    if (mode < 0 || mode > 2) return SS_EINVAL;
    ss::g_gz_policy.store(mode);
    return SS_OK;
}
