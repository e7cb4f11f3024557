// Internal helpers shared by the gfx950 translation units.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <functional>
#include <vector>
#include <string.h>

#include "strainscan_hip.h"

namespace ss {

void set_last_error(const char *what, const char *file, int line, hipError_t e);

#define SS_HIP(call)                                               \
    do {                                                           \
        hipError_t _e = (call);                                    \
        if (_e != hipSuccess) {                                    \
            ss::set_last_error(#call, __FILE__, __LINE__, _e);     \
            return (_e == hipErrorOutOfMemory) ? SS_ENOMEM         \
                   : (_e == hipErrorNoDevice)  ? SS_ENODEV         \
                                               : SS_EHIP;          \
        }                                                          \
    } while (0)

constexpr uint64_t EMPTY_KEY = ~0ull;

// Table hash.  Two odd multipliers with a fold in between: the keys are overlapping windows of
// genomes (key[p+1] = key[p] >> 2 | base << 2(k-1)), so a single multiplicative hash would map
// neighbouring windows to related slots.
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x *= 0x9E3779B97F4A7C15ull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 29;
    return x;
}

__host__ __device__ __forceinline__ uint32_t slot_of(uint64_t key, uint32_t log2cap)
{
    return (uint32_t)(mix64(key) >> (64 - log2cap));
}
// flat layout's Bloom filter (ss_scan.hip): bit index from the LOW half of the same mix (the table slot takes the top bits)
__host__ __device__ __forceinline__ uint32_t bloom_bit_of(uint64_t mixed, uint32_t bloom_bits)
{
    return (uint32_t)(mixed & 0xFFFFFFFFull) >> (32 - bloom_bits);
}

// ascii -> 2-bit code ((c >> 1) & 3: A0 C1 T2 G3) or -1
__host__ __device__ __forceinline__ int base_code(unsigned char c)
{
    unsigned char u = c & 0xDF;
    if (u == 'A' || u == 'C' || u == 'G' || u == 'T') return (c >> 1) & 3;
    return -1;
}

inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }

// CPUs this process may really use: hardware threads, capped by the cgroup's CPU quota (v2 cpu.max, v1 cfs quota).
// A container with 16 CPUs of quota on a 256-thread host runs 64 busy threads SLOWER than 16 (CFS throttles the
// whole group once the quota of a period is spent).
unsigned host_cpus();
extern std::atomic<long long> g_hook_generic_k;      // ss_test_hook 4 (ss_mini.hip)
unsigned ingest_threads();      // FASTQ parse threads of this process: its share of host_cpus() (LOCAL_WORLD_SIZE), at most 20 (ss_ingest.hip)

}  // namespace ss

// The opaque database handle.
struct ss_db {
    int k = 0;
    int device = 0;
    uint64_t n_rows = 0;
    uint64_t n_distinct = 0;
    uint32_t log2cap = 0;
    uint64_t capacity = 0;
    int layout = 0;                    // 0 = flat open-address table, 1 = minimizer pages (17 <= k <= 31)
    uint64_t n_slots = 0;              // length of d_counts: capacity (flat) or n_mslots + 8 * n_dir (pages)
    uint64_t n_mslots = 0;             // pages: length of d_mkeys (bucket headers + k-mers)
    uint64_t n_inline = 0;             // pages: database k-mers held inline in page slots
    uint64_t *d_keys = nullptr;        // flat: [capacity] table keys, EMPTY_KEY where free
    uint64_t *d_mkeys = nullptr;       // pages: [n_mslots] buckets of the minimizers with many k-mers (ss_mini.hip)
    uint64_t *d_dir = nullptr;         // pages: [n_dir][8] 64-byte pages of slots (inline k-mers, bucket references)
    uint32_t dirbits = 0;
    uint32_t n_dir = 0;                // pages: number of home pages (range of page_of)
    uint32_t n_dir_alloc = 0;          // pages: home pages + spare pages behind them (overflow never wraps)
    uint32_t *d_bloom = nullptr;       // buckets: one-probe Bloom filter over the minimizers (L2 resident), or null
    uint32_t bloom_bits = 0;           // log2 of its size in bits
    uint64_t n_buckets = 0;
    uint32_t *d_counts = nullptr;      // [n_slots] occurrences per slot (accumulated by scans)
    int expect_hits = 0;               // ss_db_expect_hits: most read k-mers are in the table (a layer-2 cluster table)
    uint64_t probe_set = 0;            // the resident read set (ss_reads::serial) whose first tiles were last probed against this
    int probe_comb = 0;                // table, and what they said: add the hits up in LDS (ss_mini.hip choose_comb)
    double probe_runs_per_tile = 0;    // found runs per tile of that probe (ss_db_info_ex)
    uint32_t *d_slot_of_row = nullptr; // [n_rows]   slot owning row i, SS_NO_SLOT if none
    uint8_t *d_row_valid = nullptr;    // [n_rows]   1 iff row i is a key of match_results
    // pinned staging for host-resident base blocks
    char *h_stage[2] = {nullptr, nullptr};
    char *d_stage[2] = {nullptr, nullptr};
    uint64_t stage_bytes = 0;
    hipStream_t streams[2] = {nullptr, nullptr};
    hipEvent_t stage_free[2] = {nullptr, nullptr};
    std::atomic<uint64_t> launches{0};
    // per-worker resources of the parallel ingest path (allocated on first use, kept for the handle's life)
    struct Worker { char *h_buf = nullptr; char *d_buf = nullptr; hipEvent_t done = nullptr; uint64_t cap = 0;
                    char *t_buf = nullptr; uint64_t t_cap = 0; };   // t_buf: private copy of the text chunk being parsed;
                                                                    // done: this worker's last copy/kernel has finished
    static constexpr int MAX_WORKERS = 64;
    Worker workers[MAX_WORKERS];
    static void free_workers(Worker *w, int n);
    uint64_t device_bytes = 0;
};

#include <chrono>
#include <mutex>

// A sample's reads resident in HBM (ss_ingest.hip loads them, ss_reorder.hip orders them for locality).
namespace ss { hipError_t big_malloc(void **p, uint64_t bytes, uint64_t *got = nullptr); void big_put(void *p, uint64_t cap); }      // (below)
struct ss_reads {
    // Blocks live back to back in a few large device slabs; every block is followed by at least one '\n'
    // and padded with '\n' to a multiple of 16 bytes, so a slab is itself one flat base block: one scan
    // launch per slab (one in all for a typical sample) instead of one per 12 MB block, and no device
    // allocation per block while loading.
    struct Slab { char *d = nullptr; uint64_t cap = 0, used = 0; bool binned = false; };   // binned: ss_reorder.hip has ordered its records
    std::vector<Slab> slabs;
    std::mutex mu;
    uint64_t n_records = 0, n_bases = 0, device_bytes = 0, n_blocks = 0;
    bool has_cut_record = false;      // a record longer than a block was cut with a 30-base overlap (k = 31 only)
    uint64_t first_slab = 0;          // size of the first slab (estimate from the file sizes)
    uint64_t serial = next_serial();  // names this set: a table remembers which set it was last probed with
    static uint64_t next_serial() { static std::atomic<uint64_t> c{0}; return ++c; }

    static uint64_t padded(uint64_t len) { return (len + 1 + 15) & ~15ull; }
    // a flat block that is already on the device (hipMalloc'ed, padded as above) becomes a slab of its own
    void adopt(char *d, uint64_t cap, uint64_t len)
    {
        std::lock_guard<std::mutex> g(mu);
        Slab sl;
        sl.d = d; sl.cap = cap; sl.used = padded(len);
        if (!slabs.empty() && slabs.back().used < slabs.back().cap) slabs.insert(slabs.end() - 1, sl);      // the open slab stays last
        else slabs.push_back(sl);
        device_bytes += cap;
        n_blocks++;
    }
    // room for a block of `len` bytes (+ padding); nullptr when the device is out of memory
    char *reserve(uint64_t len)
    {
        const uint64_t need = padded(len);
        std::lock_guard<std::mutex> g(mu);
        if (slabs.empty() || slabs.back().used + need > slabs.back().cap) {
            Slab sl;
            sl.cap = std::max<uint64_t>(need, slabs.empty() ? std::max<uint64_t>(first_slab, 64ull << 20) : 512ull << 20);
            const auto t0 = std::chrono::steady_clock::now();
            if (ss::big_malloc((void **)&sl.d, sl.cap, &sl.cap) != hipSuccess) return nullptr;
            if (getenv("SS_INGEST_TRACE"))
                fprintf(stderr, "[ingest] slab of %.0f MB: %.4f s\n", sl.cap / 1e6,
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            slabs.push_back(sl);
        }
        Slab &sl = slabs.back();
        char *p = sl.d + sl.used;
        sl.used += need;
        device_bytes += need;
        n_blocks++;
        return p;
    }
};


namespace ss {
// Records of every slab re-ordered by the minimizer of their first k-mer (ss_reorder.hip); `force`: ignore SS_READS_ORDER
int reads_order_for_locality(ss_reads *R, bool force = false);
int order_flat_dev(const char *src, uint64_t n, char **out_d, uint64_t *out_used, uint64_t *out_cap);
void reorder_release();
void reorder_counters(uint64_t out[2]);      // slabs binned by the one-length passes / by the general ones, in this process
void reorder_timing(double out[3]);      // the last binning call: count + prefix, slab allocation, place (ms)      // the binning scratch kept between calls goes back to the device (ss_gz_gpu_release)
int build_mini(ss_db *db, const uint64_t *keys, const uint8_t *flags, uint64_t n_rows, int upper_keys);
// the same index built on the device (ss_build_dev.hip); anything but SS_OK / SS_EKEY: nothing was built, use the host build
int build_mini_dev(ss_db *db, const uint64_t *keys, const uint8_t *flags, uint64_t n_rows, int upper_keys);
int mark_solid(ss_db *db);      // PG_SOLID flags of the bucket references, after either build (ss_mini.hip)
using BlockSink = std::function<int(const char *h_buf, char *d_buf, uint64_t len, hipStream_t stream)>;
int parse_file_parallel(ss_db::Worker *workers, const char *path, int shard_rank, int shard_world,
                        uint64_t *n_records, uint64_t *n_bases, bool *handled, const BlockSink &sink, bool copy = true);
int parse_text_parallel(ss_db::Worker *workers, const char *text, uint64_t n, const char *path, int shard_rank,
                        int shard_world, uint64_t *n_records, uint64_t *n_bases, bool *handled, const BlockSink &sink,
                        bool copy = true);
struct InflatedText { char *p = nullptr; uint64_t n = 0; };
bool inflate_whole(const char *path, uint64_t budget, char **text, uint64_t *len, int mode, unsigned threads);
uint64_t inflate_budget_bytes();
hipStream_t ingest_stream(unsigned i);   // a few process-wide non-blocking streams (creating one costs ~13 ms)
void free_later(char *p);
std::vector<InflatedText> inflate_gz_inputs(const char *const *paths, int n_paths);
// unless SS_GZ_GPU=0: .gz inputs are inflated on the device (ss_ginflate.hip) and strict four-line FASTQ is turned into the flat
// base block there (ss_fastq_dev.hip)
bool gz_on_gpu();
int gz_policy();                         // ss_gz_set_policy: 0 device then host, 1 device or SS_EAGAIN, 2 host
bool gpu_gunzip(const uint8_t *in, uint64_t in_n, char **text_dev, uint64_t *len, void **lease, int fd);      // the text is lent until ...
void gpu_gunzip_done(void *lease);
// a non-blocking stream for the length of a call, from a small pool (making and destroying one is ~0.6 ms); handed back synchronised
hipStream_t call_stream_get();
void call_stream_put(hipStream_t s);
// range mode (ss_gz_set_range): this rank's slices of a member.  text[at, at + len) is a slice's text, its first `keep` bytes
// end with the last record that is complete in it, `carry` holds the bytes of the record that began in the slice before
struct GzPiece { uint64_t at, len, keep; std::vector<uint8_t> carry; };
bool gz_range_active();
extern std::atomic<long long> g_hook_entry, g_hook_decline, g_hook_skip_chain;      // ss_test_hook (ss_ginflate.hip)
// The files of a load go down the chain in the order of their paths (the messages of one pair of ranks are matched in order): a
// file draws a ticket (in path order, before its thread starts), its first chain call waits for the files before it to have
// finished THEIR chain traffic, and gz_range_pass hands the turn on (idempotent; waits for the turn itself if the file never
// used it).  What a file does before its first chain call -- upload, search, the inflation of its first slice -- runs beside
// the chain of the file in front of it: both mates of a pair are in flight on every rank.
uint64_t gz_range_ticket();
void gz_range_pass(uint64_t ticket);
bool gpu_gunzip_range(const uint8_t *in, uint64_t in_n, char **text_dev, void **lease, int fd, std::vector<GzPiece> *pieces,
                      uint64_t ticket, bool decline = false);      // decline: test hook -- plan the slices, serve the chain, hand nothing back
int gz_fastq_pieces_dev(const char *path, uint64_t ticket, const std::function<int(char *, uint64_t, uint64_t, uint64_t)> &flat);
int gz_fastq_to_flat_dev(const char *path, int shard_rank, int shard_world, char **d_flat, uint64_t *flat_len, uint64_t *flat_cap,
                         uint64_t *n_records, char **text, uint64_t *text_len);
// every gzip input of a call through gz_fastq_to_flat_dev, one host thread per file: `flat(i, d_flat, len, cap, n_records)`
// takes over the device buffer of input i (called from that file's thread; returns an SS_* code); inputs that were
// only inflated come back as host texts in `texts`; done[i] = 1 for the inputs that need nothing more
int gz_inputs_on_device(const char *const *paths, int n_paths, int shard_rank, int shard_world,
                        const std::function<int(int, char *, uint64_t, uint64_t, uint64_t)> &flat, std::vector<InflatedText> &texts,
                        std::vector<char> &done);
int scan_file_parallel(ss_db *db, const char *path, uint64_t *n_records, uint64_t *n_bases, bool *handled, int shard_rank = 0,
                       int shard_world = 1);
int scan_text_parallel(ss_db *db, const char *text, uint64_t n, uint64_t *n_records, uint64_t *n_bases, bool *handled,
                       int shard_rank = 0, int shard_world = 1);
int launch_scan_mini(ss_db *db, const void *bases_dev, uint64_t n, hipStream_t stream, unsigned blocks,
                     uint64_t n_tiles, bool binned = false, uint64_t set_id = 0);
int launch_scan_mini_multi(ss_db *const *dbs, int n_dbs, const void *bases_dev, uint64_t n, hipStream_t stream, bool binned);
// Layer 2 (ss_l2.hip, ss_enet.hip) works on the CALLING THREAD's own stream and takes its temporaries from the stream-ordered
// pool (round 5): the clusters of a sample are solved on several host threads at once, and on the legacy default stream every
// synchronous copy of one thread waited for the kernels of all the others, every hipFree for the whole device (four 5 M-row
// clusters: 36 ms each in the O(K) vector pass that takes 5.5 ms alone).  Buffers that live in a handle stay with hipMalloc.
// the stream-ordered pool keeps at least `bytes` of what its users free (raised, never lowered: ss_host.hip)
void pool_keep_at_least(uint64_t bytes);
namespace l2s {
inline hipStream_t stream() { return hipStreamPerThread; }
hipError_t dmalloc(void **p, size_t n);          // (ss_host.hip: sets the pool's release threshold once)
inline hipError_t dfree(void *p) { return p ? hipFreeAsync(p, stream()) : hipSuccess; }
inline hipError_t copy(void *d, const void *s_, size_t n, hipMemcpyKind k)
{
    const hipError_t e = hipMemcpyAsync(d, s_, n, k, stream());
    return e != hipSuccess ? e : hipStreamSynchronize(stream());
}
inline hipError_t set(void *d, int v, size_t n) { return hipMemsetAsync(d, v, n, stream()); }
inline hipError_t sync() { return hipStreamSynchronize(stream()); }
}  // namespace l2s
// Large device blocks a load is done with (the text of a big .gz, the file-order slab that binning has replaced) are kept for
// the next large request of the same process instead of going back to the driver: a fresh process is handed device memory at
// ~25 GB/s and a 50 M-read .gz pair asked for ~47 GB of it, a third of that for blocks that replace each other (ss_host.hip).
// big_take: a kept block of at least `bytes` (and at most 2.5 x), or nullptr; the caller owns it and gives it back with
// big_put or hipFree.  big_put keeps blocks of 256 MB and more, at most three and 24 GB, and frees what it does not keep.
void *big_take(uint64_t bytes, uint64_t *cap = nullptr);      // *cap: what the block really holds (the caller keeps that, for the next big_put)
void big_put(void *p, uint64_t cap);
void big_release();                                      // everything kept goes back to the driver (ss_gz_gpu_release)
hipError_t big_malloc(void **p, uint64_t bytes, uint64_t *got);      // big_take, else hipMalloc (which, failing, is tried again after big_release); *got >= bytes
// a whole file into device memory through the pinned upload buffers of the .gz path (ss_ginflate.hip)
bool upload_file_to_device(int fd, uint64_t n, uint8_t *d_dst);
// ss_scan_flat_dev for a block whose records ss_reorder.hip has binned by locus (the scan may add hits up in LDS first)
// (set_id: ss_reads::serial of the resident set the block belongs to, 0 = none)
int scan_flat_dev(ss_db *db, const void *bases_dev, uint64_t n, void *stream, bool binned, uint64_t set_id = 0);
}  // namespace ss
