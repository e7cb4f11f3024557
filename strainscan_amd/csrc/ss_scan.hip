// ss_scan.hip -- device k-mer table + the read-vs-database scan for gfx950 (MI355X).
//
// Replaces the `jellyfish count -m k --if <fasta> ... ; jellyfish dump -c` pair the reference
// spawns at library/identify.py:82-87, identify_low_mem.py:73-75, identify_low_depth.py:53-59
// and Vote_Strain_L2_Lasso_new_sp.py:359-372, and the Python tail that maps the dumped k-mers
// back to rows of kmer.fa (identify.py:90-101).
//
// Data layout in HBM
//   d_keys[capacity]   u64  open-address table (linear probing, load <= 0.5), EMPTY = ~0
//   d_counts[capacity] u32  occurrences per slot; scans only ever atomicAdd into it
//   d_slot_of_row[n]   u32  slot that owns kmer.fa row i (SS_NO_SLOT if the row has no k-mer)
//   d_row_valid[n]     u8   1 iff row i is a key of the reference's match_results
// Keeping the counters slot-indexed means a hit costs one atomic on the line next to the key
// that was just read and no row-index lookup; rows are resolved once per scan by a gather.
//
// The scan kernel is HBM/latency bound (random 8-byte gathers into a table far larger than
// L2): 150 B of bases + 120 probes x 8 B per 150-bp read and ~40 VALU ops per probe.
#include "ss_common.h"
#include "ss_scan_dev.h"

#include <stdlib.h>

#include <algorithm>
#include <vector>

namespace {

using namespace ss::dev;
constexpr uint64_t STAGE_BYTES = 32ull << 20;  // pinned staging chunk for host-resident blocks

// ---------------------------------------------------------------------------------------------
// encode + probe + count.  One tile = 4096 consecutive start positions of the flat base stream
// (+30 bytes of halo).  Phase 1: coalesced 16-byte loads, SWAR 2-bit encode, codes to LDS.
// Phase 2: each lane rebuilds its 16 overlapping k-mers from three LDS dwords with funnel
// shifts, issues the 16 first-probe loads back to back (16 independent HBM gathers in flight
// per lane), then resolves them; only the rare collision chains loop.
// ---------------------------------------------------------------------------------------------
// BLOOM: a one-probe Bloom filter over the table's k-mers that fits one XCD's L2 (<= 2^25 bits; built for tables of up to
// ~4 M k-mers, i.e. cluster tables -- the flat layout serves `-k` other than 31 for them): bit = the TOP bits of the same
// mix64 whose next bits address the table, so a lookup costs one more shift.  ~95 % of a sample's k-mers are not in the
// table and stop at the filter (L2) instead of pulling a random sector of the table (MALL / HBM): 12.5 -> 3.x ms for
// 4 M reads against a 2 M-row table (scripts/bench_k.py).
template <bool ALIGNED, bool BLOOM>
__global__ __launch_bounds__(SCAN_THREADS) void scan_kernel(
    const uint8_t *__restrict__ bases, uint64_t n, uint64_t n_tiles,
    const uint64_t *__restrict__ keys, uint32_t *__restrict__ counts, uint32_t log2cap, int k,
    const uint32_t *__restrict__ bloom, uint32_t bloom_bits)
{
    __shared__ uint32_t s_code[2][SCAN_THREADS + 2];
    __shared__ uint16_t s_inv[2][SCAN_THREADS + 2];

    const int t = threadIdx.x;
    const uint64_t kmask = (~0ull) >> (64 - 2 * k);
    const uint64_t wmask = (1ull << k) - 1;
    const uint32_t smask = (uint32_t)((1ull << log2cap) - 1);
    int buf = 0;

    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        const uint64_t base = tile * (uint64_t)TILE;
        {
            uint32_t w[4], code, inv;
            load16<ALIGNED>(bases, base + (uint64_t)t * 16, n, w);
            encode16(w, code, inv);
            s_code[buf][t] = code;
            s_inv[buf][t] = (uint16_t)inv;
            if (t < 2) {  // halo: the last k-mers of the tile reach k-1 bytes past it
                load16<ALIGNED>(bases, base + TILE + (uint64_t)t * 16, n, w);
                encode16(w, code, inv);
                s_code[buf][SCAN_THREADS + t] = code;
                s_inv[buf][SCAN_THREADS + t] = (uint16_t)inv;
            }
        }
        __syncthreads();  // one barrier per tile: the next tile writes the other LDS buffer

        const uint64_t lo = (uint64_t)s_code[buf][t] | ((uint64_t)s_code[buf][t + 1] << 32);
        const uint64_t hi = (uint64_t)s_code[buf][t + 2];
        const uint64_t inv = (uint64_t)s_inv[buf][t] | ((uint64_t)s_inv[buf][t + 1] << 16) |
                             ((uint64_t)s_inv[buf][t + 2] << 32);

        uint64_t key[PPT], got[PPT];
        uint32_t slot[PPT];
        uint32_t live = 0;
        if (BLOOM) {
            uint32_t bw[PPT], bb[PPT];
#pragma unroll
            for (int j = 0; j < PPT; j++) {
                uint64_t km = (j == 0) ? lo : ((lo >> (2 * j)) | (hi << (64 - 2 * j)));
                key[j] = km & kmask;
                const uint64_t m = ss::mix64(key[j]);
                slot[j] = (uint32_t)(m >> (64 - log2cap));
                bb[j] = ss::bloom_bit_of(m, bloom_bits);
                if (((inv >> j) & wmask) == 0) live |= 1u << j;
            }
#pragma unroll
            for (int j = 0; j < PPT; j++) bw[j] = ((live >> j) & 1u) ? bloom[bb[j] >> 5] : 0u;
#pragma unroll
            for (int j = 0; j < PPT; j++) if (!((bw[j] >> (bb[j] & 31u)) & 1u)) live &= ~(1u << j);
        } else {
#pragma unroll
            for (int j = 0; j < PPT; j++) {
                uint64_t km = (j == 0) ? lo : ((lo >> (2 * j)) | (hi << (64 - 2 * j)));
                key[j] = km & kmask;
                slot[j] = ss::slot_of(key[j], log2cap);
                if (((inv >> j) & wmask) == 0) live |= 1u << j;
            }
        }
#pragma unroll
        for (int j = 0; j < PPT; j++)
            got[j] = ((live >> j) & 1u) ? keys[slot[j]] : ss::EMPTY_KEY;
#pragma unroll
        for (int j = 0; j < PPT; j++) {
            uint64_t g = got[j];
            uint32_t s = slot[j];
            while (g != key[j] && g != ss::EMPTY_KEY) {
                s = (s + 1) & smask;
                g = keys[s];
            }
            if (g == key[j]) atomicAdd(&counts[s], 1u);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// table build
// ---------------------------------------------------------------------------------------------
__global__ void build_insert_kernel(const uint64_t *__restrict__ in_keys, const uint8_t *__restrict__ flags,
                                    uint64_t n_rows, int upper_keys, uint64_t *keys, uint32_t *last_row,
                                    uint32_t *slot_of_row, uint32_t log2cap, unsigned long long *n_distinct)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const uint8_t f = flags[i];
    if (!(f & SS_ROW_VALID)) { slot_of_row[i] = SS_NO_SLOT; return; }
    const uint64_t key = in_keys[i];
    const uint32_t smask = (uint32_t)((1ull << log2cap) - 1);
    uint32_t s = ss::slot_of(key, log2cap);
    for (;;) {
        unsigned long long old = atomicCAS((unsigned long long *)&keys[s], (unsigned long long)ss::EMPTY_KEY,
                                           (unsigned long long)key);
        if (old == ss::EMPTY_KEY) { atomicAdd(n_distinct, 1ull); break; }
        if (old == key) break;
        s = (s + 1) & smask;
    }
    slot_of_row[i] = s;
    // dict overwrite at identify.py:94: the LAST row with this text owns the count.  With raw
    // (non-upper) keys a lower-case row can never equal jellyfish's upper-case dump.
    if (upper_keys == 1 || !(f & SS_ROW_LOWER)) atomicMax(&last_row[s], (uint32_t)(i + 1));
}

__global__ void build_bloom_kernel(const uint64_t *__restrict__ keys, uint64_t capacity, uint32_t bloom_bits, uint32_t *bloom)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= capacity) return;
    const uint64_t key = keys[s];
    if (key == ss::EMPTY_KEY) return;
    const uint32_t b = ss::bloom_bit_of(ss::mix64(key), bloom_bits);
    atomicOr(&bloom[b >> 5], 1u << (b & 31u));
}

__global__ void build_finalize_kernel(const uint32_t *__restrict__ slot_of_row, const uint32_t *__restrict__ last_row,
                                      uint64_t n_rows, uint8_t *row_valid)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t s = slot_of_row[i];
    row_valid[i] = (s != SS_NO_SLOT && last_row[s] == (uint32_t)(i + 1)) ? 1 : 0;
}

__global__ void build_orphans_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ last_row,
                                     uint64_t capacity, unsigned long long *n_orphans)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= capacity) return;
    if (keys[s] != ss::EMPTY_KEY && last_row[s] == 0) atomicAdd(n_orphans, 1ull);
}

__global__ void gather_rows_kernel(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ slot_of_row,
                                   const uint8_t *__restrict__ row_valid, uint64_t n_rows, uint32_t *out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows;
         i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = row_valid[i] ? counts[slot_of_row[i]] : 0u;
}

__global__ void scatter_rows_kernel(const uint32_t *__restrict__ in, const uint32_t *__restrict__ slot_of_row,
                                    const uint8_t *__restrict__ row_valid, uint64_t n_rows, uint32_t *counts)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows;
         i += (uint64_t)gridDim.x * blockDim.x)
        if (row_valid[i]) counts[slot_of_row[i]] = in[i];   // one valid row per slot: no race
}

int cu_count()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
            cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

int ensure_staging(ss_db *db)
{
    if (db->stage_bytes) return SS_OK;
    for (int i = 0; i < 2; i++) {
        SS_HIP(hipHostMalloc((void **)&db->h_stage[i], STAGE_BYTES, hipHostMallocDefault));
        SS_HIP(hipMalloc((void **)&db->d_stage[i], STAGE_BYTES));
        SS_HIP(hipStreamCreateWithFlags(&db->streams[i], hipStreamNonBlocking));
        SS_HIP(hipEventCreateWithFlags(&db->stage_free[i], hipEventDisableTiming));
    }
    db->stage_bytes = STAGE_BYTES;
    db->device_bytes += 2 * STAGE_BYTES;
    return SS_OK;
}

}  // namespace

extern "C" {

int ss_db_build(const uint64_t *keys, const uint8_t *flags, uint64_t n_rows, int k, int upper_keys,
                ss_db **out)
{
    if (!out || (n_rows && (!keys || !flags))) return SS_EINVAL;
    if (k < 1 || k > 31) return SS_ERANGE;
    if (n_rows >= 0xFFFFFFFEull) return SS_ERANGE;
    uint64_t n_valid = 0;
    for (uint64_t i = 0; i < n_rows; i++) n_valid += (flags[i] & SS_ROW_VALID) ? 1 : 0;
    uint32_t log2cap = 10;
    while ((1ull << log2cap) < 2 * n_valid) log2cap++;
    if (log2cap > 31) return SS_ERANGE;

    ss_db *db = new (std::nothrow) ss_db();
    if (!db) return SS_ENOMEM;
    db->k = k;
    db->n_rows = n_rows;
    db->log2cap = log2cap;
    db->capacity = 1ull << log2cap;
    db->n_slots = db->capacity;
    if (hipGetDevice(&db->device) != hipSuccess) { delete db; return SS_ENODEV; }
    // layout: minimizer pages for 17 <= k <= 31 (k = 31, the tree scan and the default layer-2 k, through the tuned kernel;
    // the other k through the one-lane-per-position kernel, ss_mini.hip scan_minik_kernel), flat table below that;
    // SS_LAYOUT=flat overrides for A/B measurements
    const char *lay = getenv("SS_LAYOUT");
    db->layout = (k >= ss::MINI_K_MIN && k <= 31) ? 1 : 0;
    if (lay && !strcmp(lay, "flat")) db->layout = 0;
    if (db->layout == 1) {
        int mrc = ss::build_mini(db, keys, flags, n_rows, upper_keys);
        if (mrc != SS_OK) { ss_db_destroy(db); return mrc; }
        *out = db;
        return SS_OK;
    }

    uint64_t *d_in = nullptr;
    uint8_t *d_flags = nullptr;
    uint32_t *d_last = nullptr;
    unsigned long long *d_ctr = nullptr;
    const uint64_t nr = n_rows ? n_rows : 1;
    int rc = SS_OK;
    auto fail = [&](int code) {
        hipFree(d_in); hipFree(d_flags); hipFree(d_last); hipFree(d_ctr);
        ss_db_destroy(db);
        return code;
    };
#define SS_TRY(call)                                                                     \
    do {                                                                                 \
        hipError_t _e = (call);                                                          \
        if (_e != hipSuccess) {                                                          \
            ss::set_last_error(#call, __FILE__, __LINE__, _e);                           \
            return fail(_e == hipErrorOutOfMemory ? SS_ENOMEM : SS_EHIP);                \
        }                                                                                \
    } while (0)
    SS_TRY(hipMalloc((void **)&db->d_keys, db->capacity * sizeof(uint64_t)));
    SS_TRY(hipMalloc((void **)&db->d_counts, db->capacity * sizeof(uint32_t)));
    SS_TRY(hipMalloc((void **)&db->d_slot_of_row, nr * sizeof(uint32_t)));
    SS_TRY(hipMalloc((void **)&db->d_row_valid, nr));
    SS_TRY(hipMalloc((void **)&d_in, nr * sizeof(uint64_t)));
    SS_TRY(hipMalloc((void **)&d_flags, nr));
    SS_TRY(hipMalloc((void **)&d_last, db->capacity * sizeof(uint32_t)));
    SS_TRY(hipMalloc((void **)&d_ctr, 2 * sizeof(unsigned long long)));
    db->device_bytes = db->capacity * 12 + nr * 5;
    SS_TRY(hipMemset(db->d_keys, 0xFF, db->capacity * sizeof(uint64_t)));
    SS_TRY(hipMemset(db->d_counts, 0, db->capacity * sizeof(uint32_t)));
    SS_TRY(hipMemset(d_last, 0, db->capacity * sizeof(uint32_t)));
    SS_TRY(hipMemset(d_ctr, 0, 2 * sizeof(unsigned long long)));
    if (n_rows) {
        SS_TRY(hipMemcpy(d_in, keys, n_rows * sizeof(uint64_t), hipMemcpyHostToDevice));
        SS_TRY(hipMemcpy(d_flags, flags, n_rows, hipMemcpyHostToDevice));
        const unsigned blocks = (unsigned)((n_rows + 255) / 256);
        hipLaunchKernelGGL(build_insert_kernel, dim3(blocks), dim3(256), 0, 0, d_in, d_flags, n_rows, upper_keys,
                           db->d_keys, d_last, db->d_slot_of_row, log2cap, d_ctr);
        hipLaunchKernelGGL(build_finalize_kernel, dim3(blocks), dim3(256), 0, 0, db->d_slot_of_row, d_last, n_rows,
                           db->d_row_valid);
        const unsigned cblocks = (unsigned)((db->capacity + 255) / 256);
        hipLaunchKernelGGL(build_orphans_kernel, dim3(cblocks), dim3(256), 0, 0, db->d_keys, d_last, db->capacity,
                           d_ctr + 1);
        SS_TRY(hipGetLastError());
    }
    unsigned long long ctr[2] = {0, 0};
    SS_TRY(hipMemcpy(ctr, d_ctr, sizeof(ctr), hipMemcpyDeviceToHost));
    db->n_distinct = ctr[0];
    {
        // Bloom filter over the k-mers when it fits an XCD's L2 with >= 8 bits per k-mer (cluster tables)
        int bits = 12;
        while (bits < 25 && (1ull << bits) < 16 * db->n_distinct) bits++;
        if ((1ull << bits) < 8 * db->n_distinct) bits = 0;
        if (bits >= 10 && bits <= 30 && db->n_distinct) {
            SS_TRY(hipMalloc((void **)&db->d_bloom, (1ull << bits) / 8));
            SS_TRY(hipMemset(db->d_bloom, 0, (1ull << bits) / 8));
            hipLaunchKernelGGL(build_bloom_kernel, dim3((unsigned)((db->capacity + 255) / 256)), dim3(256), 0, 0, db->d_keys, db->capacity,
                               (uint32_t)bits, db->d_bloom);
            SS_TRY(hipDeviceSynchronize());
            db->bloom_bits = (uint32_t)bits;
            db->device_bytes += (1ull << bits) / 8;
        }
    }
    hipFree(d_in); hipFree(d_flags); hipFree(d_last); hipFree(d_ctr);
    d_in = nullptr; d_flags = nullptr; d_last = nullptr; d_ctr = nullptr;
#undef SS_TRY
    if (ctr[1] != 0 && upper_keys == 0) {  // a dumped k-mer no row can own: KeyError in the reference
        ss_db_destroy(db);
        return SS_EKEY;
    }
    (void)rc;
    *out = db;
    return SS_OK;
}

int ss_db_destroy(ss_db *db)
{
    if (!db) return SS_OK;
    hipFree(db->d_keys);
    hipFree(db->d_mkeys);
    hipFree(db->d_dir);
    hipFree(db->d_bloom);
    hipFree(db->d_counts);
    hipFree(db->d_slot_of_row);
    hipFree(db->d_row_valid);
    ss_db::free_workers(db->workers, ss_db::MAX_WORKERS);
    for (int i = 0; i < 2; i++) {
        if (db->h_stage[i]) hipHostFree(db->h_stage[i]);
        if (db->d_stage[i]) hipFree(db->d_stage[i]);
        if (db->streams[i]) hipStreamDestroy(db->streams[i]);
        if (db->stage_free[i]) hipEventDestroy(db->stage_free[i]);
    }
    delete db;
    return SS_OK;
}

int ss_db_info(const ss_db *db, uint64_t *n_rows, uint64_t *n_distinct, uint64_t *capacity, int *k)
{
    if (!db) return SS_EINVAL;
    if (n_rows) *n_rows = db->n_rows;
    if (n_distinct) *n_distinct = db->n_distinct;
    if (capacity) *capacity = db->capacity;
    if (k) *k = db->k;
    return SS_OK;
}

int ss_db_row_valid(const ss_db *db, uint8_t *row_valid)
{
    if (!db || !row_valid) return SS_EINVAL;
    if (db->n_rows) SS_HIP(hipMemcpy(row_valid, db->d_row_valid, db->n_rows, hipMemcpyDeviceToHost));
    return SS_OK;
}

const uint8_t *ss_db_row_valid_dev(const ss_db *db) { return db ? db->d_row_valid : nullptr; }
uint64_t ss_db_device_bytes(const ss_db *db) { return db ? db->device_bytes : 0; }
int ss_db_index_info(const ss_db *db, uint64_t out[8])
{
    if (!db || !out) return SS_EINVAL;
    out[0] = (uint64_t)db->layout; out[1] = db->n_slots; out[2] = db->n_buckets; out[3] = db->n_dir;
    out[4] = db->d_bloom ? db->bloom_bits : 0; out[5] = db->n_distinct; out[6] = db->n_mslots; out[7] = db->n_inline;
    return SS_OK;
}
uint64_t ss_scan_kernel_launches(const ss_db *db) { return db ? db->launches.load() : (uint64_t)0; }

int ss_scan_reset(ss_db *db, void *stream)
{
    if (!db) return SS_EINVAL;
    SS_HIP(hipMemsetAsync(db->d_counts, 0, db->n_slots * sizeof(uint32_t), ss::as_stream(stream)));
    return SS_OK;
}

int ss_db_expect_hits(ss_db *db, int expect)
{
    if (!db) return SS_EINVAL;
    db->expect_hits = expect != 0;
    return SS_OK;
}

int ss_db_probe_info(const ss_db *db, uint64_t out[3])
{
    if (!db || !out) return SS_EINVAL;
    out[0] = db->probe_set;
    out[1] = db->probe_comb ? 1 : 0;
    out[2] = (uint64_t)(db->probe_runs_per_tile * 1000.0 + 0.5);
    return SS_OK;
}

int ss_scan_flat_dev(ss_db *db, const void *bases_dev, uint64_t n, void *stream) { return ss::scan_flat_dev(db, bases_dev, n, stream, false); }

}  // extern "C"

int ss::scan_flat_dev(ss_db *db, const void *bases_dev, uint64_t n, void *stream, bool binned, uint64_t set_id)
{
    if (!db || (n && !bases_dev)) return SS_EINVAL;
    if (n < (uint64_t)db->k) return SS_OK;
    const uint64_t n_tiles = (n + TILE - 1) / TILE;
    const uint64_t max_blocks = (uint64_t)cu_count() * 8;
    const unsigned blocks = (unsigned)std::min<uint64_t>(n_tiles, max_blocks);
    if (db->layout == 1) {
        int rc = ss::launch_scan_mini(db, bases_dev, n, ss::as_stream(stream), blocks, n_tiles, binned, set_id);
        if (rc == SS_OK) db->launches++;
        return rc;
    }
    const bool aligned = (((uintptr_t)bases_dev) & 15) == 0;
#define SS_FLAT(A, B) hipLaunchKernelGGL((scan_kernel<A, B>), dim3(blocks), dim3(SCAN_THREADS), 0, ss::as_stream(stream), (const uint8_t *)bases_dev, n, \
                                         n_tiles, db->d_keys, db->d_counts, db->log2cap, db->k, db->d_bloom, db->bloom_bits)
    if (db->d_bloom) { if (aligned) SS_FLAT(true, true); else SS_FLAT(false, true); }
    else             { if (aligned) SS_FLAT(true, false); else SS_FLAT(false, false); }
#undef SS_FLAT
    SS_HIP(hipGetLastError());
    db->launches++;
    return SS_OK;
}

extern "C" {

int ss_scan_flat_host(ss_db *db, const char *bases, uint64_t n)
{
    if (!db || (n && !bases)) return SS_EINVAL;
    int rc = ensure_staging(db);
    if (rc) return rc;
    SS_HIP(hipStreamSynchronize(nullptr));      // (the staging streams are non-blocking: a reset pending on the default stream, see ss_scan_files_shard)
    // Chunks overlap by k-1 bytes: a k-mer starting in the last k-1 bytes of chunk A is invalid
    // there (it runs past the end) and is counted exactly once in chunk B.
    const uint64_t ov = (uint64_t)db->k - 1;
    uint64_t pos = 0;
    int b = 0;
    bool used[2] = {false, false};
    while (pos < n) {
        const uint64_t len = std::min<uint64_t>(db->stage_bytes, n - pos);
        if (used[b]) SS_HIP(hipEventSynchronize(db->stage_free[b]));
        memcpy(db->h_stage[b], bases + pos, len);
        SS_HIP(hipMemcpyAsync(db->d_stage[b], db->h_stage[b], len, hipMemcpyHostToDevice, db->streams[b]));
        rc = ss_scan_flat_dev(db, db->d_stage[b], len, db->streams[b]);
        if (rc) return rc;
        SS_HIP(hipEventRecord(db->stage_free[b], db->streams[b]));
        used[b] = true;
        if (pos + len >= n) break;
        pos += len - ov;
        b ^= 1;
    }
    for (int i = 0; i < 2; i++)
        if (used[i]) SS_HIP(hipStreamSynchronize(db->streams[i]));
    return SS_OK;
}

// shard_world > 1: the reader still walks the whole input (a record grammar has no entry points), but only every
// shard_world-th block is copied and scanned
static int scan_files_sequential(ss_db *db, const char *const *paths, int n_paths, uint64_t *n_records,
                                 uint64_t *n_bases, int shard_rank = 0, int shard_world = 1)
{
    int rc = ensure_staging(db);
    if (rc) return rc;
    ss_reader *rd = nullptr;
    rc = ss_reader_open(paths, n_paths, &rd);
    if (rc) return rc;
    ss_reader_set_overlap(rd, db->k - 1);
    uint64_t recs = 0, total = 0, blk = 0;
    int b = 0;
    bool used[2] = {false, false};
    for (;;) {
        if (used[b]) {
            hipError_t e = hipEventSynchronize(db->stage_free[b]);
            if (e != hipSuccess) { ss_reader_close(rd); ss::set_last_error("hipEventSynchronize", __FILE__, __LINE__, e); return SS_EHIP; }
        }
        uint64_t len = 0, nr = 0;
        rc = ss_reader_next(rd, db->h_stage[b], db->stage_bytes, &len, &nr);
        if (rc) { ss_reader_close(rd); return rc; }
        if (len == 0) break;
        recs += nr;
        total += len;
        if ((int)(blk++ % (uint64_t)shard_world) != shard_rank) continue;
        hipError_t e = hipMemcpyAsync(db->d_stage[b], db->h_stage[b], len, hipMemcpyHostToDevice, db->streams[b]);
        if (e != hipSuccess) { ss_reader_close(rd); ss::set_last_error("hipMemcpyAsync", __FILE__, __LINE__, e); return SS_EHIP; }
        rc = ss_scan_flat_dev(db, db->d_stage[b], len, db->streams[b]);
        if (rc) { ss_reader_close(rd); return rc; }
        hipEventRecord(db->stage_free[b], db->streams[b]);
        used[b] = true;
        b ^= 1;
    }
    ss_reader_close(rd);
    for (int i = 0; i < 2; i++)
        if (used[i]) SS_HIP(hipStreamSynchronize(db->streams[i]));
    *n_records += recs;
    *n_bases += total;
    return SS_OK;
}

int ss_scan_files(ss_db *db, const char *const *paths, int n_paths, uint64_t *n_records, uint64_t *n_bases)
{
    return ss_scan_files_shard(db, paths, n_paths, 0, 1, n_records, n_bases);
}

int ss_scan_files_shard(ss_db *db, const char *const *paths, int n_paths, int shard_rank, int shard_world, uint64_t *n_records,
                        uint64_t *n_bases)
{
    if (!db || !paths || n_paths < 1 || shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world) return SS_EINVAL;
    // The chunks are scanned on the parse workers' streams, which are NON-BLOCKING: nothing orders them behind what the caller
    // put on the default stream before this call -- ss_scan_reset's memset above all (ss_scan_reset(db, NULL) only enqueues
    // it).  On a busy device (three ranks sharing one GPU in the tests) the memset was seen to run AFTER the first chunk's
    // scan, once in ~30 runs: a small file is one chunk, one rank's whole share, and all its counts were gone.
    SS_HIP(hipStreamSynchronize(nullptr));
    uint64_t recs = 0, total = 0;
    const char *seq_env = getenv("SS_INGEST");
    const bool allow_parallel = !(seq_env && !strcmp(seq_env, "sequential"));
    for (int i = 0; i < n_paths; i++)
        if (!paths[i]) return SS_EINVAL;
    // .gz inputs are inflated whole (libdeflate, all files at once) and parsed like plain text when possible
    std::vector<ss::InflatedText> texts;
    std::vector<char> on_device((size_t)n_paths, 0);
    int rc = SS_OK;
    if (allow_parallel) {
        // (not SS_GZ_GPU=0) inflated and reduced to the sequence lines on the device (ss_ginflate.hip, ss_fastq_dev.hip),
        // scanned from there
        std::vector<const char *> rest(paths, paths + n_paths);
        {
            std::mutex mu;
            rc = ss::gz_inputs_on_device(paths, n_paths, shard_rank, shard_world, [&](int, char *d, uint64_t len, uint64_t, uint64_t nrec) {
                std::lock_guard<std::mutex> g(mu);                    // one scan at a time on the table's stream
                int r = ss_scan_flat_dev(db, d, ss_reads::padded(len), nullptr);
                if (r == SS_OK && hipStreamSynchronize(nullptr) != hipSuccess) r = SS_EHIP;
                hipFree(d);
                recs += nrec;
                total += len;
                return r;
            }, texts, on_device);
            for (int i = 0; i < n_paths; i++)
                if (on_device[i] || texts[i].p) rest[i] = "";
        }
        if (rc == SS_OK) {
            std::vector<ss::InflatedText> more = ss::inflate_gz_inputs(rest.data(), n_paths);
            if (texts.empty()) texts = more;
            else for (int i = 0; i < n_paths; i++) if (more[i].p) texts[i] = more[i];
        }
    }
    for (int i = 0; i < n_paths && rc == SS_OK; i++) {
        if (!paths[i][0] || on_device[i]) continue;            // '' = no second file (StrainScan.py:182)
        bool handled = false;
        if (allow_parallel) {
            if (!texts.empty() && texts[i].p) rc = ss::scan_text_parallel(db, texts[i].p, texts[i].n, &recs, &total, &handled, shard_rank, shard_world);
            else rc = ss::scan_file_parallel(db, paths[i], &recs, &total, &handled, shard_rank, shard_world);
        }
        if (!texts.empty()) { ss::free_later(texts[i].p); texts[i].p = nullptr; }
        if (rc == SS_OK && !handled) rc = scan_files_sequential(db, &paths[i], 1, &recs, &total, shard_rank, shard_world);
    }
    for (auto &tx : texts) free(tx.p);
    if (rc) return rc;
    if (n_records) *n_records = recs;
    if (n_bases) *n_bases = total;
    return SS_OK;
}

int ss_counts_rows_dev(const ss_db *db, uint32_t *counts_rows_dev, void *stream)
{
    if (!db || (db->n_rows && !counts_rows_dev)) return SS_EINVAL;
    if (!db->n_rows) return SS_OK;
    const unsigned blocks = (unsigned)std::min<uint64_t>((db->n_rows + 255) / 256, (uint64_t)cu_count() * 16);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, ss::as_stream(stream), db->d_counts,
                       db->d_slot_of_row, db->d_row_valid, db->n_rows, counts_rows_dev);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int ss_counts_load_rows_dev(ss_db *db, const uint32_t *counts_rows_dev, void *stream)
{
    if (!db || (db->n_rows && !counts_rows_dev)) return SS_EINVAL;
    if (!db->n_rows) return SS_OK;
    const unsigned blocks = (unsigned)std::min<uint64_t>((db->n_rows + 255) / 256, (uint64_t)cu_count() * 16);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(blocks), dim3(256), 0, ss::as_stream(stream), counts_rows_dev,
                       db->d_slot_of_row, db->d_row_valid, db->n_rows, db->d_counts);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int ss_counts_rows(const ss_db *db, uint32_t *counts_rows)
{
    if (!db || (db->n_rows && !counts_rows)) return SS_EINVAL;
    if (!db->n_rows) return SS_OK;
    uint32_t *d = nullptr;
    SS_HIP(hipMalloc((void **)&d, db->n_rows * sizeof(uint32_t)));
    int rc = ss_counts_rows_dev(db, d, nullptr);
    if (rc == SS_OK) {
        hipError_t e = hipMemcpy(counts_rows, d, db->n_rows * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { ss::set_last_error("hipMemcpy", __FILE__, __LINE__, e); rc = SS_EHIP; }
    }
    hipFree(d);
    return rc;
}

}  // extern "C"
