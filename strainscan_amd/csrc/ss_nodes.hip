// ss_nodes.hip -- per-node reductions of the cluster-search-tree walk, all nodes in one launch.
//
// Replaces match_node + del_outlier (library/identify.py:106-127; the low-depth variant
// identify_low_depth.py:77-101 differs only by a host-side minimum on `length`), which the
// reference runs in Python per visited node after re-reading <db>/kmers/<id>.
//
// One 1024-thread workgroup per node; a node list is <= 30000 rows (StrainScan_build.py:77-78).
// Pass 0 gathers row -> valid flag -> count once (four independent gathers in flight per lane) and
// compacts the POSITIVE counts into LDS; np.median (order statistics: an exact 4 x 8-bit radix
// select over LDS histograms, no sort), the second middle element of an even-length profile and
// the outlier cut (c >= 100 * median dropped, in integers: c >= 50 * (v1 + v2)) then run from LDS.
// Most nodes of a sample have no hits and end after pass 0; the few that do used to pay seven
// passes of dependent HBM gathers each (the whole launch waited ~300 us for the largest of them).
// Positive counts are kept as 16-bit values (32768 of them); a node with more, or with a count >= 65535,
// falls back to those passes over global memory.
// Algorithmic bytes: 4 B row index + 4 B count + 1 B valid flag per node k-mer.
#include "ss_common.h"

#include <algorithm>
#include <vector>

struct ss_nodes {
    uint32_t n_nodes = 0;
    uint64_t n_rows_total = 0;
    uint32_t *d_rows = nullptr;
    uint64_t *d_offsets = nullptr;
    ss_node_stat *d_stats = nullptr;
    uint32_t *d_counts_rows = nullptr;  // scratch for the single-GPU convenience entry point
    uint64_t counts_rows_cap = 0;
};

namespace {

constexpr int NT = 1024, NW = NT / 64;
#ifndef SS_NODE_CAP
#define SS_NODE_CAP 32768            // positive counts of one node kept in LDS as 16-bit values (64 KB: two workgroups
                                     // per CU); the reference builds nodes of <= 30000 rows (StrainScan_build.py:77-78)
#endif
constexpr uint32_t CAP = SS_NODE_CAP;

__device__ __forceinline__ uint64_t block_sum(uint64_t v, uint64_t *s_red)
{
    // wave64 shuffle reduction, then NW partials through LDS
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    uint64_t r = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) r += s_red[w];
    return r;
}

__device__ __forceinline__ uint32_t block_min(uint32_t v, uint32_t *s_red)
{
    for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    uint32_t r = s_red[0];
#pragma unroll
    for (int w = 1; w < NW; w++) r = min(r, s_red[w]);
    return r;
}

__global__ __launch_bounds__(NT) void node_reduce_kernel(const uint32_t *__restrict__ rows,
                                                         const uint64_t *__restrict__ offsets,
                                                         const uint32_t *__restrict__ counts,
                                                         const uint8_t *__restrict__ valid,
                                                         ss_node_stat *__restrict__ stats)
{
    __shared__ uint16_t s_val[CAP];
    __shared__ uint32_t s_hist[256];
    __shared__ uint64_t s_red64[NW];
    __shared__ uint32_t s_red32[NW];
    __shared__ uint32_t s_pick[2];  // [0] = chosen bin, [1] = rank inside it
    __shared__ uint32_t s_n;

    const uint32_t node = blockIdx.x;
    const uint64_t lo = offsets[node], hi = offsets[node + 1];
    const int t = threadIdx.x;
    if (t == 0) s_n = 0;
    __syncthreads();

    // a positive count goes to LDS: one LDS atomic per wave (ballot + lane rank), order is irrelevant
    auto keep = [&](bool pos, uint32_t c) {
        const uint64_t m = __ballot(pos);
        if (!m) return;
        uint32_t base = 0;
        const uint32_t lane = (uint32_t)t & 63u;
        if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(&s_n, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1, 64);
        const uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (pos && idx < CAP) s_val[idx] = (uint16_t)min(c, 0xFFFFu);
    };
    uint32_t cmax = 0;                       // largest positive count: radix passes above its top byte are skipped

    // pass 0: length (valid rows), n_pos (valid rows seen at least once), positive counts -> LDS
    uint64_t len = 0;
    {
        uint64_t i = lo + t;
        for (; i + 3 * NT < hi; i += 4 * NT) {
            const uint32_t r0 = rows[i], r1 = rows[i + NT], r2 = rows[i + 2 * NT], r3 = rows[i + 3 * NT];
            const uint32_t v0 = valid[r0], v1 = valid[r1], v2 = valid[r2], v3 = valid[r3];
            const uint32_t c0 = counts[r0], c1 = counts[r1], c2 = counts[r2], c3 = counts[r3];
            len += (v0 != 0) + (v1 != 0) + (v2 != 0) + (v3 != 0);
            keep(v0 && c0, c0); keep(v1 && c1, c1); keep(v2 && c2, c2); keep(v3 && c3, c3);
            cmax = max(max(cmax, v0 ? c0 : 0u), max(max(v1 ? c1 : 0u, v2 ? c2 : 0u), v3 ? c3 : 0u));
        }
        // the tail (keep() ballots over whatever lanes are still active)
        for (uint64_t i0 = i - t; i0 < hi; i0 += NT) {
            const uint64_t j = i0 + t;
            uint32_t v = 0, c = 0;
            if (j < hi) {
                const uint32_t r = rows[j];
                v = valid[r];
                c = counts[r];
            }
            len += (v != 0);
            keep(v && c, c);
            cmax = max(cmax, v ? c : 0u);
        }
    }
    len = block_sum(len, s_red64);          // (its barriers also publish s_n and s_val)
    cmax = ~block_min(~cmax, s_red32);
    const uint64_t npos = s_n;
    const bool in_lds = npos <= CAP && cmax < 0xFFFFu;      // else: the passes below gather from global memory again

    // f(c) for every positive count of a valid row of the node
    auto for_each_pos = [&](auto f) {
        if (in_lds) {
            for (uint32_t i = (uint32_t)t; i < (uint32_t)npos; i += NT) f(s_val[i]);
        } else {
            for (uint64_t i = lo + t; i < hi; i += NT) {
                const uint32_t r = rows[i];
                if (!valid[r]) continue;
                const uint32_t c = counts[r];
                if (c) f(c);
            }
        }
    };

    uint32_t v1 = 0, v2 = 0;
    if (npos > 0) {
        // exact radix select of the element of rank k1 = (npos-1)/2 among the positive counts
        uint32_t prefix = 0;
        uint32_t rank = (uint32_t)((npos - 1) >> 1);
        const int top = cmax >> 24 ? 24 : cmax >> 16 ? 16 : cmax >> 8 ? 8 : 0;   // bytes above are zero in every count
        for (int shift = top; shift >= 0; shift -= 8) {
            if (t < 256) s_hist[t] = 0;
            __syncthreads();
            for_each_pos([&](uint32_t c) {
                if (shift == top || (c >> (shift + 8)) == prefix) atomicAdd(&s_hist[(c >> shift) & 255u], 1u);
            });
            __syncthreads();
            if (t == 0) {
                uint32_t cum = 0, b = 0;
                for (; b < 256; b++) {
                    if (cum + s_hist[b] > rank) break;
                    cum += s_hist[b];
                }
                s_pick[0] = b;
                s_pick[1] = rank - cum;
            }
            __syncthreads();
            prefix = (prefix << 8) | s_pick[0];
            rank = s_pick[1];
            __syncthreads();
        }
        v1 = prefix;
        v2 = v1;
        if ((npos & 1) == 0) {
            // second middle element (rank k1+1): v1 again if enough copies, else the next larger value
            uint64_t le = 0;
            uint32_t mg = 0xFFFFFFFFu;
            for_each_pos([&](uint32_t c) {
                if (c <= v1) le++;
                else mg = min(mg, c);
            });
            le = block_sum(le, s_red64);
            mg = block_min(mg, s_red32);
            if (le <= (npos >> 1)) v2 = mg;  // rank k2 = npos/2 (0-based) lies beyond the copies of v1
        }
    }
    // del_outlier: drop c >= 100 * median = 50 * (v1 + v2)
    const uint64_t med2 = (uint64_t)v1 + (uint64_t)v2;
    const uint64_t cut = 50ull * med2;
    uint64_t nk = 0, sk = 0;
    if (npos > 0)
        for_each_pos([&](uint32_t c) {
            if ((uint64_t)c < cut) { nk++; sk += c; }
        });
    nk = block_sum(nk, s_red64);
    sk = block_sum(sk, s_red64);
    if (t == 0) {
        ss_node_stat st;
        st.length = (uint32_t)len;
        st.n_pos = (uint32_t)npos;
        st.n_kept = (uint32_t)nk;
        st.reserved = 0;
        st.sum_kept = sk;
        st.median2 = med2;
        stats[node] = st;
    }
}

}  // namespace

extern "C" {

int ss_nodes_create(const uint32_t *rows, const uint64_t *offsets, uint32_t n_nodes, ss_nodes **out)
{
    if (!out || !offsets || (n_nodes && offsets[n_nodes] && !rows)) return SS_EINVAL;
    ss_nodes *ns = new (std::nothrow) ss_nodes();
    if (!ns) return SS_ENOMEM;
    ns->n_nodes = n_nodes;
    ns->n_rows_total = offsets[n_nodes];
    auto fail = [&](int rc) { ss_nodes_destroy(ns); return rc; };
    if (hipMalloc((void **)&ns->d_rows, std::max<uint64_t>(1, ns->n_rows_total) * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&ns->d_offsets, ((uint64_t)n_nodes + 1) * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc((void **)&ns->d_stats, std::max<uint32_t>(1, n_nodes) * sizeof(ss_node_stat)) != hipSuccess)
        return fail(SS_ENOMEM);
    if (ns->n_rows_total &&
        hipMemcpy(ns->d_rows, rows, ns->n_rows_total * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
        return fail(SS_EHIP);
    if (hipMemcpy(ns->d_offsets, offsets, ((uint64_t)n_nodes + 1) * sizeof(uint64_t), hipMemcpyHostToDevice) !=
        hipSuccess)
        return fail(SS_EHIP);
    *out = ns;
    return SS_OK;
}

int ss_nodes_destroy(ss_nodes *ns)
{
    if (!ns) return SS_OK;
    hipFree(ns->d_rows);
    hipFree(ns->d_offsets);
    hipFree(ns->d_stats);
    hipFree(ns->d_counts_rows);
    delete ns;
    return SS_OK;
}

int ss_nodes_reduce_dev(const ss_nodes *ns, const uint32_t *counts_rows_dev, const uint8_t *row_valid_dev,
                        ss_node_stat *stats_dev, void *stream)
{
    if (!ns || !counts_rows_dev || !row_valid_dev || !stats_dev) return SS_EINVAL;
    if (!ns->n_nodes) return SS_OK;
    hipLaunchKernelGGL(node_reduce_kernel, dim3(ns->n_nodes), dim3(NT), 0, ss::as_stream(stream), ns->d_rows,
                       ns->d_offsets, counts_rows_dev, row_valid_dev, stats_dev);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int ss_nodes_reduce(const ss_nodes *ns_c, const ss_db *db, ss_node_stat *stats)
{
    ss_nodes *ns = const_cast<ss_nodes *>(ns_c);
    if (!ns || !db || !stats) return SS_EINVAL;
    if (!ns->n_nodes) return SS_OK;
    uint64_t n_rows = 0;
    ss_db_info(db, &n_rows, nullptr, nullptr, nullptr);
    if (ns->counts_rows_cap < n_rows) {
        hipFree(ns->d_counts_rows);
        ns->d_counts_rows = nullptr;
        ns->counts_rows_cap = 0;
        SS_HIP(hipMalloc((void **)&ns->d_counts_rows, std::max<uint64_t>(1, n_rows) * sizeof(uint32_t)));
        ns->counts_rows_cap = n_rows;
    }
    int rc = ss_counts_rows_dev(db, ns->d_counts_rows, nullptr);
    if (rc) return rc;
    rc = ss_nodes_reduce_dev(ns, ns->d_counts_rows, ss_db_row_valid_dev(db), ns->d_stats, nullptr);
    if (rc) return rc;
    SS_HIP(hipMemcpy(stats, ns->d_stats, (uint64_t)ns->n_nodes * sizeof(ss_node_stat), hipMemcpyDeviceToHost));
    return SS_OK;
}

int ss_rows_reduce(const ss_db *db, const uint32_t *rows, uint64_t n, ss_node_stat *stat)
{
    if (!db || !stat || (n && !rows)) return SS_EINVAL;
    const uint64_t offs[2] = {0, n};
    ss_nodes *ns = nullptr;
    int rc = ss_nodes_create(rows, offs, 1, &ns);
    if (rc) return rc;
    rc = ss_nodes_reduce(ns, db, stat);
    ss_nodes_destroy(ns);
    return rc;
}

}  // extern "C"
