// ss_nodes.hip -- per-node reductions of the cluster-search-tree walk, all nodes in one launch.
//
// Replaces match_node + del_outlier (library/identify.py:106-127; the low-depth variant
// identify_low_depth.py:77-101 differs only by a host-side minimum on `length`), which the
// reference runs in Python per visited node after re-reading <db>/kmers/<id>.
//
// One 256-thread workgroup per node; a node list is <= 30000 rows (StrainScan_build.py:77-78)
// so every pass is a few L2-resident gathers.  np.median needs order statistics: an exact
// 4 x 8-bit radix select over LDS histograms (no sort), then one pass for the second middle
// element when the profile has even length, then the outlier cut (c >= 100 * median dropped)
// with integer arithmetic: c >= 50 * (v1 + v2).  Algorithmic bytes: 4 B row index + 4 B count
// (+1 B valid flag) per node k-mer per pass.
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

constexpr int NT = 256;

__device__ __forceinline__ uint64_t block_sum(uint64_t v, uint64_t *s_red)
{
    // wave64 shuffle reduction, then 4 partials through LDS
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    uint64_t r = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    return r;
}

__device__ __forceinline__ uint32_t block_min(uint32_t v, uint32_t *s_red)
{
    for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
}

__global__ __launch_bounds__(NT) void node_reduce_kernel(const uint32_t *__restrict__ rows,
                                                         const uint64_t *__restrict__ offsets,
                                                         const uint32_t *__restrict__ counts,
                                                         const uint8_t *__restrict__ valid,
                                                         ss_node_stat *__restrict__ stats)
{
    __shared__ uint32_t s_hist[256];
    __shared__ uint64_t s_red64[4];
    __shared__ uint32_t s_red32[4];
    __shared__ uint32_t s_pick[2];  // [0] = chosen bin, [1] = rank inside it

    const uint32_t node = blockIdx.x;
    const uint64_t lo = offsets[node], hi = offsets[node + 1];
    const int t = threadIdx.x;

    // pass 0: length (valid rows) and n_pos (valid rows seen at least once); four independent
    // gathers in flight per lane (most nodes of a sample have no hits: this pass is all they cost)
    uint64_t len = 0, npos = 0;
    {
        uint64_t i = lo + t;
        for (; i + 3 * NT < hi; i += 4 * NT) {
            const uint32_t r0 = rows[i], r1 = rows[i + NT], r2 = rows[i + 2 * NT], r3 = rows[i + 3 * NT];
            const uint32_t v0 = valid[r0], v1 = valid[r1], v2 = valid[r2], v3 = valid[r3];
            const uint32_t c0 = counts[r0], c1 = counts[r1], c2 = counts[r2], c3 = counts[r3];
            len += (v0 != 0) + (v1 != 0) + (v2 != 0) + (v3 != 0);
            npos += (v0 && c0) + (v1 && c1) + (v2 && c2) + (v3 && c3);
        }
        for (; i < hi; i += NT) {
            const uint32_t r = rows[i];
            if (valid[r]) {
                len++;
                if (counts[r] > 0) npos++;
            }
        }
    }
    len = block_sum(len, s_red64);
    npos = block_sum(npos, s_red64);

    uint32_t v1 = 0, v2 = 0;
    if (npos > 0) {
        // exact radix select of the element of rank k1 = (npos-1)/2 among the positive counts
        uint32_t prefix = 0;
        uint32_t rank = (uint32_t)((npos - 1) >> 1);
        for (int shift = 24; shift >= 0; shift -= 8) {
            s_hist[t] = 0;
            __syncthreads();
            for (uint64_t i = lo + t; i < hi; i += NT) {
                const uint32_t r = rows[i];
                if (!valid[r]) continue;
                const uint32_t c = counts[r];
                if (c == 0) continue;
                if (shift == 24 || (c >> (shift + 8)) == prefix) atomicAdd(&s_hist[(c >> shift) & 255u], 1u);
            }
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
            for (uint64_t i = lo + t; i < hi; i += NT) {
                const uint32_t r = rows[i];
                if (!valid[r]) continue;
                const uint32_t c = counts[r];
                if (c == 0) continue;
                if (c <= v1) le++;
                else mg = min(mg, c);
            }
            le = block_sum(le, s_red64);
            mg = block_min(mg, s_red32);
            if (le <= (npos >> 1)) v2 = mg;  // rank k2 = npos/2 (0-based) lies beyond the copies of v1
        }
    }
    // del_outlier: drop c >= 100 * median = 50 * (v1 + v2)
    const uint64_t med2 = (uint64_t)v1 + (uint64_t)v2;
    const uint64_t cut = 50ull * med2;
    uint64_t nk = 0, sk = 0;
    if (npos > 0) {
        for (uint64_t i = lo + t; i < hi; i += NT) {
            const uint32_t r = rows[i];
            if (!valid[r]) continue;
            const uint32_t c = counts[r];
            if (c == 0 || (uint64_t)c >= cut) continue;
            nk++;
            sk += c;
        }
    }
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
