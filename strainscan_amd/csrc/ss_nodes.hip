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
//
// Harvest path (ss_nodes_bind + ss_nodes_harvest_dev + ss_nodes_reduce_touched_dev): the gather above touches every
// row of every node -- 25 M random counter reads for an E. coli table whose rows lie scattered over the index -- although
// a sample has hits in a few dozen of the 1645 nodes.  Binding a node set to a database sorts the (counter, list position)
// pairs of all nodes by counter once; after a scan ONE streaming pass over that sorted list reads the counters in
// ascending order (sectors of the counter array, no gathers), writes the non-zero ones to their list positions in a
// dense node-major buffer and flags their nodes; the reduction then runs over the flagged nodes only, from contiguous
// memory, and clears what it read.  Several GPUs: flags MAX-all-reduced, the flagged segments packed, summed (RCCL),
// unpacked -- a few MB instead of 4 bytes per database row (dist.py).
#include "ss_common.h"

#include <hipcub/hipcub.hpp>

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
    // bound to one database (harvest path)
    const ss_db *bound = nullptr;
    uint64_t n_used = 0;                // list positions whose row is valid
    uint32_t *d_pkeys = nullptr;        // [n_used] counter index, ascending
    uint32_t *d_ppos = nullptr;         // [n_used] list position of that entry
    uint32_t *d_pnode = nullptr;        // [n_used] node of that list position
    uint32_t *d_len = nullptr;          // [n_nodes] valid rows per node (the `length` of match_node: static)
    uint32_t *d_val = nullptr;          // [n_rows_total] counts by list position, zero outside a harvest..reduce window
    uint32_t *d_touched = nullptr;      // [n_nodes] 1 = the node has a non-zero count
    uint64_t *d_packoff = nullptr;      // [n_nodes + 1] offsets of the touched nodes' segments in the packed buffer
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

// DENSE: counts = the node-major buffer of the harvest path (entry i = list position i, invalid rows and rows without
// hits hold 0), rows unused, valid = nullptr, len_static / touched given; the kernel clears what it read.
template <bool DENSE>
__global__ __launch_bounds__(NT) void node_reduce_kernel(const uint32_t *__restrict__ rows,
                                                         const uint64_t *__restrict__ offsets,
                                                         uint32_t *__restrict__ counts,
                                                         const uint8_t *__restrict__ valid,
                                                         ss_node_stat *__restrict__ stats,
                                                         const uint32_t *__restrict__ len_static,
                                                         uint32_t *__restrict__ touched)
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
    if (DENSE && !touched[node]) {           // no hits in this node (most nodes of a sample)
        if (t == 0) {
            ss_node_stat st;
            st.length = len_static[node]; st.n_pos = 0; st.n_kept = 0; st.reserved = 0; st.sum_kept = 0; st.median2 = 0;
            stats[node] = st;
        }
        return;
    }
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
            uint32_t r0, r1, r2, r3, v0 = 1, v1 = 1, v2 = 1, v3 = 1;
            if (DENSE) { r0 = (uint32_t)i; r1 = r0 + NT; r2 = r0 + 2 * NT; r3 = r0 + 3 * NT; }
            else {
                r0 = rows[i]; r1 = rows[i + NT]; r2 = rows[i + 2 * NT]; r3 = rows[i + 3 * NT];
                v0 = valid[r0]; v1 = valid[r1]; v2 = valid[r2]; v3 = valid[r3];
            }
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
                const uint32_t r = DENSE ? (uint32_t)j : rows[j];
                v = DENSE ? 1u : valid[r];
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
                const uint32_t r = DENSE ? (uint32_t)i : rows[i];
                if (!DENSE && !valid[r]) continue;
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
    if (DENSE) {                              // leave the buffer and the flag clean for the next scan
        for (uint64_t i = lo + t; i < hi; i += NT) counts[i] = 0;
        if (t == 0) touched[node] = 0;
        len = len_static[node];
    }
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

// ---- harvest path ---------------------------------------------------------------------------------
__global__ void bind_cslot_kernel(const uint32_t *__restrict__ rows, uint64_t n, const uint32_t *__restrict__ slot_of_row,
                                  const uint8_t *__restrict__ valid, uint64_t n_db_rows, uint32_t *__restrict__ keys,
                                  uint32_t *__restrict__ pos)
{
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = rows[p];
        keys[p] = (r < n_db_rows && valid[r]) ? slot_of_row[r] : SS_NO_SLOT;
        pos[p] = (uint32_t)p;
    }
}

// node of a list position: the last offset <= p
__device__ __forceinline__ uint32_t node_of_pos(const uint64_t *__restrict__ offsets, uint32_t n_nodes, uint64_t p)
{
    uint32_t lo = 0, hi = n_nodes;             // offsets[lo] <= p < offsets[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void bind_node_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ pos, uint64_t n,
                                 const uint64_t *__restrict__ offsets, uint32_t n_nodes, uint32_t *__restrict__ pnode)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (keys[i] == SS_NO_SLOT) continue;    // sorted: the invalid positions are the tail
        pnode[i] = node_of_pos(offsets, n_nodes, pos[i]);
    }
}

// valid rows per node (keys in list order, before the sort): one block per node
__global__ __launch_bounds__(256) void bind_len_kernel(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ offsets,
                                                       uint32_t *__restrict__ len)
{
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t c = 0;
    for (uint64_t i = offsets[blockIdx.x] + threadIdx.x; i < offsets[blockIdx.x + 1]; i += 256) c += keys[i] != SS_NO_SLOT;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) len[blockIdx.x] = s_cnt;
}

// one streaming pass: counters in ascending order -> list positions of the dense buffer, node flags
__global__ __launch_bounds__(256) void harvest_kernel(const uint32_t *__restrict__ pkeys, const uint32_t *__restrict__ ppos,
                                                      const uint32_t *__restrict__ pnode, uint64_t n_used,
                                                      const uint32_t *__restrict__ counts, uint32_t *__restrict__ val,
                                                      uint32_t *__restrict__ touched)
{
    const uint64_t n4 = n_used >> 2;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 k = reinterpret_cast<const uint4 *>(pkeys)[q];
        const uint32_t c0 = counts[k.x], c1 = counts[k.y], c2 = counts[k.z], c3 = counts[k.w];
        if (c0 | c1 | c2 | c3) {
            const uint64_t i = q << 2;
            if (c0) { val[ppos[i]] = c0; touched[pnode[i]] = 1u; }
            if (c1) { val[ppos[i + 1]] = c1; touched[pnode[i + 1]] = 1u; }
            if (c2) { val[ppos[i + 2]] = c2; touched[pnode[i + 2]] = 1u; }
            if (c3) { val[ppos[i + 3]] = c3; touched[pnode[i + 3]] = 1u; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n_used & 3)) {
        const uint64_t i = (n4 << 2) + threadIdx.x;
        const uint32_t c = counts[pkeys[i]];
        if (c) { val[ppos[i]] = c; touched[pnode[i]] = 1u; }
    }
}

// offsets of the touched nodes' segments in a packed buffer (one block; n_nodes is a few thousand)
__global__ __launch_bounds__(1024) void pack_offsets_kernel(const uint32_t *__restrict__ touched, const uint64_t *__restrict__ offsets,
                                                            uint32_t n_nodes, uint64_t *__restrict__ packoff)
{
    __shared__ uint64_t s_part[1024];
    const int t = threadIdx.x;
    const uint32_t per = (n_nodes + 1023u) / 1024u, a = min(n_nodes, (uint32_t)t * per), b = min(n_nodes, a + per);
    uint64_t sum = 0;
    for (uint32_t i = a; i < b; i++) sum += touched[i] ? offsets[i + 1] - offsets[i] : 0;
    s_part[t] = sum;
    __syncthreads();
    if (t == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; i++) { const uint64_t v = s_part[i]; s_part[i] = run; run += v; }
        packoff[n_nodes] = run;
    }
    __syncthreads();
    uint64_t run = s_part[t];
    for (uint32_t i = a; i < b; i++) { packoff[i] = run; run += touched[i] ? offsets[i + 1] - offsets[i] : 0; }
}

// PACK: packed <- val segments of the touched nodes; else the other way round
template <bool PACK>
__global__ __launch_bounds__(256) void pack_copy_kernel(const uint32_t *__restrict__ touched, const uint64_t *__restrict__ offsets,
                                                        const uint64_t *__restrict__ packoff, uint32_t *__restrict__ val,
                                                        uint32_t *__restrict__ packed, uint64_t cap)
{
    const uint32_t node = blockIdx.x;
    if (!touched[node]) return;
    const uint64_t lo = offsets[node], po = packoff[node];
    if (po >= cap) return;
    const uint64_t n = min(offsets[node + 1] - lo, cap - po);     // capped form: what does not fit stays where it is
    for (uint64_t i = threadIdx.x; i < n; i += 256) {
        if (PACK) packed[po + i] = val[lo + i];
        else val[lo + i] = packed[po + i];
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
    hipFree(ns->d_pkeys); hipFree(ns->d_ppos); hipFree(ns->d_pnode); hipFree(ns->d_len); hipFree(ns->d_val);
    hipFree(ns->d_touched); hipFree(ns->d_packoff);
    delete ns;
    return SS_OK;
}

int ss_nodes_bind(ss_nodes *ns, const ss_db *db)
{
    if (!ns || !db) return SS_EINVAL;
    if (ns->bound == db) return SS_OK;
    hipFree(ns->d_pkeys); hipFree(ns->d_ppos); hipFree(ns->d_pnode); hipFree(ns->d_len); hipFree(ns->d_val);
    hipFree(ns->d_touched); hipFree(ns->d_packoff);
    ns->d_pkeys = ns->d_ppos = ns->d_pnode = ns->d_len = ns->d_val = ns->d_touched = nullptr;
    ns->d_packoff = nullptr;
    ns->bound = nullptr;
    const uint64_t n = ns->n_rows_total, n1 = std::max<uint64_t>(1, n), nn = std::max<uint32_t>(1, ns->n_nodes);
    if (n > 0x7FFFFFFFull) return SS_ERANGE;            // hipcub's radix sort takes an int count
    uint32_t *k_in = nullptr, *p_in = nullptr;
    void *tmp = nullptr;
    auto cleanup = [&] { hipFree(k_in); hipFree(p_in); hipFree(tmp); };
#define SS_B(call) do { if ((call) != hipSuccess) { cleanup(); ss::set_last_error(#call, __FILE__, __LINE__, hipGetLastError()); return SS_EHIP; } } while (0)
    SS_B(hipMalloc((void **)&k_in, n1 * 4)); SS_B(hipMalloc((void **)&p_in, n1 * 4));
    SS_B(hipMalloc((void **)&ns->d_pkeys, n1 * 4)); SS_B(hipMalloc((void **)&ns->d_ppos, n1 * 4));
    SS_B(hipMalloc((void **)&ns->d_pnode, n1 * 4)); SS_B(hipMalloc((void **)&ns->d_val, n1 * 4));
    SS_B(hipMalloc((void **)&ns->d_len, nn * 4)); SS_B(hipMalloc((void **)&ns->d_touched, nn * 4));
    SS_B(hipMalloc((void **)&ns->d_packoff, ((uint64_t)nn + 1) * 8));
    SS_B(hipMemset(ns->d_val, 0, n1 * 4)); SS_B(hipMemset(ns->d_len, 0, nn * 4)); SS_B(hipMemset(ns->d_touched, 0, nn * 4));
    if (n) {
        uint64_t n_db_rows = 0;
        ss_db_info(db, &n_db_rows, nullptr, nullptr, nullptr);
        const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, 65536);
        hipLaunchKernelGGL(bind_cslot_kernel, dim3(blocks), dim3(256), 0, 0, ns->d_rows, n, db->d_slot_of_row, db->d_row_valid,
                           n_db_rows, k_in, p_in);
        hipLaunchKernelGGL(bind_len_kernel, dim3(ns->n_nodes), dim3(256), 0, 0, k_in, ns->d_offsets, ns->d_len);
        size_t tmp_bytes = 0;
        SS_B(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k_in, ns->d_pkeys, p_in, ns->d_ppos, (int)n));
        SS_B(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)));
        SS_B(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, k_in, ns->d_pkeys, p_in, ns->d_ppos, (int)n));
        hipLaunchKernelGGL(bind_node_kernel, dim3(blocks), dim3(256), 0, 0, ns->d_pkeys, ns->d_ppos, n, ns->d_offsets, ns->n_nodes,
                           ns->d_pnode);
        SS_B(hipGetLastError());
    }
    std::vector<uint32_t> len(nn, 0);
    SS_B(hipMemcpy(len.data(), ns->d_len, (uint64_t)nn * 4, hipMemcpyDeviceToHost));
#undef SS_B
    cleanup();
    uint64_t used = 0;
    for (uint32_t i = 0; i < ns->n_nodes; i++) used += len[i];
    ns->n_used = used;
    ns->bound = db;
    return SS_OK;
}

int ss_nodes_harvest_dev(ss_nodes *ns, const ss_db *db, void *stream)
{
    if (!ns || !db || ns->bound != db) return SS_EINVAL;
    if (!ns->n_used) return SS_OK;
    const unsigned blocks = (unsigned)std::min<uint64_t>(((ns->n_used >> 2) + 255) / 256 + 1, 256 * 32);
    hipLaunchKernelGGL(harvest_kernel, dim3(blocks), dim3(256), 0, ss::as_stream(stream), ns->d_pkeys, ns->d_ppos, ns->d_pnode,
                       ns->n_used, db->d_counts, ns->d_val, ns->d_touched);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int ss_nodes_reduce_touched_dev(ss_nodes *ns, ss_node_stat *stats_dev, void *stream)
{
    if (!ns || !ns->bound || !stats_dev) return SS_EINVAL;
    if (!ns->n_nodes) return SS_OK;
    hipLaunchKernelGGL((node_reduce_kernel<true>), dim3(ns->n_nodes), dim3(NT), 0, ss::as_stream(stream), (const uint32_t *)nullptr,
                       ns->d_offsets, ns->d_val, (const uint8_t *)nullptr, stats_dev, ns->d_len, ns->d_touched);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int ss_nodes_touched_get_dev(const ss_nodes *ns, uint32_t *flags_dev, void *stream)
{
    if (!ns || !ns->bound || !flags_dev) return SS_EINVAL;
    SS_HIP(hipMemcpyAsync(flags_dev, ns->d_touched, (uint64_t)ns->n_nodes * 4, hipMemcpyDeviceToDevice, ss::as_stream(stream)));
    return SS_OK;
}

int ss_nodes_touched_set_dev(ss_nodes *ns, const uint32_t *flags_dev, void *stream)
{
    if (!ns || !ns->bound || !flags_dev) return SS_EINVAL;
    SS_HIP(hipMemcpyAsync(ns->d_touched, flags_dev, (uint64_t)ns->n_nodes * 4, hipMemcpyDeviceToDevice, ss::as_stream(stream)));
    return SS_OK;
}

int ss_nodes_pack_dev(ss_nodes *ns, uint32_t *packed_dev, uint64_t cap, uint64_t *n_packed, void *stream)
{
    if (!ns || !ns->bound || !n_packed) return SS_EINVAL;
    *n_packed = 0;
    if (!ns->n_nodes) return SS_OK;
    hipStream_t st = ss::as_stream(stream);
    hipLaunchKernelGGL(pack_offsets_kernel, dim3(1), dim3(1024), 0, st, ns->d_touched, ns->d_offsets, ns->n_nodes, ns->d_packoff);
    uint64_t total = 0;
    SS_HIP(hipMemcpyAsync(&total, ns->d_packoff + ns->n_nodes, 8, hipMemcpyDeviceToHost, st));
    SS_HIP(hipStreamSynchronize(st));
    *n_packed = total;
    if (!packed_dev) return SS_OK;              // size query
    if (total > cap) return SS_ERANGE;
    hipLaunchKernelGGL((pack_copy_kernel<true>), dim3(ns->n_nodes), dim3(256), 0, st, ns->d_touched, ns->d_offsets, ns->d_packoff,
                       ns->d_val, packed_dev, ~0ull);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int ss_nodes_pack_capped_dev(ss_nodes *ns, uint32_t *packed_dev, uint64_t cap, uint64_t *total_dev, void *stream)
{
    if (!ns || !ns->bound || !packed_dev || !total_dev) return SS_EINVAL;
    hipStream_t st = ss::as_stream(stream);
    if (!ns->n_nodes) { SS_HIP(hipMemsetAsync(total_dev, 0, 8, st)); return SS_OK; }
    hipLaunchKernelGGL(pack_offsets_kernel, dim3(1), dim3(1024), 0, st, ns->d_touched, ns->d_offsets, ns->n_nodes, ns->d_packoff);
    SS_HIP(hipMemcpyAsync(total_dev, ns->d_packoff + ns->n_nodes, 8, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL((pack_copy_kernel<true>), dim3(ns->n_nodes), dim3(256), 0, st, ns->d_touched, ns->d_offsets, ns->d_packoff,
                       ns->d_val, packed_dev, cap);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int ss_nodes_unpack_capped_dev(ss_nodes *ns, const uint32_t *packed_dev, uint64_t cap, void *stream)
{
    if (!ns || !ns->bound || !packed_dev) return SS_EINVAL;
    if (!ns->n_nodes) return SS_OK;
    hipLaunchKernelGGL((pack_copy_kernel<false>), dim3(ns->n_nodes), dim3(256), 0, ss::as_stream(stream), ns->d_touched, ns->d_offsets,
                       ns->d_packoff, ns->d_val, const_cast<uint32_t *>(packed_dev), cap);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int ss_nodes_clear_dev(ss_nodes *ns, void *stream)
{
    if (!ns || !ns->bound) return SS_EINVAL;
    hipStream_t st = ss::as_stream(stream);
    SS_HIP(hipMemsetAsync(ns->d_val, 0, std::max<uint64_t>(1, ns->n_rows_total) * 4, st));
    SS_HIP(hipMemsetAsync(ns->d_touched, 0, (uint64_t)std::max<uint32_t>(1, ns->n_nodes) * 4, st));
    return SS_OK;
}

int ss_nodes_unpack_dev(ss_nodes *ns, const uint32_t *packed_dev, void *stream)
{
    if (!ns || !ns->bound || !packed_dev) return SS_EINVAL;
    if (!ns->n_nodes) return SS_OK;
    hipLaunchKernelGGL((pack_copy_kernel<false>), dim3(ns->n_nodes), dim3(256), 0, ss::as_stream(stream), ns->d_touched, ns->d_offsets,
                       ns->d_packoff, ns->d_val, const_cast<uint32_t *>(packed_dev), ~0ull);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int ss_nodes_reduce_dev(const ss_nodes *ns, const uint32_t *counts_rows_dev, const uint8_t *row_valid_dev,
                        ss_node_stat *stats_dev, void *stream)
{
    if (!ns || !counts_rows_dev || !row_valid_dev || !stats_dev) return SS_EINVAL;
    if (!ns->n_nodes) return SS_OK;
    hipLaunchKernelGGL((node_reduce_kernel<false>), dim3(ns->n_nodes), dim3(NT), 0, ss::as_stream(stream), ns->d_rows,
                       ns->d_offsets, const_cast<uint32_t *>(counts_rows_dev), row_valid_dev, stats_dev, (const uint32_t *)nullptr,
                       (uint32_t *)nullptr);
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
