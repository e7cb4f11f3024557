// ss_l2.hip -- intra-cluster (layer 2) kernels: the k-mer x strain matrix as bit planes.
//
// Replaces the dense numpy passes of library/identify_strains_L2_Enet_Pscan_new_sp.py:
//   stat_cov / cal_cov_all (:33-49), get_remainc (:94-108), get_candidate_arr (:121-134),
//   optimize_dominat_y (:136-175), get_avg_depth (:110-120), the Pre_Scan loop (:302-371),
//   the row filter (:402-415) and the fold slicing + Gram products inside ElasticNetCV.fit
//   (:437-442, scikit-learn _path_residuals).
//
// X is binary (Build_kmer_sets_..._sp.py:412-414 writes only 1s), so the reference's products
//   ix * iy > 1   <=>   X bit set AND y > 1
// and every "count" it derives is a popcount of ANDed bit vectors.  Layout in HBM: one bit plane
// per strain, K bits each (W = ceil(K/32) dwords), plane-major: Xb[s * W + w].  A pre-scan
// iteration streams S*K/8 bytes once; HBM-bound, coalesced 16-byte loads, wave popcounts.
//
// The elastic-net inputs are sufficient statistics: with <= 16 selected strains a row of X is a
// <= 16-bit pattern m, and every Gram entry, X'y, y'y and test-fold residual sum is a function
// of per-pattern (count, sum y, sum y^2).  Those are exact integers (u64), so the fold
// statistics equal the reference's BLAS results to the last bit regardless of summation order.
#include "ss_common.h"

#include <hipcub/hipcub.hpp>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <thread>
#include <vector>

struct ss_l2 {
    uint64_t K = 0;
    uint32_t S = 0;
    uint64_t W = 0;          // dwords per plane (multiple of 4 for 16-byte loads)
    uint32_t *d_x = nullptr; // [S][W]
    // overlap_matrix.npz (CSR int8, K x #clusters) for ss_l2_prepare, uploaded once per cluster (ss_l2_set_overlap)
    int64_t *d_om_ptr = nullptr;
    int32_t *d_om_idx = nullptr;
    int8_t *d_om_val = nullptr;
    uint32_t om_cols = 0;
    bool has_om = false;
    uint8_t *d_blob = nullptr; // ss_l2_import: ONE allocation holding the cluster image file; the pointers above point into it
};

namespace {

constexpr int NT = 256;

__global__ void pack_csr_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                uint64_t K, uint32_t S, uint64_t W, uint32_t *x, int *bad)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    for (int64_t j = indptr[k]; j < indptr[k + 1]; j++) {
        const int32_t s = indices[j];
        if (s < 0 || (uint32_t)s >= S) { *bad = 1; continue; }
        atomicOr(&x[(uint64_t)s * W + (k >> 5)], 1u << (k & 31));
    }
}

// The same without atomics for rows whose column indices ascend (scipy's canonical CSR): one lane per row, the wave
// walks the S columns together -- ballot(lane's next index == s) IS the two 32-bit words of column s for these 64
// rows.  513 M atomicOr (16 ms at K = 5 M, S = 300) become 23 M 8-byte stores.  A row whose list is not consumed by
// the walk (unsorted or out-of-range indices) raises *redo: the caller runs pack_csr_kernel.
__global__ __launch_bounds__(256) void pack_csr_sorted_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                              uint64_t K, uint32_t S, uint64_t W, uint32_t *x, int *redo)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t p = 0, e = 0;
    if (k < K) { p = indptr[k]; e = indptr[k + 1]; }
    int32_t nxt = p < e ? indices[p] : 0x7FFFFFFF;
    const uint64_t w0 = (k & ~63ull) >> 5;                   // first of the wave's two words
    const bool writer = (threadIdx.x & 63) == 0;
    for (uint32_t s = 0; s < S; s++) {
        const bool has = nxt == (int32_t)s;
        const uint64_t m = __ballot(has);
        if (has) {
            do { p++; nxt = p < e ? indices[p] : 0x7FFFFFFF; } while (nxt == (int32_t)s);   // duplicates collapse
        }
        if (writer && m && w0 < W) {
            uint32_t *dst = x + (uint64_t)s * W + w0;
            dst[0] = (uint32_t)m;
            if (w0 + 1 < W) dst[1] = (uint32_t)(m >> 32);
        }
    }
    if (p < e) *redo = 1;
}

// CSR -> bit planes with COALESCED reads of the index array.  The kernel above gives every lane a row and lets it walk its
// own list four bytes at a time: neighbouring lanes are a whole row (~100 entries) apart, every step touches 64 different
// sectors and each sector is fetched again for each of its 16 entries -- 19.9 GB fetched for a 2 GB CSR (K = 5 M, S = 300,
// 5.6 ms).  Here a wave owns 64 consecutive rows = one CONTIGUOUS span of the index array (their lists lie back to back):
// the lanes read the span 64 entries at a time, find each entry's row by binary search over the wave's 65 row offsets (LDS),
// and OR the row's bit into the column's 64-bit word in LDS; the S words are then stored to the planes.  Any column order,
// duplicates welcome (OR); a column index outside [0, S) raises *bad.  S <= PACK_SMAX (the LDS words), else the kernels above.
constexpr uint32_t PACK_SMAX = 4096;

__global__ __launch_bounds__(64) void pack_csr_span_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                           uint64_t K, uint32_t S, uint64_t W, uint32_t *x, int *bad)
{
    extern __shared__ unsigned long long s_words[];      // [S] column words of these 64 rows, then 65 row offsets
    const int lane = threadIdx.x;
    const uint64_t k0 = (uint64_t)blockIdx.x * 64;
    long long *s_off = reinterpret_cast<long long *>(s_words + S);
    for (uint32_t c = lane; c < S; c += 64) s_words[c] = 0;
    const uint64_t kr = min(K, k0 + lane);
    s_off[lane] = indptr[kr];
    if (lane == 0) s_off[64] = indptr[min(K, k0 + 64)];
    __syncthreads();
    const long long lo = s_off[0], hi = s_off[64];
    int flag = 0;
    for (long long e = lo + lane; e < hi; e += 64) {
        const int32_t c = indices[e];
        int a = 0, b = 63;                               // the last row whose offset is <= e
        while (a < b) {
            const int mid = (a + b + 1) >> 1;
            if (s_off[mid] <= e) a = mid; else b = mid - 1;
        }
        if (c < 0 || (uint32_t)c >= S) flag = 1;
        else atomicOr(&s_words[c], 1ull << a);
    }
    if (flag) *bad = 1;
    __syncthreads();
    const uint64_t w0 = k0 >> 5;
    for (uint32_t c = lane; c < S; c += 64) {
        const unsigned long long m = s_words[c];
        if (!m) continue;                                // the planes were zeroed
        uint32_t *dst = x + (uint64_t)c * W + w0;
        dst[0] = (uint32_t)m;
        if (w0 + 1 < W) dst[1] = (uint32_t)(m >> 32);
    }
}

// out1[s] = popc(X_s & A), out2[s] = popc(X_s & A & B); A or B may be null (= all ones)
__global__ __launch_bounds__(NT) void popc2_kernel(const uint32_t *__restrict__ x, uint64_t W,
                                                   const uint32_t *__restrict__ A, const uint32_t *__restrict__ B,
                                                   unsigned long long *out1, unsigned long long *out2)
{
    const uint32_t s = blockIdx.y;
    const uint4 *xs = reinterpret_cast<const uint4 *>(x + (uint64_t)s * W);
    const uint4 *a4 = reinterpret_cast<const uint4 *>(A);
    const uint4 *b4 = reinterpret_cast<const uint4 *>(B);
    const uint64_t W4 = W >> 2;
    uint32_t c1 = 0, c2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < W4; i += (uint64_t)gridDim.x * NT) {
        uint4 v = xs[i];
        if (A) { const uint4 a = a4[i]; v.x &= a.x; v.y &= a.y; v.z &= a.z; v.w &= a.w; }
        c1 += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
        if (B) { const uint4 b = b4[i]; v.x &= b.x; v.y &= b.y; v.z &= b.z; v.w &= b.w; }
        c2 += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    for (int off = 32; off > 0; off >>= 1) {
        c1 += __shfl_down(c1, off, 64);
        c2 += __shfl_down(c2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (c1) atomicAdd(&out1[s], (unsigned long long)c1);
        if (c2) atomicAdd(&out2[s], (unsigned long long)c2);
    }
}

__global__ void andnot_col_kernel(const uint32_t *__restrict__ x, uint64_t W, uint32_t col, uint32_t *nu)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < W; i += (uint64_t)gridDim.x * blockDim.x)
        nu[i] &= ~x[(uint64_t)col * W + i];
}

__global__ void max_u32_kernel(const uint32_t *__restrict__ y, uint64_t n, uint32_t *out)
{
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) m = max(m, y[i]);
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// ---- masked order statistics: per listed column, values y[k] over rows with X bit set and
// y[k] != 0; two ranks per column are selected together (8-bit radix passes) ------------------
struct SelState {
    unsigned long long n;     // number of non-zero values in the column
    uint32_t prefix[2];       // selected high bits so far
    unsigned long long rank[2];
    uint32_t done;
};

// Lane = four consecutive rows -- their values are one 16-byte load of y, their bits one nibble of a plane's word (eight
// lanes share a word) -- and a workgroup takes CT columns at a time for its rows, so y is read once per CT columns instead
// of once per column.  (Walking the set bits of a word per lane gathered y four bytes at a time from 64 different lines per
// instruction: 1.03 ms per pass over K = 5 M x S = 300, bound by the L1's address path; one column per workgroup with
// coalesced y: 0.97 ms, now bound by re-reading y 300 times out of the MALL.)  k-mer counts crowd into a few values, i.e. a
// few bins: every column's histogram is kept in HCOPY copies (copy = lane mod HCOPY) so that the lanes of one LDS
// instruction rarely share a word.
constexpr int CT = 8, HCOPY = 4;

__device__ __forceinline__ uint4 load_y4(const uint32_t *__restrict__ y, uint64_t k, uint64_t K)
{
    if (k + 4 <= K) return *reinterpret_cast<const uint4 *>(y + k);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (k < K) v.x = y[k];
    if (k + 1 < K) v.y = y[k + 1];
    if (k + 2 < K) v.z = y[k + 2];
    return v;
}

// blockIdx.y = tile of CT listed columns; hist[ncols][2][256]
template <bool FIRST>
__global__ __launch_bounds__(NT) void sel_hist_kernel(const uint32_t *__restrict__ x, uint64_t W, uint64_t K,
                                                      const uint32_t *__restrict__ y, const uint32_t *__restrict__ cols, uint32_t ncols,
                                                      int shift, const SelState *__restrict__ st, uint32_t *hist)
{
    constexpr int NH = FIRST ? 1 : 2;
    // (the passes after the first keep TWO histograms per column: 64 KB + a few words -- gfx950's 160 KB of LDS per CU hold two
    //  such workgroups; a 64 KB-LDS target could not build this, and this library is built for gfx950 only)
    static_assert(sizeof(uint32_t) * NH * CT * HCOPY * 256 + 1024 <= 80 * 1024, "sel_hist_kernel: two workgroups per CU must fit gfx950's 160 KB of LDS");
    __shared__ uint32_t h[NH][CT][HCOPY][256];
    __shared__ uint32_t s_p[2][CT];
    __shared__ const uint32_t *s_x[CT];
    const uint32_t c0 = blockIdx.y * CT, nc = min((uint32_t)CT, ncols - c0);
    for (int i = threadIdx.x; i < NH * CT * HCOPY * 256; i += NT) (&h[0][0][0][0])[i] = 0;
    if (threadIdx.x < nc) {
        s_x[threadIdx.x] = x + (uint64_t)cols[c0 + threadIdx.x] * W;
        s_p[0][threadIdx.x] = st[c0 + threadIdx.x].prefix[0];
        s_p[1][threadIdx.x] = st[c0 + threadIdx.x].prefix[1];
    }
    __syncthreads();
    const int cp = threadIdx.x % HCOPY;
    const uint64_t K4 = (K + 3) >> 2;
    for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < K4; i += (uint64_t)gridDim.x * NT) {
        const uint4 y4 = load_y4(y, i << 2, K);
        const uint32_t vv[4] = {y4.x, y4.y, y4.z, y4.w};
        if (!(vv[0] | vv[1] | vv[2] | vv[3])) continue;
        uint32_t d[4], hi[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { d[r] = (vv[r] >> shift) & 255u; hi[r] = FIRST ? 0u : vv[r] >> (shift + 8); }
        const uint32_t sh = (uint32_t)(i & 7) * 4u;
        uint32_t xw[CT];                          // the CT planes' words first: their round trips overlap
#pragma unroll
        for (int c = 0; c < CT; c++) xw[c] = (uint32_t)c < nc ? s_x[c][i >> 3] : 0u;
#pragma unroll
        for (int c = 0; c < CT; c++) {
            const uint32_t nib = (xw[c] >> sh) & 15u;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (!((nib >> r) & 1u) || vv[r] == 0) continue;
                if (FIRST) {                      // no prefix yet (the bytes above `shift` are zero in every value)
                    atomicAdd(&h[0][c][cp][d[r]], 1u);
                } else {
                    if (hi[r] == s_p[0][c]) atomicAdd(&h[0][c][cp][d[r]], 1u);
                    if (hi[r] == s_p[1][c]) atomicAdd(&h[NH - 1][c][cp][d[r]], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t c = 0; c < nc; c++) {
        uint32_t *g = hist + (uint64_t)(c0 + c) * 512;
        for (int t = 0; t < NH; t++) {
            uint32_t tot = 0;
            for (int k = 0; k < HCOPY; k++) tot += h[t][c][k][threadIdx.x];
            if (tot) atomicAdd(&g[256 * t + threadIdx.x], tot);
        }
    }
}

// numpy.percentile(..., interpolation='nearest'): index = around(q/100 * (n-1)), half to even
__device__ __forceinline__ unsigned long long nearest_rank(double q, unsigned long long n)
{
    return (unsigned long long)rint((q / 100.0) * (double)(n - 1));
}

__global__ void sel_pick_kernel(int first, double q_lo, double q_hi, SelState *st, uint32_t *hist, uint32_t ncols)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    uint32_t *g = hist + (uint64_t)c * 512;
    SelState s = st[c];
    if (first) {
        unsigned long long n = 0;
        for (int b = 0; b < 256; b++) n += g[b];
        s.n = n;
        s.rank[0] = n ? nearest_rank(q_lo, n) : 0;
        s.rank[1] = n ? nearest_rank(q_hi, n) : 0;
        s.prefix[0] = s.prefix[1] = 0;
    }
    if (s.n) {
        for (int t = 0; t < 2; t++) {
            const uint32_t *h = first ? g : g + 256 * t;
            unsigned long long cum = 0;
            int b = 0;
            for (; b < 255; b++) {
                if (cum + h[b] > s.rank[t]) break;
                cum += h[b];
            }
            s.rank[t] -= cum;
            s.prefix[t] = (s.prefix[t] << 8) | (uint32_t)b;
        }
    }
    st[c] = s;
    for (int b = 0; b < 512; b++) g[b] = 0;
}

// sums over rows with X bit set and lo <= y <= hi (y != 0): count and sum of y (lane = four rows, CT columns per
// workgroup, as above; the sums live in registers)
__global__ __launch_bounds__(NT) void sel_sum_kernel(const uint32_t *__restrict__ x, uint64_t W, uint64_t K,
                                                     const uint32_t *__restrict__ y, const uint32_t *__restrict__ cols, uint32_t ncols,
                                                     const SelState *__restrict__ st, unsigned long long *out /*[ncols][2]*/)
{
    __shared__ uint32_t s_lo[CT], s_hi[CT], s_on[CT];
    __shared__ const uint32_t *s_x[CT];
    const uint32_t c0 = blockIdx.y * CT, nc = min((uint32_t)CT, ncols - c0);
    if (threadIdx.x < CT) {
        const bool in = threadIdx.x < nc;
        s_x[threadIdx.x] = x + (uint64_t)(in ? cols[c0 + threadIdx.x] : 0u) * W;
        s_lo[threadIdx.x] = in ? st[c0 + threadIdx.x].prefix[0] : 1u;
        s_hi[threadIdx.x] = in ? st[c0 + threadIdx.x].prefix[1] : 0u;          // lo > hi: nothing passes
        s_on[threadIdx.x] = in && st[c0 + threadIdx.x].n ? 1u : 0u;
    }
    __syncthreads();
    uint32_t cnt[CT];
    unsigned long long sum[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) { cnt[c] = 0; sum[c] = 0; }
    const uint64_t K4 = (K + 3) >> 2;
    for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < K4; i += (uint64_t)gridDim.x * NT) {
        const uint4 y4 = load_y4(y, i << 2, K);
        const uint32_t vv[4] = {y4.x, y4.y, y4.z, y4.w};
        if (!(vv[0] | vv[1] | vv[2] | vv[3])) continue;
        const uint32_t sh = (uint32_t)(i & 7) * 4u;
#pragma unroll
        for (int c = 0; c < CT; c++) {
            if (!s_on[c]) continue;
            const uint32_t nib = (s_x[c][i >> 3] >> sh) & 15u, lo = s_lo[c], hi = s_hi[c];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t v = vv[r];
                const bool in = ((nib >> r) & 1u) && v != 0 && v >= lo && v <= hi;
                cnt[c] += in ? 1u : 0u;
                sum[c] += in ? v : 0u;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CT; c++) {
        unsigned long long a = cnt[c], b = sum[c];
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off, 64);
            b += __shfl_down(b, off, 64);
        }
        if ((threadIdx.x & 63) == 0 && a && (uint32_t)c < nc) {
            atomicAdd(&out[(uint64_t)(c0 + c) * 2], a);
            atomicAdd(&out[(uint64_t)(c0 + c) * 2 + 1], b);
        }
    }
}

// ---- per-pattern sufficient statistics for the elastic net ------------------------------------
// rows: fold[k] bit 31 = row kept by the filter (:402-415); bit f (< n_folds) = row is in the
// TEST half of fold f.  blockIdx.y = f for f < n_folds (test-half stats) and n_folds = all kept
// rows.  stats[f][m] = {count, sum y, sum y^2} as u64.
__global__ __launch_bounds__(NT) void pattern_stats_kernel(const uint32_t *__restrict__ x, uint64_t W, uint64_t K,
                                                           const uint32_t *__restrict__ cols, int p,
                                                           const uint32_t *__restrict__ y,
                                                           const uint32_t *__restrict__ fold, int n_folds,
                                                           unsigned long long *stats, int use_lds)
{
    extern __shared__ unsigned long long s_acc[];   // [M][3] when use_lds
    const int f = blockIdx.y;
    const uint32_t M = 1u << p;
    const uint32_t want = (f < n_folds) ? ((1u << 31) | (1u << f)) : (1u << 31);
    unsigned long long *g = stats + (uint64_t)f * M * 3;
    if (use_lds) {
        for (uint32_t i = threadIdx.x; i < M * 3; i += NT) s_acc[i] = 0;
        __syncthreads();
    }
    for (uint64_t w = (uint64_t)blockIdx.x * NT + threadIdx.x; w < W; w += (uint64_t)gridDim.x * NT) {
        uint32_t xw[16];
#pragma unroll
        for (int j = 0; j < 16; j++) xw[j] = (j < p) ? x[(uint64_t)cols[j] * W + w] : 0u;
        const uint64_t k0 = w << 5;
        const int nb = (int)min((uint64_t)32, K > k0 ? K - k0 : 0);
        for (int b = 0; b < nb; b++) {
            const uint32_t fb = fold[k0 + b];
            if ((fb & want) != want) continue;
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) m |= ((xw[j] >> b) & 1u) << j;
            const unsigned long long v = y[k0 + b];
            if (use_lds) {
                atomicAdd(&s_acc[m * 3], 1ull);
                if (v) { atomicAdd(&s_acc[m * 3 + 1], v); atomicAdd(&s_acc[m * 3 + 2], v * v); }
            } else {
                atomicAdd(&g[(uint64_t)m * 3], 1ull);
                if (v) { atomicAdd(&g[(uint64_t)m * 3 + 1], v); atomicAdd(&g[(uint64_t)m * 3 + 2], v * v); }
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < M * 3; i += NT)
            if (s_acc[i]) atomicAdd(&g[i], s_acc[i]);
    }
}

// The same statistics for p <= 6 in ONE pass over the rows (the kernel above reads the planes, y and the fold words once
// per fold group: 21 times): lane = row -- fold word and y arrive coalesced, the pattern's p bits come from the two words
// of each plane that the wave's 64 rows share -- and every group the row belongs to (the kept rows, the test halves it is
// in: ~11 of 21) takes its three additions in LDS.  With at most 64 patterns all lanes would hammer a handful of LDS words
// (one wave instruction then takes as many passes as lanes share its word), so the tables are kept in COPIES copies
// (copy = lane mod COPIES) that are added up when the workgroup is done.  [groups][patterns][3] u64 per copy: 32 KB at p = 6.
template <int COPIES>
__global__ __launch_bounds__(NT) void pattern_stats_once_kernel(const uint32_t *__restrict__ x, uint64_t W, uint64_t K,
                                                                const uint32_t *__restrict__ cols, int p,
                                                                const uint32_t *__restrict__ y, const uint32_t *__restrict__ fold,
                                                                int n_folds, unsigned long long *stats)
{
    extern __shared__ unsigned long long s_tab[];       // [COPIES][G][M][3]
    const uint32_t M = 1u << p, G = (uint32_t)n_folds + 1u, per_copy = G * M * 3u;
    for (uint32_t i = threadIdx.x; i < per_copy * COPIES; i += NT) s_tab[i] = 0;
    __syncthreads();
    unsigned long long *mine = s_tab + (threadIdx.x % COPIES) * per_copy;
    const uint32_t fold_mask = n_folds >= 32 ? 0xFFFFFFFFu : ((1u << n_folds) - 1u);
    const uint64_t n_batches = (K + 63) / 64;            // a wave takes 64 rows at a time
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint64_t bt = (uint64_t)blockIdx.x * (NT / 64) + wave; bt < n_batches; bt += (uint64_t)gridDim.x * (NT / 64)) {
        const uint64_t k = bt * 64 + lane;
        uint32_t groups = 0, v = 0, m = 0;
        if (k < K) {
            const uint32_t fb = fold[k];
            if (fb >> 31) groups = (fb & fold_mask) | (1u << n_folds);       // kept: its test halves + "all kept rows"
            v = y[k];
        }
        const uint64_t w = bt * 2 + (lane >> 5);                             // the word of every plane with this row's bit
        for (int j = 0; j < p; j++) m |= ((w < W ? x[(uint64_t)cols[j] * W + w] : 0u) >> (lane & 31) & 1u) << j;
        const unsigned long long vv = (unsigned long long)v * v;
        for (uint32_t f = 0; f < G; f++) {
            if (!((groups >> f) & 1u)) continue;
            unsigned long long *e = mine + ((uint64_t)f * M + m) * 3u;
            atomicAdd(&e[0], 1ull);
            if (v) { atomicAdd(&e[1], (unsigned long long)v); atomicAdd(&e[2], vv); }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < per_copy; i += NT) {
        unsigned long long t = 0;
        for (int c = 0; c < COPIES; c++) t += s_tab[(uint32_t)c * per_copy + i];
        if (t) atomicAdd(&stats[i], t);
    }
}

unsigned grid_for(uint64_t work_items, unsigned rows)
{
    const uint64_t want = (work_items + NT - 1) / NT;
    const uint64_t cap = std::max<uint64_t>(1, (256ull * 8) / std::max(1u, rows));
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, std::max<uint64_t>(cap, 8)));
}

// ---- the O(K) vectors of detect_strains (identify_strains_L2_Enet_Pscan_new_sp.py:191-197, 36-38, 402-415), one lane per
// row: ln = the row of overlap_matrix over the identified clusters' columns, summed, values above one zeroed; py_u = py * ln;
// the bit vectors py > 1 and py_u > 1; the rows the regression keeps (npp25 <= py <= min(npp75, npp_out), compared as doubles
// like numpy does: a NaN bound keeps every row) and their y.  out[0] = kept rows, out[1] = any py_u > 0, out[2] = a count
// that does not fit 32 bits or is negative (the caller raises, as the host path did).
__global__ __launch_bounds__(NT) void l2_prepare_kernel(const long long *__restrict__ y, uint64_t K, const int64_t *__restrict__ om_ptr,
                                                        const int32_t *__restrict__ om_idx, const int8_t *__restrict__ om_val,
                                                        const uint8_t *__restrict__ sel, uint32_t n_cols, double npp25, double npp75,
                                                        double npp_out, uint32_t *__restrict__ y32, uint32_t *__restrict__ yu32,
                                                        uint32_t *__restrict__ G, uint32_t *__restrict__ Gu, uint32_t *__restrict__ keep,
                                                        uint32_t *__restrict__ ykeep, unsigned long long *__restrict__ out, uint64_t W)
{
    const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;          // the grid covers every word of the padded planes
    const bool in = i < K;
    const long long v = in ? y[i] : 0;
    long long ln = 0;
    if (in && om_ptr) {
        for (int64_t j = om_ptr[i], e = om_ptr[i + 1]; j < e; j++) {
            const uint32_t c = (uint32_t)om_idx[j];
            if (c < n_cols) ln += (long long)om_val[j] * sel[c];
        }
    }
    if (ln > 1) ln = 0;
    const bool bad = v < 0 || v > 0xFFFFFFFFll || ln < 0;
    const uint32_t yv = (uint32_t)v, yu = ln == 1 ? yv : 0u;
    const double d = (double)v;
    const bool kp = in && !(d < npp25 || d > npp75 || d > npp_out);
    // all four ballots with the whole wave active: np.sum(py_u) > 0 (:279-286, :331) is over EVERY row
    const uint64_t bg = __ballot(in && yv > 1u), bu = __ballot(yu > 1u), bk = __ballot(kp), bp = __ballot(yu > 0u);
    if (in) { y32[i] = yv; yu32[i] = yu; ykeep[i] = kp ? yv : 0u; }
    const int lane = threadIdx.x & 63;
    if (lane == 0 || lane == 32) {
        const int sh = lane;                                              // 0 or 32
        const uint64_t w = i >> 5;
        if (w < W) {                                                      // W is a multiple of 4, a block covers 8 words
            G[w] = (uint32_t)(bg >> sh); Gu[w] = (uint32_t)(bu >> sh); keep[w] = (uint32_t)(bk >> sh);
        }
    }
    if (lane == 0) {
        if (bk) atomicAdd(&out[0], (unsigned long long)__popcll(bk));
        if (bp) atomicOr(&out[1], 1ull);
    }
    if (bad) atomicOr(&out[2], 1ull);
}

__global__ __launch_bounds__(NT) void l2_popc_words_kernel(const uint32_t *__restrict__ bits, uint64_t W, uint32_t *__restrict__ cnt)
{
    const uint64_t w = (uint64_t)blockIdx.x * NT + threadIdx.x;
    if (w < W) cnt[w] = (uint32_t)__popc(bits[w]);
}
// fold word of row i: 0 when the row is not kept, else bit 31 | the test-fold bits of the row's RANK among the kept rows
// (ShuffleSplit permutes the kept rows: identify_strains_L2_Enet_Pscan_new_sp.py:402-442)
// (inv: 0 when split_bits are the TEST bits; the mask of the folds when they are the TRAINING bits of ss_split_dev_*)
__global__ __launch_bounds__(NT) void l2_fold_kernel(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pre, uint64_t K,
                                                     const uint32_t *__restrict__ split_bits, uint32_t inv, uint32_t *__restrict__ fold)
{
    const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= K) return;
    const uint32_t w = keep[i >> 5], b = (uint32_t)i & 31u;
    fold[i] = ((w >> b) & 1u) ? (((split_bits[pre[i >> 5] + (uint32_t)__popc(w & ((1u << b) - 1u))] ^ inv) & 0x7FFFFFFFu) | 0x80000000u) : 0u;
}

// a CSR row-pointer array as the kernels walk it: starts at 0, never decreases, ends at (and so never exceeds) nnz.
// A truncated or damaged .npz fails here (SS_EINVAL) instead of sending a kernel past the index arrays.
__global__ __launch_bounds__(NT) void csr_ptr_check_kernel(const int64_t *__restrict__ ptr, uint64_t K, int64_t nnz, int *__restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
    if (i > K) return;
    const int64_t a = ptr[i];
    bool b = a < 0 || a > nnz;
    if (i == 0) b = b || a != 0;
    if (i < K) b = b || ptr[i + 1] < a;
    if (b) atomicOr(bad, 1);
}

// bits beyond row K in the last words of a plane (a plane has W words for K rows)
__global__ __launch_bounds__(NT) void planes_padding_check_kernel(const uint32_t *__restrict__ x, uint64_t K, uint32_t S, uint64_t W, int *__restrict__ bad)
{
    const uint32_t s = blockIdx.x * NT + threadIdx.x;
    if (s >= S) return;
    const uint32_t *pl = x + (uint64_t)s * W;
    const uint64_t full = K >> 5, rem = K & 31;
    bool b = rem && (pl[full] >> rem);
    for (uint64_t w = full + (rem ? 1 : 0); w < W; w++) b = b || pl[w] != 0u;
    if (b) atomicOr(bad, 1);
}

int csr_ptr_check(const int64_t *d_ptr, uint64_t K, int64_t nnz)
{
    int *d_bad = nullptr, bad = 0;
    if (nnz < 0) return SS_EINVAL;
    if (ss::l2s::dmalloc((void **)&d_bad, 4) != hipSuccess) return SS_ENOMEM;
    int rc = SS_OK;
    if (ss::l2s::set(d_bad, 0, 4) != hipSuccess) rc = SS_EHIP;
    else {
        hipLaunchKernelGGL(csr_ptr_check_kernel, dim3((unsigned)((K + 1 + NT - 1) / NT)), dim3(NT), 0, ss::l2s::stream(), d_ptr, K, nnz, d_bad);
        if (hipGetLastError() != hipSuccess || ss::l2s::copy(&bad, d_bad, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = SS_EHIP;
        else if (bad) rc = SS_EINVAL;
    }
    ss::l2s::dfree(d_bad);
    return rc;
}

}  // namespace

extern "C" {

int ss_l2_set_overlap(ss_l2 *h, const int64_t *indptr, const int32_t *indices, const int8_t *data, uint32_t n_cols)
{
    if (!h || !indptr || indptr[h->K] < 0) return SS_EINVAL;
    const uint64_t nnz = (uint64_t)indptr[h->K];
    if (nnz && (!indices || !data)) return SS_EINVAL;
    if (h->d_blob) return SS_EINVAL;      // a handle from ss_l2_import: its overlap arrays live inside the image's one allocation
    hipFree(h->d_om_ptr); hipFree(h->d_om_idx); hipFree(h->d_om_val);
    h->d_om_ptr = nullptr; h->d_om_idx = nullptr; h->d_om_val = nullptr; h->has_om = false;
    SS_HIP(hipMalloc((void **)&h->d_om_ptr, (h->K + 1) * 8));
    SS_HIP(hipMalloc((void **)&h->d_om_idx, std::max<uint64_t>(1, nnz) * 4));
    SS_HIP(hipMalloc((void **)&h->d_om_val, std::max<uint64_t>(1, nnz)));
    SS_HIP(ss::l2s::copy(h->d_om_ptr, indptr, (h->K + 1) * 8, hipMemcpyHostToDevice));
    if (const int rc = csr_ptr_check(h->d_om_ptr, h->K, (int64_t)nnz)) {
        hipFree(h->d_om_ptr); hipFree(h->d_om_idx); hipFree(h->d_om_val);
        h->d_om_ptr = nullptr; h->d_om_idx = nullptr; h->d_om_val = nullptr;
        return rc;
    }
    if (nnz) {
        SS_HIP(ss::l2s::copy(h->d_om_idx, indices, nnz * 4, hipMemcpyHostToDevice));
        SS_HIP(ss::l2s::copy(h->d_om_val, data, nnz, hipMemcpyHostToDevice));
    }
    h->om_cols = n_cols;
    h->has_om = true;
    return SS_OK;
}

int ss_l2_prepare(const ss_l2 *h, const int64_t *y_host, const uint8_t *col_sel, double npp25, double npp75, double npp_out,
                  uint32_t *y_dev, uint32_t *yu_dev, uint32_t *G_dev, uint32_t *Gu_dev, uint32_t *keep_dev, uint32_t *ykeep_dev,
                  uint64_t out[3])
{
    if (!h || !h->has_om || !out || (h->K && (!y_host || !y_dev || !yu_dev || !G_dev || !Gu_dev || !keep_dev || !ykeep_dev))) return SS_EINVAL;
    if (h->om_cols && !col_sel) return SS_EINVAL;
    out[0] = out[1] = out[2] = 0;
    if (!h->K) return SS_OK;
    long long *d_y = nullptr;
    uint8_t *d_sel = nullptr;
    unsigned long long *d_out = nullptr;
    int rc = SS_OK;
    if (ss::l2s::dmalloc((void **)&d_y, h->K * 8) != hipSuccess || ss::l2s::dmalloc((void **)&d_sel, std::max<uint32_t>(1, h->om_cols)) != hipSuccess ||
        ss::l2s::dmalloc((void **)&d_out, 24) != hipSuccess)
        rc = SS_ENOMEM;
    hipError_t e = hipSuccess;
    if (!rc) {
        e = ss::l2s::copy(d_y, y_host, h->K * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess && h->om_cols) e = ss::l2s::copy(d_sel, col_sel, h->om_cols, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = ss::l2s::set(d_out, 0, 24);
        if (e == hipSuccess) {
            const uint64_t rows = h->W * 32;                                // every word of the padded bit vectors is written
            hipLaunchKernelGGL(l2_prepare_kernel, dim3((unsigned)((rows + NT - 1) / NT)), dim3(NT), 0, ss::l2s::stream(), d_y, h->K, h->d_om_ptr, h->d_om_idx,
                               h->d_om_val, d_sel, h->om_cols, npp25, npp75, npp_out, y_dev, yu_dev, G_dev, Gu_dev, keep_dev, ykeep_dev, d_out, h->W);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = ss::l2s::copy(out, d_out, 24, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { ss::set_last_error("ss_l2_prepare", __FILE__, __LINE__, e); rc = SS_EHIP; }
    }
    ss::l2s::dfree(d_y); ss::l2s::dfree(d_sel); ss::l2s::dfree(d_out);
    return rc;
}

// the row filter of l2_prepare_kernel (same comparison, as doubles) counted on host threads
int ss_l2_count_keep(const int64_t *y_host, uint64_t K, double npp25, double npp75, double npp_out, uint64_t *n_keep)
{
    if (!n_keep || (K && !y_host)) return SS_EINVAL;
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min(8u, ss::host_cpus()), K >> 18));
    std::vector<uint64_t> part(nt, 0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            uint64_t c = 0;
            for (uint64_t i = K * t / nt, e = K * (t + 1) / nt; i < e; i++) {
                const double d = (double)y_host[i];
                c += !(d < npp25 || d > npp75 || d > npp_out);
            }
            part[t] = c;
        });
    for (auto &x : th) x.join();
    uint64_t total = 0;
    for (uint64_t c : part) total += c;
    *n_keep = total;
    return SS_OK;
}

static int l2_fold_impl(const ss_l2 *h, const uint32_t *keep_dev, const uint32_t *split_bits, bool bits_on_device, uint32_t inv, uint64_t n_keep,
                        uint32_t *fold_dev);

int ss_l2_fold(const ss_l2 *h, const uint32_t *keep_dev, const uint32_t *split_bits, uint64_t n_keep, uint32_t *fold_dev)
{
    return l2_fold_impl(h, keep_dev, split_bits, false, 0u, n_keep, fold_dev);
}

int ss_l2_fold_train(const ss_l2 *h, const uint32_t *keep_dev, const uint32_t *train_bits_dev, uint64_t n_keep, int n_splits, uint32_t *fold_dev)
{
    if (n_splits < 1 || n_splits > 31) return SS_EINVAL;
    return l2_fold_impl(h, keep_dev, train_bits_dev, true, (1u << n_splits) - 1u, n_keep, fold_dev);
}

static int l2_fold_impl(const ss_l2 *h, const uint32_t *keep_dev, const uint32_t *split_bits, bool bits_on_device, uint32_t inv, uint64_t n_keep,
                        uint32_t *fold_dev)
{
    if (!h || (h->K && (!keep_dev || !fold_dev)) || (n_keep && !split_bits) || n_keep > h->K) return SS_EINVAL;
    if (!h->K) return SS_OK;
    uint32_t *d_cnt = nullptr, *d_pre = nullptr, *d_bits = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    int rc = SS_OK;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_cnt, d_pre, (int)h->W);
    if (e != hipSuccess || ss::l2s::dmalloc((void **)&d_cnt, h->W * 4) != hipSuccess || ss::l2s::dmalloc((void **)&d_pre, h->W * 4) != hipSuccess ||
        (!bits_on_device && ss::l2s::dmalloc((void **)&d_bits, std::max<uint64_t>(1, n_keep) * 4) != hipSuccess) ||
        ss::l2s::dmalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)) != hipSuccess)
        rc = SS_ENOMEM;
    if (!rc) {
        if (n_keep && !bits_on_device) e = ss::l2s::copy(d_bits, split_bits, n_keep * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(l2_popc_words_kernel, dim3((unsigned)((h->W + NT - 1) / NT)), dim3(NT), 0, ss::l2s::stream(), keep_dev, h->W, d_cnt);
            e = hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_cnt, d_pre, (int)h->W, ss::l2s::stream());
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(l2_fold_kernel, dim3((unsigned)((h->K + NT - 1) / NT)), dim3(NT), 0, ss::l2s::stream(), keep_dev, d_pre, h->K,
                               bits_on_device ? split_bits : d_bits, inv, fold_dev);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = ss::l2s::sync();
        if (e != hipSuccess) { ss::set_last_error("ss_l2_fold", __FILE__, __LINE__, e); rc = SS_EHIP; }
    }
    ss::l2s::dfree(d_cnt); ss::l2s::dfree(d_pre); ss::l2s::dfree(d_bits); ss::l2s::dfree(d_tmp);
    return rc;
}

static int l2_create_impl(const int64_t *indptr, const int32_t *indices, bool indices_on_device, uint64_t K, uint32_t S, ss_l2 **out);

int ss_l2_create(const int64_t *indptr, const int32_t *indices, uint64_t K, uint32_t S, ss_l2 **out)
{
    return l2_create_impl(indptr, indices, false, K, S, out);
}

/* the same with the column indices ALREADY on the device (the inflated `indices.npy` member of all_strains_re.npz,
 * ss_npz_member_dev: 2.5 GB for a 5 M x 300 cluster that neither visits the host nor crosses PCIe uncompressed) */
int ss_l2_create_dev(const int64_t *indptr, const int32_t *indices_dev, uint64_t K, uint32_t S, ss_l2 **out)
{
    if (indices_dev && (((uintptr_t)indices_dev) & 3u)) return SS_EINVAL;
    return l2_create_impl(indptr, indices_dev, true, K, S, out);
}

static int l2_create_impl(const int64_t *indptr, const int32_t *indices, bool indices_on_device, uint64_t K, uint32_t S, ss_l2 **out)
{
    if (!out || !indptr || indptr[K] < 0 || (indptr[K] && !indices)) return SS_EINVAL;
    ss_l2 *h = new (std::nothrow) ss_l2();
    if (!h) return SS_ENOMEM;
    h->K = K;
    h->S = S;
    h->W = ((K + 31) / 32 + 3) & ~3ull;
    if (h->W == 0) h->W = 4;
    const uint64_t nnz = (uint64_t)indptr[K];
    int64_t *d_ptr = nullptr;
    int32_t *d_idx = nullptr;
    int *d_bad = nullptr;
    int rc = SS_OK, bad = 0;
    const uint64_t xbytes = std::max<uint64_t>(1, (uint64_t)S) * h->W * 4;
    if (hipMalloc((void **)&h->d_x, xbytes) != hipSuccess || ss::l2s::dmalloc((void **)&d_ptr, (K + 1) * 8) != hipSuccess ||
        (!indices_on_device && ss::l2s::dmalloc((void **)&d_idx, std::max<uint64_t>(1, nnz) * 4) != hipSuccess) ||
        ss::l2s::dmalloc((void **)&d_bad, 4) != hipSuccess) {
        rc = SS_ENOMEM;
    } else if (ss::l2s::set(h->d_x, 0, xbytes) != hipSuccess || ss::l2s::set(d_bad, 0, 4) != hipSuccess ||
               ss::l2s::copy(d_ptr, indptr, (K + 1) * 8, hipMemcpyHostToDevice) != hipSuccess ||
               (nnz && !indices_on_device && ss::l2s::copy(d_idx, indices, nnz * 4, hipMemcpyHostToDevice) != hipSuccess)) {
        rc = SS_EHIP;
    } else if ((rc = csr_ptr_check(d_ptr, K, (int64_t)nnz)) != SS_OK) {
        // (a row pointer array the pack kernels could not walk safely)
    } else if (K) {
        if (indices_on_device) d_idx = const_cast<int32_t *>(indices);                     // (the caller's: read only, not freed here)
        const bool force_atomic = getenv("SS_L2_PACK_ATOMIC") != nullptr;                 // tests: the general kernel
        int redo = 1;
        const bool no_span = getenv("SS_L2_PACK_WALK") != nullptr;                    // A/B and tests: the row-walk kernel
        if (!force_atomic && !no_span && S <= PACK_SMAX) {
            hipLaunchKernelGGL(pack_csr_span_kernel, dim3((unsigned)((K + 63) / 64)), dim3(64), (size_t)S * 8 + 65 * 8, ss::l2s::stream(), d_ptr, d_idx, K, S,
                               h->W, h->d_x, d_bad);
            if (hipGetLastError() != hipSuccess || ss::l2s::copy(&bad, d_bad, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = SS_EHIP;
            else if (bad) rc = SS_EINVAL;
            redo = 0;
        } else if (!force_atomic) {
            hipLaunchKernelGGL(pack_csr_sorted_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, ss::l2s::stream(), d_ptr, d_idx, K, S,
                               h->W, h->d_x, d_bad);
            if (ss::l2s::copy(&redo, d_bad, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = SS_EHIP;
        }
        if (!rc && redo) {
            ss::l2s::set(d_bad, 0, 4);
            ss::l2s::set(h->d_x, 0, xbytes);
            hipLaunchKernelGGL(pack_csr_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, ss::l2s::stream(), d_ptr, d_idx, K, S,
                               h->W, h->d_x, d_bad);
            if (ss::l2s::copy(&bad, d_bad, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = SS_EHIP;
            else if (bad) rc = SS_EINVAL;
        }
    }
    if (indices_on_device) { d_idx = nullptr; if (ss::l2s::sync() != hipSuccess && !rc) rc = SS_EHIP; }      // (the pack kernels are done with the caller's array)
    ss::l2s::dfree(d_ptr); ss::l2s::dfree(d_idx); ss::l2s::dfree(d_bad);
    if (rc) { hipFree(h->d_x); delete h; return rc; }
    *out = h;
    return SS_OK;
}

int ss_l2_create_planes(const uint32_t *planes, uint64_t K, uint32_t S, ss_l2 **out)
{
    if (!out || (!planes && K && S)) return SS_EINVAL;
    ss_l2 *h = new (std::nothrow) ss_l2();
    if (!h) return SS_ENOMEM;
    h->K = K;
    h->S = S;
    h->W = ((K + 31) / 32 + 3) & ~3ull;
    if (h->W == 0) h->W = 4;
    const uint64_t words = (uint64_t)S * h->W, xbytes = std::max<uint64_t>(1, (uint64_t)S) * h->W * 4;
    // the padding bits of every plane must be zero (popcounts run over whole words)
    const uint64_t full = K >> 5, rem = K & 31;
    for (uint32_t s = 0; s < S && K; s++) {
        const uint32_t *pl = planes + (uint64_t)s * h->W;
        if (rem && (pl[full] >> rem)) { delete h; return SS_EINVAL; }
        for (uint64_t w = full + (rem ? 1 : 0); w < h->W; w++)
            if (pl[w]) { delete h; return SS_EINVAL; }
    }
    if (hipMalloc((void **)&h->d_x, xbytes) != hipSuccess) { delete h; return SS_ENOMEM; }
    if ((K == 0 && ss::l2s::set(h->d_x, 0, xbytes) != hipSuccess) ||
        (K && words && ss::l2s::copy(h->d_x, planes, words * 4, hipMemcpyHostToDevice) != hipSuccess)) {
        hipFree(h->d_x);
        delete h;
        return SS_EHIP;
    }
    *out = h;
    return SS_OK;
}

int ss_l2_export_planes(const ss_l2 *h, uint32_t *planes)
{
    if (!h || (!planes && h->S)) return SS_EINVAL;
    if (h->S) SS_HIP(ss::l2s::copy(planes, h->d_x, (uint64_t)h->S * h->W * 4, hipMemcpyDeviceToHost));
    return SS_OK;
}

int ss_l2_destroy(ss_l2 *h)
{
    if (!h) return SS_OK;
    if (h->d_blob) hipFree(h->d_blob);
    else { hipFree(h->d_x); hipFree(h->d_om_ptr); hipFree(h->d_om_idx); hipFree(h->d_om_val); }
    delete h;
    return SS_OK;
}

// The cluster image file of strainscan_amd (identify_strains_L2_Enet_Pscan_new_sp.py _write_l2_cache: bit planes, then the
// overlap matrix's CSR arrays, each at a 64-byte boundary of the file) straight to the device: the whole file travels through
// the pinned upload buffers of the .gz path (ss_ginflate.hip upload_file: four threads, ~40 GB/s from the page cache) into ONE
// allocation, and the handle's pointers point into it.  Through a memory map and pageable copies the 253 MB of a 5 M x 300
// cluster took 25 ms of the 70 its vote takes; this way ~7.  The caller gives the layout it read from the file's header;
// everything the other constructors check is checked here too (padding bits of the planes, row pointers, sizes).
int ss_l2_import(const char *path, uint64_t K, uint32_t S, uint64_t off_planes, uint64_t off_ptr, uint64_t off_idx, uint64_t off_val,
                 uint64_t nnz, uint32_t n_cols, ss_l2 **out)
{
    if (!path || !out || K > 0x7FFFFFFFFFull) return SS_EINVAL;
    const uint64_t W = std::max<uint64_t>(4, ((K + 31) / 32 + 3) & ~3ull);
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return SS_EIO;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { close(fd); return SS_EIO; }
    const uint64_t n = (uint64_t)sb.st_size;
    auto inside = [&](uint64_t off, uint64_t bytes) { return (off & 63) == 0 && off <= n && bytes <= n - off; };
    if (!inside(off_planes, (uint64_t)S * W * 4) || !inside(off_ptr, (K + 1) * 8) || !inside(off_idx, nnz * 4) || !inside(off_val, nnz)) { close(fd); return SS_EINVAL; }
    ss_l2 *h = new (std::nothrow) ss_l2();
    if (!h) { close(fd); return SS_ENOMEM; }
    h->K = K; h->S = S; h->W = W;
    int rc = SS_OK;
    if (hipMalloc((void **)&h->d_blob, n + 64) != hipSuccess) rc = SS_ENOMEM;
    else if (!ss::upload_file_to_device(fd, n, h->d_blob)) rc = SS_EIO;
    close(fd);
    if (!rc) {
        h->d_x = reinterpret_cast<uint32_t *>(h->d_blob + off_planes);
        h->d_om_ptr = reinterpret_cast<int64_t *>(h->d_blob + off_ptr);
        h->d_om_idx = reinterpret_cast<int32_t *>(h->d_blob + off_idx);
        h->d_om_val = reinterpret_cast<int8_t *>(h->d_blob + off_val);
        h->om_cols = n_cols;
        // the padding bits of every plane must be zero (popcounts run over whole words), the row pointers in order and ending at nnz
        int *d_bad = nullptr, bad = 0;
        if (ss::l2s::dmalloc((void **)&d_bad, 4) != hipSuccess) rc = SS_ENOMEM;
        else {
            ss::l2s::set(d_bad, 0, 4);
            if (S && K) hipLaunchKernelGGL(planes_padding_check_kernel, dim3((unsigned)((S + NT - 1) / NT)), dim3(NT), 0, ss::l2s::stream(), h->d_x, K, S, W, d_bad);
            hipLaunchKernelGGL(csr_ptr_check_kernel, dim3((unsigned)((K + 1 + NT - 1) / NT)), dim3(NT), 0, ss::l2s::stream(), h->d_om_ptr, K, (int64_t)nnz, d_bad);
            if (hipGetLastError() != hipSuccess || ss::l2s::copy(&bad, d_bad, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = SS_EHIP;
            else if (bad) rc = SS_EINVAL;
            ss::l2s::dfree(d_bad);
        }
        if (!rc) {
            int64_t last = 0;                                  // (csr_ptr_check_kernel: never above nnz; here: exactly nnz)
            if (ss::l2s::copy(&last, h->d_om_ptr + K, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = SS_EHIP;
            else if ((uint64_t)last != nnz) rc = SS_EINVAL;
        }
        h->has_om = rc == SS_OK;
    }
    if (rc) { hipFree(h->d_blob); delete h; return rc; }
    *out = h;
    return SS_OK;
}

int ss_l2_info(const ss_l2 *h, uint64_t *K, uint32_t *S, uint64_t *words_per_plane)
{
    if (!h) return SS_EINVAL;
    if (K) *K = h->K;
    if (S) *S = h->S;
    if (words_per_plane) *words_per_plane = h->W;
    return SS_OK;
}

int ss_l2_popc2(const ss_l2 *h, const uint32_t *A_dev, const uint32_t *B_dev, uint64_t *out1, uint64_t *out2)
{
    if (!h || !out1 || !out2) return SS_EINVAL;
    if (!h->S) return SS_OK;
    unsigned long long *d = nullptr;
    SS_HIP(ss::l2s::dmalloc((void **)&d, (uint64_t)h->S * 16));
    ss::l2s::set(d, 0, (uint64_t)h->S * 16);
    hipLaunchKernelGGL(popc2_kernel, dim3(grid_for(h->W / 4, h->S), h->S), dim3(NT), 0, ss::l2s::stream(), h->d_x, h->W, A_dev, B_dev,
                       d, d + h->S);
    hipError_t e = ss::l2s::copy(out1, d, (uint64_t)h->S * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = ss::l2s::copy(out2, d + h->S, (uint64_t)h->S * 8, hipMemcpyDeviceToHost);
    ss::l2s::dfree(d);
    if (e != hipSuccess) { ss::set_last_error("ss_l2_popc2", __FILE__, __LINE__, e); return SS_EHIP; }
    return SS_OK;
}

int ss_l2_andnot_col(const ss_l2 *h, uint32_t col, uint32_t *nu_dev)
{
    if (!h || !nu_dev || col >= h->S) return SS_EINVAL;
    hipLaunchKernelGGL(andnot_col_kernel, dim3((unsigned)std::min<uint64_t>((h->W + 255) / 256, 2048)), dim3(256), 0, ss::l2s::stream(),
                       h->d_x, h->W, col, nu_dev);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int ss_l2_quantile_sums(const ss_l2 *h, const uint32_t *y_dev, const uint32_t *cols, uint32_t ncols, double q_lo,
                        double q_hi, uint64_t *n_nz, uint32_t *v_lo, uint32_t *v_hi, uint64_t *cnt_in,
                        uint64_t *sum_in)
{
    if (!h || !y_dev || (ncols && !cols)) return SS_EINVAL;
    if (!ncols) return SS_OK;
    for (uint32_t i = 0; i < ncols; i++) if (cols[i] >= h->S) return SS_EINVAL;
    uint32_t *d_cols = nullptr, *d_hist = nullptr;
    SelState *d_st = nullptr;
    unsigned long long *d_out = nullptr;
    int rc = SS_OK;
    if (ss::l2s::dmalloc((void **)&d_cols, ncols * 4) != hipSuccess || ss::l2s::dmalloc((void **)&d_hist, (uint64_t)ncols * 2048) != hipSuccess ||
        ss::l2s::dmalloc((void **)&d_st, ncols * sizeof(SelState)) != hipSuccess ||
        ss::l2s::dmalloc((void **)&d_out, (uint64_t)ncols * 16) != hipSuccess)
        rc = SS_ENOMEM;
    if (!rc) {
        ss::l2s::copy(d_cols, cols, ncols * 4, hipMemcpyHostToDevice);
        ss::l2s::set(d_hist, 0, (uint64_t)ncols * 2048);
        ss::l2s::set(d_st, 0, ncols * sizeof(SelState));
        ss::l2s::set(d_out, 0, (uint64_t)ncols * 16);
        const unsigned ctiles = (ncols + CT - 1) / CT;
        const dim3 grid(grid_for((h->K + 3) / 4, ctiles), ctiles);      // one lane per four rows, CT columns per workgroup
        // the radix passes start at the highest byte that is non-zero in any value (k-mer counts are small numbers:
        // usually ONE pass instead of four, each of which reads the bit planes and gathers y for every set bit)
        uint32_t *d_max = reinterpret_cast<uint32_t *>(d_out), ymax = 0;       // d_out is zero and unused until sel_sum_kernel
        hipLaunchKernelGGL(max_u32_kernel, dim3((unsigned)std::min<uint64_t>((h->K + 1023) / 1024, 512)), dim3(256), 0, ss::l2s::stream(), y_dev, h->K, d_max);
        hipError_t e0 = hipGetLastError();
        if (e0 == hipSuccess) e0 = ss::l2s::copy(&ymax, d_max, 4, hipMemcpyDeviceToHost);
        if (e0 == hipSuccess) e0 = ss::l2s::set(d_max, 0, 4);
        if (e0 != hipSuccess) {             // without the maximum the passes cannot be shortened safely: report, do not guess
            ss::set_last_error("ss_l2_quantile_sums (max of y)", __FILE__, __LINE__, e0);
            ss::l2s::dfree(d_cols); ss::l2s::dfree(d_hist); ss::l2s::dfree(d_st); ss::l2s::dfree(d_out);
            return SS_EHIP;
        }
        const int top = ymax >> 24 ? 24 : ymax >> 16 ? 16 : ymax >> 8 ? 8 : 0;
        for (int shift = top; shift >= 0; shift -= 8) {
            if (shift == top) hipLaunchKernelGGL((sel_hist_kernel<true>), grid, dim3(NT), 0, ss::l2s::stream(), h->d_x, h->W, h->K, y_dev, d_cols, ncols, shift, d_st, d_hist);
            else hipLaunchKernelGGL((sel_hist_kernel<false>), grid, dim3(NT), 0, ss::l2s::stream(), h->d_x, h->W, h->K, y_dev, d_cols, ncols, shift, d_st, d_hist);
            hipLaunchKernelGGL(sel_pick_kernel, dim3((ncols + 63) / 64), dim3(64), 0, ss::l2s::stream(), (int)(shift == top), q_lo, q_hi, d_st, d_hist, ncols);
        }
        hipLaunchKernelGGL(sel_sum_kernel, grid, dim3(NT), 0, ss::l2s::stream(), h->d_x, h->W, h->K, y_dev, d_cols, ncols, d_st, d_out);
        std::vector<SelState> st(ncols);
        std::vector<unsigned long long> o((size_t)ncols * 2);
        hipError_t e = ss::l2s::copy(st.data(), d_st, ncols * sizeof(SelState), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = ss::l2s::copy(o.data(), d_out, (uint64_t)ncols * 16, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { ss::set_last_error("ss_l2_quantile_sums", __FILE__, __LINE__, e); rc = SS_EHIP; }
        else
            for (uint32_t i = 0; i < ncols; i++) {
                if (n_nz) n_nz[i] = st[i].n;
                if (v_lo) v_lo[i] = st[i].prefix[0];
                if (v_hi) v_hi[i] = st[i].prefix[1];
                if (cnt_in) cnt_in[i] = o[(size_t)i * 2];
                if (sum_in) sum_in[i] = o[(size_t)i * 2 + 1];
            }
    }
    ss::l2s::dfree(d_cols); ss::l2s::dfree(d_hist); ss::l2s::dfree(d_st); ss::l2s::dfree(d_out);
    return rc;
}

int ss_l2_pattern_stats(const ss_l2 *h, const uint32_t *cols, int p, const uint32_t *y_dev,
                        const uint32_t *fold_dev, int n_folds, uint64_t *stats /* [(n_folds+1)][2^p][3] host */)
{
    if (!h || !cols || !y_dev || !fold_dev || !stats) return SS_EINVAL;
    if (p < 1 || p > 16 || n_folds < 0 || n_folds > 30) return SS_ERANGE;      // bit 31 of a fold word = "row kept"; 30 folds at most
    for (int i = 0; i < p; i++) if (cols[i] >= h->S) return SS_EINVAL;
    const uint64_t M = 1ull << p;
    const uint64_t n = (uint64_t)(n_folds + 1) * M * 3;
    uint32_t *d_cols = nullptr;
    unsigned long long *d_stats = nullptr;
    SS_HIP(ss::l2s::dmalloc((void **)&d_cols, 16 * 4));
    if (ss::l2s::dmalloc((void **)&d_stats, n * 8) != hipSuccess) { ss::l2s::dfree(d_cols); return SS_ENOMEM; }
    uint32_t c16[16] = {0};
    for (int i = 0; i < p; i++) c16[i] = cols[i];
    ss::l2s::copy(d_cols, c16, 64, hipMemcpyHostToDevice);
    ss::l2s::set(d_stats, 0, n * 8);
    const size_t once_bytes = (size_t)(n_folds + 1) * M * 24;                        // one copy of the tables of the one-pass kernel
    if (p <= 6 && once_bytes <= 32 * 1024) {
        const unsigned blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(512, (h->K + 64 * (NT / 64) - 1) / (64 * (NT / 64))));
        const int copies = once_bytes * 8 <= 64 * 1024 ? 8 : once_bytes * 4 <= 64 * 1024 ? 4 : 2;
#define SS_ONCE(C) hipLaunchKernelGGL((pattern_stats_once_kernel<C>), dim3(blocks), dim3(NT), once_bytes * C, ss::l2s::stream(), h->d_x, h->W, h->K, d_cols, \
                                      p, y_dev, fold_dev, n_folds, d_stats)
        if (copies == 8) SS_ONCE(8); else if (copies == 4) SS_ONCE(4); else SS_ONCE(2);
#undef SS_ONCE
    } else {
        const int use_lds = (M * 24 <= 48 * 1024) ? 1 : 0;   // p <= 11
        const size_t lds = use_lds ? (size_t)M * 24 : 0;
        hipLaunchKernelGGL(pattern_stats_kernel, dim3(grid_for(h->W, (unsigned)n_folds + 1), (unsigned)n_folds + 1), dim3(NT),
                           lds, ss::l2s::stream(), h->d_x, h->W, h->K, d_cols, p, y_dev, fold_dev, n_folds, d_stats, use_lds);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = ss::l2s::copy(stats, d_stats, n * 8, hipMemcpyDeviceToHost);
    ss::l2s::dfree(d_cols); ss::l2s::dfree(d_stats);
    if (e != hipSuccess) { ss::set_last_error("ss_l2_pattern_stats", __FILE__, __LINE__, e); return SS_EHIP; }
    return SS_OK;
}

}  // extern "C"
