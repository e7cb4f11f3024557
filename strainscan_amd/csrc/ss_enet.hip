// ss_enet.hip -- positive elastic-net coordinate descent on Gram statistics, one workgroup per
// cross-validation fold.
//
// Replaces the solver the reference borrows from scikit-learn (pinned 0.23.1):
//   ElasticNetCV(...).fit  -> _path_residuals -> enet_path -> _cd_fast.enet_coordinate_descent_gram
//   ElasticNet(...).fit    -> _cd_fast.enet_coordinate_descent
// (call sites library/identify_strains_L2_Enet_Pscan_new_sp.py:437-455).
//
// The design matrix has p <= 16 binary columns, so each fold is a p x p problem: Q = Xt'Xt,
// q = Xt'yt, yy = yt'yt (exact integers, built from the per-pattern statistics of ss_l2.hip).
// One wave per fold: lane j owns H[j] = (Q w)[j]; the cyclic sweep is sequential over
// coordinates exactly like the Cython loop (same update order, same stopping rule: duality gap
// evaluated only when max|dw|/max|w| < tol), so the iterates follow scikit-learn's to rounding.
// The plain (precompute=False) solver's update and gap are algebraically the Gram ones
// (R'R = yy - 2 q'w + w'Qw, R'y = yy - q'w, X'R = q - Qw), so the refit uses the same kernel.
// Test-fold MSE: mean((X_te w - y_te)^2) = sum_m [c_m pred_m^2 - 2 pred_m s_m + t_m] / n_te.
#include "ss_common.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace {

constexpr int MAXP = 16;

struct FoldIn {
    double Q[MAXP * MAXP];
    double q[MAXP];
    double yy;
    double n_train;
    double n_test;
};

__global__ __launch_bounds__(64) void enet_path_kernel(const FoldIn *__restrict__ in, int p,
                                                       const double *__restrict__ alphas, int n_alphas,
                                                       double l1_ratio, int max_iter, double tol_in, int positive,
                                                       const unsigned long long *__restrict__ test_stats, uint32_t M,
                                                       double *__restrict__ mse /*[n_alphas][F]*/,
                                                       double *__restrict__ coefs /*[F][n_alphas][p]*/,
                                                       int *__restrict__ iters /*[F][n_alphas]*/,
                                                       double *__restrict__ gaps /*[F][n_alphas]*/)
{
    __shared__ double sQ[MAXP * MAXP];
    __shared__ double sw[MAXP], sH[MAXP], sq[MAXP];
    __shared__ int s_flag;
    const int f = blockIdx.x, F = gridDim.x;
    const int lane = threadIdx.x;
    const FoldIn &I = in[f];
    for (int i = lane; i < p * p; i += 64) sQ[i] = I.Q[i];
    if (lane < p) { sq[lane] = I.q[lane]; sw[lane] = 0.0; sH[lane] = 0.0; }
    __syncthreads();
    const double yy = I.yy;
    const double tol = tol_in * yy;
    const double myq = (lane < p) ? sq[lane] : 0.0;
    double H = 0.0;   // (Q w)[lane]; w starts at 0 and is warm-started along the path
    double w = 0.0;   // w[lane]

    for (int a = 0; a < n_alphas; a++) {
        const double l1 = alphas[a] * l1_ratio * I.n_train;
        const double l2 = alphas[a] * (1.0 - l1_ratio) * I.n_train;
        int n_iter = 0;
        for (n_iter = 0; n_iter < max_iter; n_iter++) {
            double w_max = 0.0, d_w_max = 0.0;
            for (int ii = 0; ii < p; ii++) {
                const double Qii = sQ[ii * p + ii];
                if (Qii == 0.0) continue;
                const double w_ii = __shfl(w, ii, 64);
                const double Qrow = (lane < p) ? sQ[ii * p + lane] : 0.0;
                if (w_ii != 0.0) H -= w_ii * Qrow;
                const double tmp = __shfl(myq - H, ii, 64);
                double nw;
                if (positive && tmp < 0) nw = 0.0;
                else {
                    const double sg = (tmp > 0) - (tmp < 0);
                    nw = sg * fmax(fabs(tmp) - l1, 0.0) / (Qii + l2);
                }
                if (nw != 0.0) H += nw * Qrow;
                if (lane == ii) w = nw;
                const double d = fabs(nw - w_ii);
                if (d > d_w_max) d_w_max = d;
                if (fabs(nw) > w_max) w_max = fabs(nw);
            }
            if (w_max == 0.0 || d_w_max / w_max < tol_in || n_iter == max_iter - 1) {
                // duality gap, evaluated sequentially by lane 0 in the Cython order
                if (lane < p) { sw[lane] = w; sH[lane] = H; }
                __syncthreads();
                if (lane == 0) {
                    double q_dot_w = 0.0;
                    for (int i = 0; i < p; i++) q_dot_w += sw[i] * sq[i];
                    double dual = 0.0;
                    for (int i = 0; i < p; i++) {
                        const double xta = sq[i] - sH[i] - l2 * sw[i];
                        const double v = positive ? xta : fabs(xta);
                        if (i == 0 || v > dual) dual = v;
                    }
                    double t2 = 0.0;
                    for (int i = 0; i < p; i++) t2 += sw[i] * sH[i];
                    const double R2 = yy + t2 - 2.0 * q_dot_w;
                    double w2 = 0.0, wl1 = 0.0;
                    for (int i = 0; i < p; i++) { w2 += sw[i] * sw[i]; wl1 += fabs(sw[i]); }
                    double c, g;
                    if (dual > l1) { c = l1 / dual; g = 0.5 * (R2 + R2 * (c * c)); }
                    else { c = 1.0; g = R2; }
                    g += l1 * wl1 - c * yy + c * q_dot_w + 0.5 * l2 * (1 + c * c) * w2;
                    s_flag = (g < tol) ? 1 : 0;
                    if (gaps) gaps[(size_t)f * n_alphas + a] = g;
                }
                __syncthreads();
                if (s_flag) break;
            }
        }
        if (lane < p) sw[lane] = w;
        __syncthreads();
        if (coefs && lane < p) coefs[((size_t)f * n_alphas + a) * p + lane] = w;
        if (iters && lane == 0) iters[(size_t)f * n_alphas + a] = (n_iter < max_iter) ? n_iter + 1 : max_iter;
        if (test_stats) {
            const unsigned long long *ts = test_stats + (size_t)f * M * 3;
            double acc = 0.0;
            for (uint32_t m = lane; m < M; m += 64) {
                const unsigned long long c = ts[(size_t)m * 3];
                if (!c) continue;
                double pred = 0.0;
                for (int j = 0; j < p; j++)
                    if ((m >> j) & 1u) pred += sw[j];
                const double s = (double)ts[(size_t)m * 3 + 1], t = (double)ts[(size_t)m * 3 + 2];
                acc += (double)c * pred * pred - 2.0 * pred * s + t;
            }
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
            if (lane == 0) mse[(size_t)a * F + f] = acc / I.n_test;
        }
        __syncthreads();
    }
}

// ---- the residual form: R = y - X w on the device beside X ---------------------------------------------------------
// _cd_fast.enet_coordinate_descent (ElasticNet(precompute=False).fit, identify_strains...:451-455) keeps the residual
// vector and touches a whole column per coordinate: per coordinate ONE pass over the rows -- R loses the update of the
// coordinate before (applied late: it would be a pass of its own), gains w_j X_j, and X_j'R is summed on the way -- in
// a fixed grid whose block sums are added in a fixed order (the same bits on every run), then one block finishes the
// coordinate (soft threshold, the sweep's max |w| and max |dw|).  32 bytes per row and coordinate: HBM-bound.
constexpr int CD_BLOCKS = 1024, CD_THREADS = 256, CD_MAXP = 64;

struct CdState {
    double w[CD_MAXP];
    double norm[CD_MAXP];
    double pend_w;          // R still holds + pend_w * X[pend_j]
    int pend_j;
    double w_max, d_w_max;
    double red[CD_MAXP + 3];     // finished sums: X_j'R, R'R, R'y, y'y
};

__device__ __forceinline__ double block_sum(double v, double *sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < CD_THREADS / 64; i++) t += sh[i];
    return t;                                                   // (thread 0 holds the block's sum)
}

// R = y - sum_j w_j X_j; partial sums of X_j'X_j (j < p) and y'y (slot p)
__global__ __launch_bounds__(CD_THREADS) void cd_prep_kernel(const double *__restrict__ X, const double *__restrict__ y, uint64_t N, int p,
                                                               const CdState *__restrict__ st, double *__restrict__ R, double *__restrict__ part)
{
    __shared__ double sh[CD_THREADS / 64];
    __shared__ double sw[CD_MAXP];
    if (threadIdx.x < p) sw[threadIdx.x] = st->w[threadIdx.x];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * CD_THREADS;
    double yy = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * CD_THREADS + threadIdx.x; i < N; i += stride) {
        const double yi = y[i];
        double r = yi;
        for (int j = 0; j < p; j++) { const double wj = sw[j]; if (wj != 0.0) r -= wj * X[(uint64_t)j * N + i]; }
        R[i] = r;
        yy += yi * yi;
    }
    const double t = block_sum(yy, sh);
    if (threadIdx.x == 0) part[(uint64_t)p * gridDim.x + blockIdx.x] = t;
    for (int j = 0; j < p; j++) {
        double s = 0.0;
        const double *Xj = X + (uint64_t)j * N;
        for (uint64_t i = (uint64_t)blockIdx.x * CD_THREADS + threadIdx.x; i < N; i += stride) s += Xj[i] * Xj[i];
        const double tj = block_sum(s, sh);
        if (threadIdx.x == 0) part[(uint64_t)j * gridDim.x + blockIdx.x] = tj;
    }
}
// sums of `n` partial rows -> st->red[0 .. n); what == 1: they are the column norms and y'y
__global__ __launch_bounds__(CD_THREADS) void cd_sum_kernel(const double *__restrict__ part, int n, int blocks, CdState *st, int what)
{
    __shared__ double sh[CD_THREADS / 64];
    for (int k = 0; k < n; k++) {
        double s = 0.0;
        for (int b = threadIdx.x; b < blocks; b += CD_THREADS) s += part[(uint64_t)k * blocks + b];
        const double t = block_sum(s, sh);
        if (threadIdx.x == 0) { st->red[k] = t; if (what == 1 && k < n - 1) st->norm[k] = t; }
    }
}
// coordinate j: R <- R - pend_w X_pend + w_j X_j, partial sums of X_j'R
__global__ __launch_bounds__(CD_THREADS) void cd_step_kernel(const double *__restrict__ X, uint64_t N, int j, const CdState *__restrict__ st,
                                                               double *__restrict__ R, double *__restrict__ part)
{
    __shared__ double sh[CD_THREADS / 64];
    const double wj = st->w[j], pw = st->pend_w;
    const int pj = st->pend_j;
    const double *Xj = X + (uint64_t)j * N, *Xp = X + (uint64_t)(pj < 0 ? 0 : pj) * N;
    const uint64_t stride = (uint64_t)gridDim.x * CD_THREADS;
    double s = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * CD_THREADS + threadIdx.x; i < N; i += stride) {
        const double xj = Xj[i];
        double r = R[i];
        if (pj >= 0 && pw != 0.0) r -= pw * Xp[i];
        if (wj != 0.0) r += wj * xj;
        R[i] = r;
        s += xj * r;
    }
    const double t = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ __launch_bounds__(CD_THREADS) void cd_update_kernel(const double *__restrict__ part, int blocks, int j, double l1, double l2, int positive, CdState *st)
{
    __shared__ double sh[CD_THREADS / 64];
    double s = 0.0;
    for (int b = threadIdx.x; b < blocks; b += CD_THREADS) s += part[b];
    const double tmp = block_sum(s, sh);
    if (threadIdx.x == 0) {
        const double w_old = st->w[j];
        double w_new;
        if (positive && tmp < 0) w_new = 0.0;
        else {
            const double sg = (double)((tmp > 0) - (tmp < 0));
            w_new = sg * fmax(fabs(tmp) - l1, 0.0) / (st->norm[j] + l2);
        }
        st->w[j] = w_new;
        st->pend_j = j;
        st->pend_w = w_new;
        const double d = fabs(w_new - w_old);
        if (d > st->d_w_max) st->d_w_max = d;
        if (fabs(w_new) > st->w_max) st->w_max = fabs(w_new);
    }
}
// the pending update applied; partial sums of X_j'R (j < p), R'R (p), R'y (p + 1)
__global__ __launch_bounds__(CD_THREADS) void cd_gap_kernel(const double *__restrict__ X, const double *__restrict__ y, uint64_t N, int p,
                                                              const CdState *__restrict__ st, double *__restrict__ R, double *__restrict__ part)
{
    __shared__ double sh[CD_THREADS / 64];
    const double pw = st->pend_w;
    const int pj = st->pend_j;
    const double *Xp = X + (uint64_t)(pj < 0 ? 0 : pj) * N;
    const uint64_t stride = (uint64_t)gridDim.x * CD_THREADS;
    double rr = 0.0, ry = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * CD_THREADS + threadIdx.x; i < N; i += stride) {
        double r = R[i];
        if (pj >= 0 && pw != 0.0) r -= pw * Xp[i];
        R[i] = r;
        rr += r * r;
        ry += r * y[i];
    }
    double t = block_sum(rr, sh);
    if (threadIdx.x == 0) part[(uint64_t)p * gridDim.x + blockIdx.x] = t;
    t = block_sum(ry, sh);
    if (threadIdx.x == 0) part[(uint64_t)(p + 1) * gridDim.x + blockIdx.x] = t;
    __threadfence();
    __syncthreads();
    for (int j = 0; j < p; j++) {
        double s = 0.0;
        const double *Xj = X + (uint64_t)j * N;
        for (uint64_t i = (uint64_t)blockIdx.x * CD_THREADS + threadIdx.x; i < N; i += stride) s += Xj[i] * R[i];
        const double tj = block_sum(s, sh);
        if (threadIdx.x == 0) part[(uint64_t)j * gridDim.x + blockIdx.x] = tj;
    }
}
__global__ void cd_clear_pending_kernel(CdState *st, int sweep_only)
{
    if (!sweep_only) { st->pend_j = -1; st->pend_w = 0.0; }
    st->w_max = 0.0;
    st->d_w_max = 0.0;
}

}  // namespace

extern "C" {

// Q [F][p][p] row-major, q [F][p], yy/n_train/n_test [F], alphas [n_alphas] (descending),
// test_stats [F][2^p][3] u64 or NULL, outputs: mse [n_alphas][F] (if test_stats), coefs
// [F][n_alphas][p], iters [F][n_alphas], gaps [F][n_alphas]; all HOST pointers.
int ss_enet_path_gram(const double *Q, const double *q, const double *yy, const double *n_train,
                      const double *n_test, int F, int p, const double *alphas, int n_alphas, double l1_ratio,
                      int max_iter, double tol, int positive, const uint64_t *test_stats, double *mse,
                      double *coefs, int *iters, double *gaps)
{
    if (!Q || !q || !yy || !n_train || !alphas || F < 1 || n_alphas < 1) return SS_EINVAL;
    if (p < 1 || p > MAXP) return SS_ERANGE;
    if (test_stats && (!mse || !n_test)) return SS_EINVAL;
    std::vector<FoldIn> in((size_t)F);
    for (int f = 0; f < F; f++) {
        memset(&in[f], 0, sizeof(FoldIn));
        memcpy(in[f].Q, Q + (size_t)f * p * p, sizeof(double) * p * p);
        memcpy(in[f].q, q + (size_t)f * p, sizeof(double) * p);
        in[f].yy = yy[f];
        in[f].n_train = n_train[f];
        in[f].n_test = n_test ? n_test[f] : 0.0;
    }
    const uint32_t M = 1u << p;
    FoldIn *d_in = nullptr;
    double *d_alphas = nullptr, *d_mse = nullptr, *d_coefs = nullptr, *d_gaps = nullptr;
    int *d_iters = nullptr;
    unsigned long long *d_ts = nullptr;
    int rc = SS_OK;
    const size_t na = (size_t)F * n_alphas;
    if (ss::l2s::dmalloc((void **)&d_in, sizeof(FoldIn) * F) != hipSuccess || ss::l2s::dmalloc((void **)&d_alphas, 8 * n_alphas) != hipSuccess ||
        ss::l2s::dmalloc((void **)&d_mse, 8 * na) != hipSuccess || ss::l2s::dmalloc((void **)&d_coefs, 8 * na * p) != hipSuccess ||
        ss::l2s::dmalloc((void **)&d_gaps, 8 * na) != hipSuccess || ss::l2s::dmalloc((void **)&d_iters, 4 * na) != hipSuccess ||
        (test_stats && ss::l2s::dmalloc((void **)&d_ts, (size_t)F * M * 24) != hipSuccess))
        rc = SS_ENOMEM;
    if (!rc) {
        ss::l2s::copy(d_in, in.data(), sizeof(FoldIn) * F, hipMemcpyHostToDevice);
        ss::l2s::copy(d_alphas, alphas, 8 * n_alphas, hipMemcpyHostToDevice);
        ss::l2s::set(d_gaps, 0, 8 * na);
        if (test_stats) ss::l2s::copy(d_ts, test_stats, (size_t)F * M * 24, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(enet_path_kernel, dim3(F), dim3(64), 0, ss::l2s::stream(), d_in, p, d_alphas, n_alphas, l1_ratio, max_iter,
                           tol, positive, d_ts, M, d_mse, d_coefs, d_iters, d_gaps);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = ss::l2s::sync();
        if (e == hipSuccess && test_stats) e = ss::l2s::copy(mse, d_mse, 8 * na, hipMemcpyDeviceToHost);
        if (e == hipSuccess && coefs) e = ss::l2s::copy(coefs, d_coefs, 8 * na * p, hipMemcpyDeviceToHost);
        if (e == hipSuccess && iters) e = ss::l2s::copy(iters, d_iters, 4 * na, hipMemcpyDeviceToHost);
        if (e == hipSuccess && gaps) e = ss::l2s::copy(gaps, d_gaps, 8 * na, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { ss::set_last_error("ss_enet_path_gram", __FILE__, __LINE__, e); rc = SS_EHIP; }
    }
    ss::l2s::dfree(d_in); ss::l2s::dfree(d_alphas); ss::l2s::dfree(d_mse); ss::l2s::dfree(d_coefs); ss::l2s::dfree(d_gaps); ss::l2s::dfree(d_iters); ss::l2s::dfree(d_ts);
    return rc;
}

// The residual form (_cd_fast.enet_coordinate_descent; ElasticNet(precompute=False).fit at identify_strains...:451-455).
// X: HOST, column-major [p][N] doubles (Fortran order, as scikit-learn holds it); y: host [N]; w [p]: in = the start, out =
// the coefficients; l1 = alpha * l1_ratio * N and l2 = alpha * (1 - l1_ratio) * N, as ElasticNet.fit passes them.
int ss_enet_cd(const double *X, const double *y, uint64_t N, int p, double l1, double l2, int max_iter, double tol, int positive,
               double *w, double *gap_out, int *n_iter_out)
{
    if (!X || !y || !w || N == 0 || max_iter < 1) return SS_EINVAL;
    if (p < 1 || p > CD_MAXP) return SS_ERANGE;
    double *d_X = nullptr, *d_y = nullptr, *d_R = nullptr, *d_part = nullptr;
    CdState *d_st = nullptr;
    CdState h;
    memset(&h, 0, sizeof h);
    for (int j = 0; j < p; j++) h.w[j] = w[j];
    h.pend_j = -1;
    const int blocks = (int)std::min<uint64_t>(CD_BLOCKS, (N + CD_THREADS - 1) / CD_THREADS);
    int rc = SS_OK;
    hipError_t e = hipSuccess;
#define CD(call) do { if (e == hipSuccess) e = (call); } while (0)
    if (ss::l2s::dmalloc((void **)&d_X, (uint64_t)p * N * 8) != hipSuccess || ss::l2s::dmalloc((void **)&d_y, N * 8) != hipSuccess ||
        ss::l2s::dmalloc((void **)&d_R, N * 8) != hipSuccess || ss::l2s::dmalloc((void **)&d_part, (uint64_t)(p + 2) * blocks * 8) != hipSuccess ||
        ss::l2s::dmalloc((void **)&d_st, sizeof(CdState)) != hipSuccess)
        rc = SS_ENOMEM;
    double gap = 0.0;
    int n_iter = 0;
    if (!rc) {
        CD(ss::l2s::copy(d_X, X, (uint64_t)p * N * 8, hipMemcpyHostToDevice));
        CD(ss::l2s::copy(d_y, y, N * 8, hipMemcpyHostToDevice));
        CD(ss::l2s::copy(d_st, &h, sizeof h, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cd_prep_kernel, dim3(blocks), dim3(CD_THREADS), 0, ss::l2s::stream(), d_X, d_y, N, p, d_st, d_R, d_part);
        hipLaunchKernelGGL(cd_sum_kernel, dim3(1), dim3(CD_THREADS), 0, ss::l2s::stream(), d_part, p + 1, blocks, d_st, 1);
        CD(ss::l2s::copy(&h, d_st, sizeof h, hipMemcpyDeviceToHost));
        const double yy = h.red[p];
        const double d_w_tol = tol;
        tol *= yy;
        gap = tol + 1.0;
        for (n_iter = 0; n_iter < max_iter && e == hipSuccess; n_iter++) {
            hipLaunchKernelGGL(cd_clear_pending_kernel, dim3(1), dim3(1), 0, ss::l2s::stream(), d_st, 1);
            for (int j = 0; j < p; j++) {
                if (h.norm[j] == 0.0) continue;
                hipLaunchKernelGGL(cd_step_kernel, dim3(blocks), dim3(CD_THREADS), 0, ss::l2s::stream(), d_X, N, j, d_st, d_R, d_part);
                hipLaunchKernelGGL(cd_update_kernel, dim3(1), dim3(CD_THREADS), 0, ss::l2s::stream(), d_part, blocks, j, l1, l2, positive, d_st);
            }
            CdState s2;
            CD(ss::l2s::copy(&s2, d_st, sizeof s2, hipMemcpyDeviceToHost));
            if (e != hipSuccess) break;
            if (s2.w_max == 0.0 || s2.d_w_max / s2.w_max < d_w_tol || n_iter == max_iter - 1) {
                hipLaunchKernelGGL(cd_gap_kernel, dim3(blocks), dim3(CD_THREADS), 0, ss::l2s::stream(), d_X, d_y, N, p, d_st, d_R, d_part);
                hipLaunchKernelGGL(cd_sum_kernel, dim3(1), dim3(CD_THREADS), 0, ss::l2s::stream(), d_part, p + 2, blocks, d_st, 0);
                hipLaunchKernelGGL(cd_clear_pending_kernel, dim3(1), dim3(1), 0, ss::l2s::stream(), d_st, 0);
                CD(ss::l2s::copy(&s2, d_st, sizeof s2, hipMemcpyDeviceToHost));
                if (e != hipSuccess) break;
                double dual = 0.0, w_norm2 = 0.0, l1_norm = 0.0;
                for (int j = 0; j < p; j++) {
                    const double xta = s2.red[j] - l2 * s2.w[j];
                    const double v = positive ? xta : fabs(xta);
                    if (j == 0 || v > dual) dual = v;
                    w_norm2 += s2.w[j] * s2.w[j];
                    l1_norm += fabs(s2.w[j]);
                }
                const double R_norm2 = s2.red[p], Ry = s2.red[p + 1];
                double cst = 1.0;
                if (dual > l1) {
                    cst = l1 / dual;
                    gap = 0.5 * (R_norm2 + R_norm2 * cst * cst);
                } else {
                    gap = R_norm2;
                }
                gap += l1 * l1_norm - cst * Ry + 0.5 * l2 * (1.0 + cst * cst) * w_norm2;
                if (gap < tol) { h = s2; n_iter++; break; }
            }
            h = s2;
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { ss::set_last_error("ss_enet_cd", __FILE__, __LINE__, e); rc = SS_EHIP; }
    }
#undef CD
    ss::l2s::dfree(d_X); ss::l2s::dfree(d_y); ss::l2s::dfree(d_R); ss::l2s::dfree(d_part); ss::l2s::dfree(d_st);
    if (!rc) {
        for (int j = 0; j < p; j++) w[j] = h.w[j];
        if (gap_out) *gap_out = gap;
        if (n_iter_out) *n_iter_out = n_iter;
    }
    return rc;
}

}  // extern "C"
