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
    if (hipMalloc((void **)&d_in, sizeof(FoldIn) * F) != hipSuccess || hipMalloc((void **)&d_alphas, 8 * n_alphas) != hipSuccess ||
        hipMalloc((void **)&d_mse, 8 * na) != hipSuccess || hipMalloc((void **)&d_coefs, 8 * na * p) != hipSuccess ||
        hipMalloc((void **)&d_gaps, 8 * na) != hipSuccess || hipMalloc((void **)&d_iters, 4 * na) != hipSuccess ||
        (test_stats && hipMalloc((void **)&d_ts, (size_t)F * M * 24) != hipSuccess))
        rc = SS_ENOMEM;
    if (!rc) {
        hipMemcpy(d_in, in.data(), sizeof(FoldIn) * F, hipMemcpyHostToDevice);
        hipMemcpy(d_alphas, alphas, 8 * n_alphas, hipMemcpyHostToDevice);
        hipMemset(d_gaps, 0, 8 * na);
        if (test_stats) hipMemcpy(d_ts, test_stats, (size_t)F * M * 24, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(enet_path_kernel, dim3(F), dim3(64), 0, 0, d_in, p, d_alphas, n_alphas, l1_ratio, max_iter,
                           tol, positive, d_ts, M, d_mse, d_coefs, d_iters, d_gaps);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess && test_stats) e = hipMemcpy(mse, d_mse, 8 * na, hipMemcpyDeviceToHost);
        if (e == hipSuccess && coefs) e = hipMemcpy(coefs, d_coefs, 8 * na * p, hipMemcpyDeviceToHost);
        if (e == hipSuccess && iters) e = hipMemcpy(iters, d_iters, 4 * na, hipMemcpyDeviceToHost);
        if (e == hipSuccess && gaps) e = hipMemcpy(gaps, d_gaps, 8 * na, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { ss::set_last_error("ss_enet_path_gram", __FILE__, __LINE__, e); rc = SS_EHIP; }
    }
    hipFree(d_in); hipFree(d_alphas); hipFree(d_mse); hipFree(d_coefs); hipFree(d_gaps); hipFree(d_iters); hipFree(d_ts);
    return rc;
}

}  // extern "C"
