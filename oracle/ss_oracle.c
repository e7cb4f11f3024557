/*
 * ss_oracle.c -- CPU restatement of StrainScan's identification hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, the smoke check in
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load it.  The product
 * (strainscan_amd/) never imports, links or executes anything under oracle/.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
 * against golden vectors produced in the build container by the real reference
 * (library/jellyfish-linux 2.3.0 ELF, the reference Python under scikit-learn 0.24.2);
 * the generating script is tests/golden/make_golden.py.
 *
 * Third-party arithmetic restated here (absent as source from /root/reference):
 *   - Jellyfish 2.3.0 `count --if` + `dump -c`  (binary only; call sites
 *     library/identify.py:73-103, library/Vote_Strain_L2_Lasso_new_sp.py:354-372)
 *   - scikit-learn 0.23.1/0.24.2 linear_model/_cd_fast.pyx
 *     enet_coordinate_descent_gram / enet_coordinate_descent
 *     (call sites library/identify_strains_L2_Enet_Pscan_new_sp.py:437-455)
 *
 * Plain C11, no dependencies beyond libc/libm (+ OpenMP for the timed baseline loop).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_ENOMEM (-12)
#define ORC_EINVAL (-22)
#define ORC_EKEY (-2) /* the reference would raise KeyError (identify.py:101) */

/* ------------------------------------------------------------------------------------------
 * k-mer table: jellyfish semantics restated.
 *   - strand-specific (no -C at identify.py:82,86), case-insensitive, any non-ACGT byte resets
 *     the window; the --if FASTA is itself k-merised and only its k-mers are counted; every
 *     --if k-mer is dumped, zero counts included  [all probed with the 2.3.0 ELF].
 * Key encoding: A=0 C=1 G=2 T=3, first base most significant (2 bits per base, k <= 31).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t *keys;   /* EMPTY = ~0 */
    uint32_t *vals;   /* dense id of the distinct k-mer */
    uint64_t cap;     /* power of two */
    uint64_t n;       /* distinct k-mers */
    int k;
} orc_table;

static const uint64_t ORC_EMPTY = ~(uint64_t)0;

static inline int orc_code(unsigned char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

static inline uint64_t orc_mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

static int orc_table_init(orc_table *t, uint64_t expect, int k)
{
    uint64_t cap = 16;
    while (cap < expect * 2 + 2) cap <<= 1;
    t->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    t->vals = (uint32_t *)malloc(cap * sizeof(uint32_t));
    if (!t->keys || !t->vals) return ORC_ENOMEM;
    memset(t->keys, 0xff, cap * sizeof(uint64_t));
    t->cap = cap; t->n = 0; t->k = k;
    return ORC_OK;
}

static void orc_table_free(orc_table *t)
{
    free(t->keys); free(t->vals);
    t->keys = NULL; t->vals = NULL;
}

/* insert-or-find; returns dense id */
static uint32_t orc_table_put(orc_table *t, uint64_t key)
{
    uint64_t m = t->cap - 1, s = orc_mix(key) & m;
    for (;;) {
        if (t->keys[s] == key) return t->vals[s];
        if (t->keys[s] == ORC_EMPTY) {
            t->keys[s] = key;
            t->vals[s] = (uint32_t)t->n;
            return (uint32_t)t->n++;
        }
        s = (s + 1) & m;
    }
}

static inline int64_t orc_table_get(const orc_table *t, uint64_t key)
{
    uint64_t m = t->cap - 1, s = orc_mix(key) & m;
    for (;;) {
        uint64_t kk = t->keys[s];
        if (kk == key) return t->vals[s];
        if (kk == ORC_EMPTY) return -1;
        s = (s + 1) & m;
    }
}

/* rolling k-mer state */
typedef struct { uint64_t key; int run; } orc_roll;

/* ------------------------------------------------------------------------------------------
 * FASTA/FASTQ text walker (what `jellyfish count` does with its input files) [probed]:
 *   '>' record : header line, then sequence lines until the next line starting with '>';
 *   '@' record : header line, sequence lines until a line starting with '+', then quality
 *                lines until as many quality characters as sequence characters were read;
 *   sequence lines of one record are concatenated (a k-mer may span a line break); blank
 *   lines between records are skipped; '\r' is an ordinary non-ACGT byte.
 * cb(ctx, ptr, len, end_of_record) receives each sequence line.
 * ------------------------------------------------------------------------------------------ */
typedef void (*orc_seq_cb)(void *ctx, const unsigned char *p, size_t len, int end_of_record);

static void orc_walk_fastx(const unsigned char *t, size_t n, orc_seq_cb cb, void *ctx)
{
    size_t i = 0;
    while (i < n) {
        if (t[i] == '\n') { i++; continue; }
        if (t[i] == '>') {
            while (i < n && t[i] != '\n') i++;
            if (i < n) i++;
            while (i < n && t[i] != '>') {
                size_t s = i;
                while (i < n && t[i] != '\n') i++;
                cb(ctx, t + s, i - s, 0);
                if (i < n) i++;
            }
            cb(ctx, t, 0, 1);
        } else if (t[i] == '@') {
            size_t seqlen = 0, qlen = 0;
            while (i < n && t[i] != '\n') i++;
            if (i < n) i++;
            while (i < n && t[i] != '+') {
                size_t s = i;
                while (i < n && t[i] != '\n') i++;
                cb(ctx, t + s, i - s, 0);
                seqlen += i - s;
                if (i < n) i++;
            }
            cb(ctx, t, 0, 1);
            while (i < n && t[i] != '\n') i++; /* '+' line */
            if (i < n) i++;
            while (i < n && qlen < seqlen) {
                size_t s = i;
                while (i < n && t[i] != '\n') i++;
                qlen += i - s;
                if (i < n) i++;
            }
        } else {
            /* stray line: skip it */
            while (i < n && t[i] != '\n') i++;
        }
    }
}

typedef struct {
    orc_table *tab;
    orc_roll r;
    uint64_t mask;
    uint32_t *counts; /* per dense id, or NULL when building */
    int building;
    uint64_t n_records;
} orc_walk_ctx;

static void orc_cb_kmers(void *vctx, const unsigned char *p, size_t len, int eor)
{
    orc_walk_ctx *c = (orc_walk_ctx *)vctx;
    if (eor) { c->r.run = 0; c->r.key = 0; c->n_records++; return; }
    const int k = c->tab->k;
    for (size_t i = 0; i < len; i++) {
        int code = orc_code(p[i]);
        if (code < 0) { c->r.run = 0; c->r.key = 0; continue; }
        c->r.key = ((c->r.key << 2) | (uint64_t)code) & c->mask;
        if (++c->r.run >= k) {
            if (c->building) {
                orc_table_put(c->tab, c->r.key);
            } else {
                int64_t id = orc_table_get(c->tab, c->r.key);
                if (id >= 0) c->counts[id]++;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * orc_jellyfish_count  ==  identify.jellyfish_count (identify.py:73-103) as a whole:
 *   the `jellyfish count --if kmer.fa` + `dump -c` pair, then the Python tail that maps each
 *   dumped k-mer string back to a row of kmer.fa.
 *
 * kmer_fa / kmer_fa_len : text of <db>/kmer.fa  (">id\nKMER\n" per row; row = record index,
 *                         i.e. line 2*i+1, identify.py:93-94)
 * reads[], reads_len[]  : text of the FASTA/FASTQ inputs (jellyfish concatenates files)
 * upper                 : 1 = identify.py (keys are .upper()'d, :94); 0 = identify_low_mem.py:81 /
 *                         identify_low_depth.py:64 (raw keys)
 * counts[n_rows], valid[n_rows] : match_results[row] and (row in match_results)
 * Returns ORC_OK, or ORC_EKEY when the reference would hit KeyError at identify.py:101.
 * ------------------------------------------------------------------------------------------ */
int orc_jellyfish_count(const char *kmer_fa, uint64_t kmer_fa_len, int k,
                        const char *const *reads, const uint64_t *reads_len, int n_files,
                        int upper, uint64_t n_rows, uint32_t *counts, uint8_t *valid)
{
    if (k < 1 || k > 31) return ORC_EINVAL;
    const unsigned char *t = (const unsigned char *)kmer_fa;
    /* --- lines of kmer.fa, exactly like f.readlines(): row i text = line 2i+1 rstrip()'d --- */
    uint64_t nl = 0;
    for (uint64_t i = 0; i < kmer_fa_len; i++) if (t[i] == '\n') nl++;
    if (kmer_fa_len && t[kmer_fa_len - 1] != '\n') nl++;
    if (n_rows != nl / 2) return ORC_EINVAL;

    orc_table tab;
    int rc = orc_table_init(&tab, n_rows + 16, k);
    if (rc) return rc;
    orc_walk_ctx ctx = { &tab, {0, 0}, (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1), NULL, 1, 0 };
    orc_walk_fastx(t, kmer_fa_len, orc_cb_kmers, &ctx); /* --if FASTA is k-merised */

    uint32_t *dcounts = (uint32_t *)calloc(tab.n + 1, sizeof(uint32_t));
    int64_t *row_of = (int64_t *)malloc((tab.n + 1) * sizeof(int64_t));
    if (!dcounts || !row_of) { orc_table_free(&tab); free(dcounts); free(row_of); return ORC_ENOMEM; }
    for (uint64_t i = 0; i < tab.n; i++) row_of[i] = -1;

    ctx.building = 0; ctx.counts = dcounts;
    for (int f = 0; f < n_files; f++) {
        ctx.r.run = 0; ctx.r.key = 0;
        orc_walk_fastx((const unsigned char *)reads[f], reads_len[f], orc_cb_kmers, &ctx);
    }

    /* --- Python tail: kmer_index_dict[line(.upper())] = i, last row wins (identify.py:93-94) --- */
    uint64_t pos = 0, line = 0;
    while (pos < kmer_fa_len) {
        uint64_t s = pos;
        while (pos < kmer_fa_len && t[pos] != '\n') pos++;
        uint64_t e = pos;
        if (pos < kmer_fa_len) pos++;
        if (line & 1) {
            uint64_t row = line >> 1;
            while (e > s && (t[e - 1] == ' ' || t[e - 1] == '\r' || t[e - 1] == '\t')) e--; /* rstrip */
            if (row < n_rows && e - s == (uint64_t)k) {
                uint64_t key = 0; int ok = 1;
                for (uint64_t j = s; j < e; j++) {
                    int code = orc_code(t[j]);
                    /* a raw (non-upper) key can only equal jellyfish's upper-case dump if it is
                       upper-case itself */
                    if (code < 0 || (!upper && t[j] >= 'a')) { ok = 0; break; }
                    key = (key << 2) | (uint64_t)code;
                }
                if (ok) {
                    int64_t id = orc_table_get(&tab, key);
                    if (id >= 0) row_of[id] = (int64_t)row;
                }
            }
        }
        line++;
    }
    memset(counts, 0, n_rows * sizeof(uint32_t));
    memset(valid, 0, n_rows);
    rc = ORC_OK;
    for (uint64_t id = 0; id < tab.n; id++) {
        if (row_of[id] < 0) { rc = ORC_EKEY; break; } /* dumped k-mer missing from the dict */
        counts[row_of[id]] = dcounts[id];
        valid[row_of[id]] = 1;
    }
    orc_table_free(&tab); free(dcounts); free(row_of);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Flat-stream counter used for the timed CPU baseline and for seeded parity inputs:
 *   db_keys[n_keys]  : distinct 2-bit k-mers (first base most significant)
 *   bases[n]         : sequence bytes, records separated by '\n' (a record never spans a '\n')
 *   counts[n_keys]   : occurrences, same semantics as above (a1 of SURVEY 8a)
 * OpenMP over chunks cut at record separators; threads <= 0 means omp default.
 * ------------------------------------------------------------------------------------------ */
int orc_count_flat(const uint64_t *db_keys, uint64_t n_keys, int k,
                   const char *bases, uint64_t n, uint32_t *counts, int threads)
{
    if (k < 1 || k > 31) return ORC_EINVAL;
    orc_table tab;
    int rc = orc_table_init(&tab, n_keys + 16, k);
    if (rc) return rc;
    for (uint64_t i = 0; i < n_keys; i++) {
        uint32_t id = orc_table_put(&tab, db_keys[i]);
        if (id != i) { orc_table_free(&tab); return ORC_EINVAL; } /* keys must be distinct */
    }
    memset(counts, 0, n_keys * sizeof(uint32_t));
    const uint64_t mask = (1ULL << (2 * k)) - 1;
    const unsigned char *b = (const unsigned char *)bases;
    int nchunks = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    nchunks = omp_get_max_threads() * 16;
#else
    (void)threads;
#endif
    if ((uint64_t)nchunks > n / 4096 + 1) nchunks = (int)(n / 4096 + 1);
#pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < nchunks; c++) {
        uint64_t lo = n / nchunks * c, hi = (c == nchunks - 1) ? n : n / nchunks * (c + 1);
        /* move both ends forward to the byte after a separator */
        if (c > 0) { while (lo < n && b[lo - 1] != '\n') lo++; }
        if (c < nchunks - 1) { while (hi < n && b[hi - 1] != '\n') hi++; }
        uint64_t key = 0; int run = 0;
        for (uint64_t i = lo; i < hi; i++) {
            int code = orc_code(b[i]);
            if (code < 0) { run = 0; key = 0; continue; }
            key = ((key << 2) | (uint64_t)code) & mask;
            if (++run >= k) {
                int64_t id = orc_table_get(&tab, key);
                if (id >= 0) {
#pragma omp atomic
                    counts[id]++;
                }
            }
        }
    }
    orc_table_free(&tab);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * match_node + del_outlier  (identify.py:106-127; identify_low_depth.py:77-101 adds min_valid)
 *   idx[n_idx]    : the node's k-mer row list (file kmers/<id>), duplicates allowed (set())
 *   counts/valid  : match_results as arrays
 *   min_valid     : 0 for identify.py; 1000 for identify_low_depth.py:90-91 (returns 0, [])
 * out: length = |valid & set(idx)|, n_kept/sum_kept = len/sum of k_profile after del_outlier,
 *      n_pos = len(k_profile) before del_outlier, median = np.median(k_profile) (NaN if empty)
 * ------------------------------------------------------------------------------------------ */
static int orc_cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}
static int orc_cmp_i64(const void *a, const void *b)
{
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

int orc_match_node(const uint32_t *counts, const uint8_t *valid, uint64_t n_rows,
                   const int64_t *idx, uint64_t n_idx, int64_t min_valid,
                   int64_t *length, int64_t *n_pos, int64_t *n_kept, int64_t *sum_kept,
                   double *median)
{
    *length = 0; *n_pos = 0; *n_kept = 0; *sum_kept = 0; *median = NAN;
    if (n_idx == 0) return ORC_OK;
    int64_t *u = (int64_t *)malloc(n_idx * sizeof(int64_t));
    uint32_t *prof = (uint32_t *)malloc(n_idx * sizeof(uint32_t));
    if (!u || !prof) { free(u); free(prof); return ORC_ENOMEM; }
    memcpy(u, idx, n_idx * sizeof(int64_t));
    qsort(u, n_idx, sizeof(int64_t), orc_cmp_i64);
    uint64_t np_ = 0; int64_t len = 0;
    for (uint64_t i = 0; i < n_idx; i++) {
        if (i && u[i] == u[i - 1]) continue;           /* set(map(int, ...)) */
        if (u[i] < 0 || (uint64_t)u[i] >= n_rows) continue;
        if (!valid[u[i]]) continue;                     /* valid_kmers & d */
        len++;
        if (counts[u[i]] > 0) prof[np_++] = counts[u[i]]; /* "don't ignore count 1" */
    }
    if (len < min_valid) { free(u); free(prof); return ORC_OK; } /* low_depth: return 0, [] */
    *length = len; *n_pos = (int64_t)np_;
    if (np_ > 0) {
        qsort(prof, np_, sizeof(uint32_t), orc_cmp_u32);
        double med = (np_ & 1) ? (double)prof[np_ / 2]
                               : ((double)prof[np_ / 2 - 1] + (double)prof[np_ / 2]) / 2.0;
        double cutoff = 100.0 * med;                    /* identify.py:107 */
        int64_t nk = 0, sk = 0;
        for (uint64_t i = 0; i < np_; i++)
            if (!((double)prof[i] >= cutoff)) { nk++; sk += prof[i]; }
        *n_kept = nk; *sum_kept = sk; *median = med;
    }
    free(u); free(prof);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * scikit-learn _cd_fast.enet_coordinate_descent_gram, positive=True, selection='cyclic'
 * (sklearn/linear_model/_cd_fast.pyx).  Q is p x p row-major, w is
 * warm-started in place.  Returns the number of sweeps; *gap_out the last dual gap.
 * ------------------------------------------------------------------------------------------ */
int orc_enet_cd_gram(double *w, double l1, double l2, const double *Q, const double *q,
                     double y_norm2, int p, int max_iter, double tol, int positive,
                     double *gap_out)
{
    double H[64], XtA[64];
    if (p > 64) return ORC_EINVAL;
    for (int i = 0; i < p; i++) {
        double s = 0.0;
        for (int j = 0; j < p; j++) s += Q[i * p + j] * w[j];
        H[i] = s;
    }
    double gap = tol + 1.0;
    const double d_w_tol = tol;
    tol = tol * y_norm2;
    int n_iter = 0;
    for (n_iter = 0; n_iter < max_iter; n_iter++) {
        double w_max = 0.0, d_w_max = 0.0;
        for (int ii = 0; ii < p; ii++) {
            if (Q[ii * p + ii] == 0.0) continue;
            double w_ii = w[ii];
            if (w_ii != 0.0)
                for (int j = 0; j < p; j++) H[j] -= w_ii * Q[ii * p + j];
            double tmp = q[ii] - H[ii];
            if (positive && tmp < 0) w[ii] = 0.0;
            else {
                double sg = (tmp > 0) - (tmp < 0);
                w[ii] = sg * fmax(fabs(tmp) - l1, 0) / (Q[ii * p + ii] + l2);
            }
            if (w[ii] != 0.0)
                for (int j = 0; j < p; j++) H[j] += w[ii] * Q[ii * p + j];
            double d_w_ii = fabs(w[ii] - w_ii);
            if (d_w_ii > d_w_max) d_w_max = d_w_ii;
            if (fabs(w[ii]) > w_max) w_max = fabs(w[ii]);
        }
        if (w_max == 0.0 || d_w_max / w_max < d_w_tol || n_iter == max_iter - 1) {
            double q_dot_w = 0.0;
            for (int i = 0; i < p; i++) q_dot_w += w[i] * q[i];
            for (int i = 0; i < p; i++) XtA[i] = q[i] - H[i] - l2 * w[i];
            double dual_norm = positive ? XtA[0] : fabs(XtA[0]);
            for (int i = 1; i < p; i++) {
                double v = positive ? XtA[i] : fabs(XtA[i]);
                if (v > dual_norm) dual_norm = v;
            }
            double tmp = 0.0;
            for (int i = 0; i < p; i++) tmp += w[i] * H[i];
            double R_norm2 = y_norm2 + tmp - 2.0 * q_dot_w;
            double w_norm2 = 0.0, l1_norm = 0.0;
            for (int i = 0; i < p; i++) { w_norm2 += w[i] * w[i]; l1_norm += fabs(w[i]); }
            double const_;
            if (dual_norm > l1) {
                const_ = l1 / dual_norm;
                double A_norm2 = R_norm2 * (const_ * const_);
                gap = 0.5 * (R_norm2 + A_norm2);
            } else {
                const_ = 1.0;
                gap = R_norm2;
            }
            gap += (l1 * l1_norm - const_ * y_norm2 + const_ * q_dot_w
                    + 0.5 * l2 * (1 + const_ * const_) * w_norm2);
            if (gap < tol) break;
        }
    }
    if (gap_out) *gap_out = gap;
    return (n_iter < max_iter) ? n_iter + 1 : max_iter;
}

/* ------------------------------------------------------------------------------------------
 * scikit-learn _cd_fast.enet_coordinate_descent (plain residual form), positive, cyclic.
 * X is n x p column-major (Fortran) double, as sklearn passes it.
 * ------------------------------------------------------------------------------------------ */
int orc_enet_cd(double *w, double l1, double l2, const double *X, const double *y,
                int64_t n, int p, int max_iter, double tol, int positive, double *gap_out)
{
    if (p > 64) return ORC_EINVAL;
    double norm_cols[64], XtA[64];
    double *R = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    if (!R) return ORC_ENOMEM;
    for (int j = 0; j < p; j++) {
        double s = 0.0;
        for (int64_t i = 0; i < n; i++) s += X[j * n + i] * X[j * n + i];
        norm_cols[j] = s;
    }
    for (int64_t i = 0; i < n; i++) R[i] = y[i];
    for (int j = 0; j < p; j++)
        if (w[j] != 0.0)
            for (int64_t i = 0; i < n; i++) R[i] -= w[j] * X[j * n + i];
    double yy = 0.0;
    for (int64_t i = 0; i < n; i++) yy += y[i] * y[i];
    double gap = tol + 1.0;
    const double d_w_tol = tol;
    tol *= yy;
    int n_iter;
    for (n_iter = 0; n_iter < max_iter; n_iter++) {
        double w_max = 0.0, d_w_max = 0.0;
        for (int ii = 0; ii < p; ii++) {
            if (norm_cols[ii] == 0.0) continue;
            const double *Xi = X + (size_t)ii * n;
            double w_ii = w[ii];
            if (w_ii != 0.0) for (int64_t i = 0; i < n; i++) R[i] += w_ii * Xi[i];
            double tmp = 0.0;
            for (int64_t i = 0; i < n; i++) tmp += Xi[i] * R[i];
            if (positive && tmp < 0) w[ii] = 0.0;
            else {
                double sg = (tmp > 0) - (tmp < 0);
                w[ii] = sg * fmax(fabs(tmp) - l1, 0) / (norm_cols[ii] + l2);
            }
            if (w[ii] != 0.0) for (int64_t i = 0; i < n; i++) R[i] -= w[ii] * Xi[i];
            double d = fabs(w[ii] - w_ii);
            if (d > d_w_max) d_w_max = d;
            if (fabs(w[ii]) > w_max) w_max = fabs(w[ii]);
        }
        if (w_max == 0.0 || d_w_max / w_max < d_w_tol || n_iter == max_iter - 1) {
            for (int j = 0; j < p; j++) {
                double s = 0.0;
                for (int64_t i = 0; i < n; i++) s += X[(size_t)j * n + i] * R[i];
                XtA[j] = s - l2 * w[j];
            }
            double dual_norm = positive ? XtA[0] : fabs(XtA[0]);
            for (int j = 1; j < p; j++) {
                double v = positive ? XtA[j] : fabs(XtA[j]);
                if (v > dual_norm) dual_norm = v;
            }
            double R_norm2 = 0.0, Ry = 0.0, w_norm2 = 0.0, l1_norm = 0.0;
            for (int64_t i = 0; i < n; i++) { R_norm2 += R[i] * R[i]; Ry += R[i] * y[i]; }
            for (int j = 0; j < p; j++) { w_norm2 += w[j] * w[j]; l1_norm += fabs(w[j]); }
            double const_;
            if (dual_norm > l1) {
                const_ = l1 / dual_norm;
                gap = 0.5 * (R_norm2 + R_norm2 * (const_ * const_));
            } else {
                const_ = 1.0;
                gap = R_norm2;
            }
            gap += (l1 * l1_norm - const_ * Ry + 0.5 * l2 * (1 + const_ * const_) * w_norm2);
            if (gap < tol) break;
        }
    }
    free(R);
    if (gap_out) *gap_out = gap;
    return (n_iter < max_iter) ? n_iter + 1 : max_iter;
}

/* ------------------------------------------------------------------------------------------
 * seqpy.revcomp (library/seqpy.c:5-36): reverse + complement through the IUPAC table.
 * ------------------------------------------------------------------------------------------ */
static unsigned char orc_comp(unsigned char c)
{
    /* library/seqpy.c:5-22 as a rule: letters map through the IUPAC complement, case kept;
       '`' and '@' both map to '@' (table rows 0x40 and 0x60 start with 64). */
    static const char up[] = "TVGHEFCDIJMLKNOPQYSAABWXRZ"; /* complement of 'A'..'Z' */
    if (c >= 'A' && c <= 'Z') return (unsigned char)up[c - 'A'];
    if (c >= 'a' && c <= 'z') return (unsigned char)(up[c - 'a'] + 32);
    if (c == 0x60) return 0x40;
    return c;
}

void orc_revcomp(const char *in, char *out, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++) out[n - 1 - i] = (char)orc_comp((unsigned char)in[i]);
}

int orc_omp_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
