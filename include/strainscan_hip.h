/*
 * strainscan_hip.h -- C ABI of the MI355X (gfx950) identification hot path of StrainScan.
 *
 * This is the drop-in boundary: plain C, caller-owned buffers, integer status codes (0 = ok,
 * negative = error, see ss_strerror).  Python reaches it through ctypes
 * (strainscan_amd/_lib.py); nothing in the signatures depends on PyTorch.  Each entry point
 * names the reference interface (file:line under liaoherui/StrainScan) it replaces.
 *
 * Conventions
 *   - "dev" pointers are HIP device pointers on the current device; "stream" is a hipStream_t
 *     passed as void* (NULL = the default stream).  Functions whose name ends in _dev only
 *     enqueue work on that stream; everything else is synchronous on return.
 *   - k-mer keys are 2 bits per base, FIRST base in the LEAST significant bits, with the code
 *     (ascii >> 1) & 3, i.e. A=0 C=1 T=2 G=3 (case-insensitive); k <= 31.
 *   - a "flat base block" is the device input format of the scan: the sequence bytes of the
 *     reads, one '\n' after every record.  Any byte outside ACGTacgt ends a k-mer window, so
 *     k-mers never span records and N / IUPAC codes behave as in `jellyfish count`.
 */
#ifndef STRAINSCAN_HIP_H
#define STRAINSCAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_OK 0
#define SS_EINVAL (-22)   /* bad argument */
#define SS_ENOMEM (-12)   /* host or device allocation failed */
#define SS_EIO (-5)       /* file could not be read */
#define SS_EHIP (-1000)   /* a HIP runtime call failed; ss_last_error() has the text */
#define SS_ENODEV (-19)   /* no usable GPU */
#define SS_EKEY (-2)      /* the reference would raise KeyError here */
#define SS_ERANGE (-34)   /* value out of the supported range (k > 31, p > 16, ...) */
#define SS_EAGAIN (-11)   /* strict .gz policy (ss_gz_set_policy(1)): the device path declined an input; agree with the other
                             ranks, then load again under policy 2 */

#define SS_ROW_VALID 1u   /* row flag: ACGT-only k-mer of length k */
#define SS_ROW_LOWER 2u   /* row flag: text contains lower-case letters */
#define SS_NO_SLOT 0xFFFFFFFFu

int ss_version(void);
const char *ss_strerror(int code);
const char *ss_last_error(void);         /* thread-local text of the last SS_EHIP */
int ss_device_count(int *n);             /* does not initialise the GPU runtime beyond counting */
int ss_set_device(int dev);
int ss_device_sync(void);
int ss_stream_sync(void *stream);

/* device memory helpers for callers that do not bring their own allocator */
int ss_dev_alloc(void **dptr, uint64_t bytes);
int ss_dev_free(void *dptr);
/* Stream-ordered twins on the CALLING THREAD's stream (hipStreamPerThread), from the device's memory pool: what layer 2 uses
 * for its per-call vectors, so that clusters solved on several host threads do not meet in hipFree's device-wide
 * synchronisation.  A buffer must be freed by the thread that used it last (strainscan_amd/l2.py DevBuf falls back to
 * ss_dev_free otherwise). */
int ss_dev_alloc_async(void **dptr, uint64_t bytes);
int ss_dev_free_async(void *dptr);
int ss_memcpy_h2d(void *dst_dev, const void *src, uint64_t bytes, void *stream);
int ss_memcpy_d2h(void *dst, const void *src_dev, uint64_t bytes, void *stream);
int ss_memset_dev(void *dst_dev, int byte, uint64_t bytes, void *stream);

/* --------------------------------------------------------------------------------------------
 * seqpy.revcomp  (library/seqpy.c:24-36): reverse + IUPAC complement, case preserved.
 * ss_revcomp is the host form (what the CPython extension did); ss_revcomp_dev runs the same
 * table on the GPU for n_seq equal-length sequences stored back to back.
 * ------------------------------------------------------------------------------------------ */
int ss_revcomp(const char *in, char *out, uint64_t n);

/* CPUs this process may use: hardware threads capped by the cgroup CPU quota (SS_HOST_CPUS overrides).  The host-side
 * thread pools (parsers, inflater, index build) size themselves with it. */
int ss_host_cpus(void);

/* Whole-file gunzip as ss_scan_files / ss_reads_load do it for .gz inputs (the reference pipes `zcat`,
 * identify.py:81-84).  mode 0: one gzip member is inflated by `threads` threads (0 = all, up to 32) when it is
 * large enough -- entry points found inside the deflate stream, chunks decoded against an unknown 32 KB window
 * and resolved afterwards, result verified against the member's CRC-32 and length -- else by libdeflate;
 * 1: the threaded inflater or nothing; 2: libdeflate only.  SS_ERANGE = not inflated here (not gzip, too small,
 * several members, over the memory budget, verification failed): stream it with zlib.  *text is released with
 * ss_gz_free.  Host only. */
int ss_gz_inflate(const char *path, int threads, int mode, char **text, uint64_t *len);
void ss_gz_free(char *text);
/* The same text written to out_path (the threaded inflater writes through a shared mapping of the file): the ranks
 * of one node inflate a .gz sample ONCE into /dev/shm and each parses its share of the plain text. */
int ss_gz_inflate_to_file(const char *path, const char *out_path, int threads, uint64_t *len);
/* The same on the GPU (ss_ginflate.hip: the two-pass scheme with one wave per chunk of the deflate data, the
 * text verified against CRC-32 and ISIZE of the trailer); *text is a host buffer released with ss_gz_free.  SS_ERANGE:
 * not handled on the device (a damaged file, data that expands beyond the symbol budget, ...): use ss_gz_inflate. */
int ss_gz_inflate_gpu(const char *path, char **text, uint64_t *len);
/* Members the device inflater has produced / has declined in this process.  Unless SS_GZ_GPU=0, the .gz inputs of
 * ss_scan_files and ss_reads_load (one rank, files of 1 MB and more) go through it first and strict four-line FASTQ is
 * reduced to its sequence lines on the device too (ss_fastq_dev.hip); what it declines goes to the host inflaters. */
int ss_gz_gpu_counters(uint64_t *handled, uint64_t *declined);
/* The device inflater keeps its scratch (symbol streams, window maps, the text buffer: ~6-12 GB, at most two sets; its
 * pinned upload buffers) for the next call, the binning of resident reads its 0.2 GB -- large allocations are slow to come
 * by on this platform; this hands all of it back, and the large device blocks kept by ss_dev_big_blocks' stash. */
int ss_gz_gpu_release(void);
/* Device blocks of 256 MB and more that a load is done with -- the text of a large .gz, the file-order slab that binning has
 * replaced, the slabs of a destroyed read set -- are kept (at most three, 24 GB) and handed to the next large request of this
 * process (slabs, binned slabs, .gz texts) instead of going back to the driver, which hands fresh device memory out at
 * ~25 GB/s (ss_host.hip ss::big_take).  out[0] = blocks kept now, out[1] = their bytes, out[2] = requests served from kept blocks. */
int ss_dev_big_blocks(uint64_t out[3]);
/* A block is tagged with the device it lives on and is only ever served to a request made under that device (ss_set_device).
 * A caller that shares the GPU with other allocators (torch, another library) and wants the kept blocks back in the driver's
 * hands calls this: every kept block is hipFree'd.  (ss_gz_gpu_release does the same and more.) */
int ss_dev_big_release(void);
/* The pinned upload buffers of n_files (<= 2) concurrent .gz inputs, made ahead of time (~40 ms a set; a command-line process
 * calls this on its warm-up thread): a file of 32 MB or more then travels through them (8 ms instead of 12-30 per 66 MB);
 * without them only files of 256 MB or more make their own. */
int ss_gz_warm_up(int n_files);
/* Who inflates the .gz inputs of the next ss_reads_load / ss_scan_files* calls of this process: 0 (default) the device, and
 * the host inflaters for whatever it declines; 1 the device or NOBODY -- a declined input makes the call return SS_EAGAIN
 * with nothing loaded; 2 the host inflaters.  The two paths give a rank different shares of the reads (blocks of 4096
 * records / parse chunks), and the device path can decline for reasons of its own (free memory, a HIP error), so the
 * ranks of a sharded run must all take the same path for a file: they load under policy 1, all-reduce the outcome and,
 * if any of them was declined, all load again under policy 2 (strainscan_amd/dist.py load_agreed). */
int ss_gz_set_policy(int mode);
/* Several ranks SHARE the inflation of a gzip member (ss_ginflate.hip, range mode; the reference pipes one `zcat` per file,
 * identify.py:81-84).  The deflate data is cut into slices, slice s belongs to rank s mod world: a rank finds the block
 * starts of its slices, inflates them to symbols (the expensive part, all ranks at once) and receives what lies in front
 * of each slice -- 32 KB of text, the count of newlines so far, the bytes of the record that straddles the cut, CRC-32 and
 * length so far -- from the owner of the slice before, through `chain`: direction 0 = fill msg with what the owner of
 * slice - 1 sent, 1 = send msg to the owner of slice + 1 (blocking; return 0, anything else breaks the chain off).  The
 * rank keeps ALL records that begin in its slices (no further sharding of these files).  One member, several members
 * joined with cat, or bgzip (a slice = the members that begin in its byte range).  Active while world > 1 and chain
 * != NULL, and only under policy 1: a rank that cannot take part (a wrong entry at a slice's edge, no room, text that
 * is not four-line FASTQ, ...) still serves the chain, passes the bad news on and returns SS_EAGAIN; the ranks then settle for the
 * whole-file path (range off) or the host inflaters.  slice_bytes = 0: file size / (2 world), 4 MB .. 128 MB. */
typedef int (*ss_gz_chain_fn)(void *msg, uint64_t bytes, int slice, int direction, void *user);
int ss_gz_set_range(int rank, int world, uint64_t slice_bytes, ss_gz_chain_fn chain, void *user);
/* files this process has inflated its share of in range mode, and the slices that came to */
int ss_gz_range_counters(uint64_t *files, uint64_t *pieces);
/* Test hooks, switched by this call only (nothing in the environment does): which = 1: plant a wrong block entry in search
 * chunk `value` of every .gz inflated on the device (0 = off); 2: this process declines .gz inputs on the device (in range mode
 * it still serves the chain); 3: this rank leaves range mode WITHOUT serving the chain -- what a crashed peer looks like;
 * 4: which scan kernel of the page index a table goes through -- value 1: tables of k = 31 through the one-lane-per-position
 * kernel, 2: tables of every k through it, 3: tables of every k through the run-queue kernel (k at run time), 0: the product's choice
 * -- so that a test can hold the kernels to each other on one index. */
int ss_test_hook(int which, long long value);

/* The test sets of ShuffleSplit(n_splits, test_size, random_state=seed).split(range(n)) as scikit-learn 0.23
 * draws them for ElasticNetCV (identify_strains_L2_Enet_Pscan_new_sp.py:436-442: cv=ShuffleSplit(20, test_size=.5,
 * random_state=0)): bits[i] bit f = row i is in the test set of split f (the first n_test entries of the f-th
 * numpy.random.RandomState(seed).permutation(n)).  Host only; n < 2^32, n_splits <= 31. */
int ss_shuffle_split_bits(uint64_t n, int n_splits, uint64_t n_test, uint32_t seed, uint32_t *bits);
int ss_revcomp_dev(const char *in_dev, char *out_dev, uint64_t seq_len, uint64_t n_seq, void *stream);

/* --------------------------------------------------------------------------------------------
 * k-mer FASTA rows  (the `--if <db>/kmer.fa` / `all_kmer.fasta` argument of
 * library/identify.py:82-86 and library/Vote_Strain_L2_Lasso_new_sp.py:359-371, and the
 * kmer_index_dict loop of identify.py:90-95).
 * Row i = i-th record (line 2i+1).  A row whose text is an ACGT-only k-mer (either case) gets
 * SS_ROW_VALID and its 2-bit key; other rows get flags 0.
 * ------------------------------------------------------------------------------------------ */
int ss_kmerfa_count_rows(const char *path, uint64_t *n_rows);
/* <Tree_database>/kmers/<id> (one line of 0-based row numbers of kmer.fa each, Build_tree.py:686-698; read by match_node,
 * identify.py:116-118) for all listed ids, on the host's threads.  rows == NULL: counts[i] = row numbers of file i; else they are
 * written to rows + offsets[i] (counts from the first call are checked).  A token that is not a decimal number below n_rows:
 * SS_EINVAL, *bad = index of the first such file.  Host only. */
int ss_node_lists_parse(const char *kmers_dir, const long long *ids, uint32_t n_ids, uint64_t n_rows, uint64_t *counts,
                        const uint64_t *offsets, uint32_t *rows, uint32_t *bad);
int ss_kmerfa_encode(const char *path, int k, uint64_t n_rows, uint64_t *keys, uint8_t *flags, int threads);
/* same on an in-memory text */
int ss_kmerfa_encode_mem(const char *text, uint64_t len, int k, uint64_t n_rows, uint64_t *keys, uint8_t *flags);
/* one ASCII k-mer -> key; returns SS_EINVAL if it is not ACGT-only */
int ss_encode_kmer(const char *kmer, int k, uint64_t *key);

/* --------------------------------------------------------------------------------------------
 * Device k-mer database = what `jellyfish count --if` builds from the FASTA
 * (identify.py:82-86): an open-address table of the distinct valid row k-mers.
 *   upper_keys = 1: identify.py semantics (dict keyed by .upper(), identify.py:94)
 *   upper_keys = 0: identify_low_mem.py:81 / identify_low_depth.py:64 (raw keys): a lower-case
 *                   row never owns a k-mer; ss_db_build returns SS_EKEY when a k-mer is left
 *                   without an owning row (the reference raises KeyError at :88).
 *   upper_keys = 2: raw keys, lenient -- Vote_Strain_L2_Lasso_new_sp.py:312-322 looks each
 *                   all_kid.pkl key up in the dump and takes 0 when it is absent, so a lower-case
 *                   row simply never gets a count.
 * Duplicate k-mers: the LAST row owns the count (dict overwrite); earlier rows are not valid.
 * ------------------------------------------------------------------------------------------ */
typedef struct ss_db ss_db;
int ss_db_build(const uint64_t *keys, const uint8_t *flags, uint64_t n_rows, int k, int upper_keys,
                ss_db **out);
int ss_db_destroy(ss_db *db);
int ss_db_info(const ss_db *db, uint64_t *n_rows, uint64_t *n_distinct, uint64_t *capacity, int *k);
/* The built index as a file (device arrays dumped verbatim): a database is indexed once, later
 * runs import the image.  Only the k = 31 minimizer layout is exported (SS_ERANGE otherwise);
 * ss_db_import returns SS_EINVAL / SS_EIO for anything that is not a complete image. */
int ss_db_export(const ss_db *db, const char *path);
int ss_db_import(const char *path, ss_db **out);
/* row_valid[i] = 1 iff row i is a key of the reference's match_results (identify.py:96-101) */
int ss_db_row_valid(const ss_db *db, uint8_t *row_valid);
const uint8_t *ss_db_row_valid_dev(const ss_db *db);
uint64_t ss_db_device_bytes(const ss_db *db);
/* Shape of the built index, for benchmarks and logs (no reference counterpart: jellyfish prints nothing):
 * out[0] layout (0 flat table, 1 minimizer pages), [1] counters, [2] minimizers, [3] 64-byte pages,
 * [4] log2 of the filter size in bits (0 = none), [5] distinct k-mers, [6] bucket-array slots, [7] k-mers inline in pages. */
int ss_db_index_info(const ss_db *db, uint64_t out[8]);
/* A hint, not a semantic switch (counts are the same either way): most k-mers of the reads ARE in this table -- a layer-2
 * cluster table, all_kmer.fasta of a cluster the sample was found to contain (Vote_Strain_L2_Lasso_new_sp.py:354-372; the
 * reference runs the same jellyfish command for it as for the tree).  Scans of such a table skip the minimizer filter and,
 * for a resident read set in locality order, add the hits of neighbouring reads up in LDS before they go to the counters
 * (ss_mini.hip: a global atomic costs the same whatever it carries, 27 G line requests/s on MI355X). */
int ss_db_expect_hits(ss_db *db, int expect);
/* What the scan of a binned resident read set (ss_reads) learnt about a table nobody has flagged: the set's first 8192
 * tiles run through the plain kernel and report their found runs; at 8 per tile and above the rest of the set goes through
 * the combining kernel (hits added up in LDS).  out[0] = serial of the read set last probed (0: none yet), [1] = 1 when the
 * combining kernel was chosen, [2] = found runs per tile, times 1000.  One probe per (read set, table); no reference
 * counterpart (jellyfish has one counting loop). */
int ss_db_probe_info(const ss_db *db, uint64_t out[3]);

/* --------------------------------------------------------------------------------------------
 * The scan  (the `jellyfish count` + `dump -c` pair of identify.py:82-87,
 * identify_low_mem.py:73-75, identify_low_depth.py:53-59, Vote_...:359-372).
 * Counts accumulate in the db handle until ss_scan_reset.
 *   ss_scan_flat_dev : flat base block already in HBM (n bytes), enqueued on `stream`
 *   ss_scan_flat_host: flat base block in host memory; staged through pinned buffers, overlapping
 *                      copies with kernels; synchronous
 *   ss_scan_files    : FASTA/FASTQ files (plain or .gz), parsed on the host the way jellyfish
 *                      reads them (multi-line records, '+'/'@' quality lines), then scanned
 *                      (on streams of the parser's own: the call first waits for what is pending on
 *                      the default stream -- an ss_scan_reset(db, NULL) just before it)
 *   ss_scan_reset    : zeroes the counters, asynchronously on `stream`
 * ------------------------------------------------------------------------------------------ */
int ss_scan_reset(ss_db *db, void *stream);
int ss_scan_flat_dev(ss_db *db, const void *bases_dev, uint64_t n, void *stream);
int ss_scan_flat_host(ss_db *db, const char *bases, uint64_t n);
int ss_scan_files(ss_db *db, const char *const *paths, int n_paths, uint64_t *n_records, uint64_t *n_bases);
/* The same for one rank of a sharded scan (reads shard across GPUs): only the chunks c of the input with
 * c % shard_world == shard_rank are parsed, copied and scanned (the sequential reader, where a file needs it, walks
 * everything and keeps every shard_world-th block).  n_records / n_bases: this rank's share for chunk-parsed files. */
int ss_scan_files_shard(ss_db *db, const char *const *paths, int n_paths, int shard_rank, int shard_world,
                        uint64_t *n_records, uint64_t *n_bases);
/* counts per ROW (match_results as an array; 0 for rows that are not valid) */
int ss_counts_rows_dev(const ss_db *db, uint32_t *counts_rows_dev, void *stream);
int ss_counts_rows(const ss_db *db, uint32_t *counts_rows);
/* replace the accumulated counts by the given per-row counts (multi-GPU: every rank scans its
 * read shard, the uint32[n_rows] vectors are sum-all-reduced over RCCL, and each rank loads the
 * global vector back so that all later reductions see whole-sample counts) */
int ss_counts_load_rows_dev(ss_db *db, const uint32_t *counts_rows_dev, void *stream);
uint64_t ss_scan_kernel_launches(const ss_db *db);   /* scan kernels enqueued so far (diagnostics) */

/* --------------------------------------------------------------------------------------------
 * Resident read sets.  The reference re-reads the FASTQ for the tree scan, for every identified
 * multi-strain cluster and twice more with -b (identify.py:409; Vote_Strain_L2_Lasso_new_sp.py:
 * 354-372; identify_low_depth.py:119,124).  Counting does not depend on the order of the records, so the resident
 * records CAN be kept in locality order -- sorted by the minimizer of their first k-mer, reads that start within the same
 * 17 bases of a genome become neighbours and share their page lookups: scans 20-35 % faster at high coverage, 3.5 ms
 * per 20 M reads once (ss_reorder.hip; the default since round 3, SS_READS_ORDER=file keeps the file order: it pays from
 * the second scan of a sample on, a sample that is scanned once loses ~1.4 ms per 20 M reads).  ss_reads_load parses the files ONCE (worker threads
 * for plain files) and keeps the flat base blocks in HBM; ss_scan_reads scans them against any
 * database image.  shard_rank / shard_world: keep every shard_world-th block (multi-GPU).
 * ------------------------------------------------------------------------------------------ */
typedef struct ss_reads ss_reads;
int ss_reads_load(const char *const *paths, int n_paths, int shard_rank, int shard_world, ss_reads **out);
/* The one-time costs of the first ss_reads_load of a process (pinned parse buffers, streams: ~0.1 s), paid ahead of time --
 * a command-line process calls it on a worker thread while the interpreter starts up. */
int ss_ingest_warm_up(void);
/* FASTQ parse threads a load of this process uses: min(20, CPUs this process may use / LOCAL_WORLD_SIZE) -- the cgroup quota
 * counts, and the ranks of one node (one process per GPU under torchrun) share it; SS_INGEST_THREADS overrides. */
int ss_ingest_threads(int *n);
/* A resident read set from a flat base block that is already on the device (copied; order != 0: its records are put
 * in locality order, see below). */
int ss_reads_from_flat_dev(const void *flat_dev, uint64_t n, int order, ss_reads **out);
/* Where the last binning of a read set spent its time, in ms: out[0] count pass + prefix over the bins, out[1] the driver's
 * allocation of the new slab (0.3 ms, or 60-90 ms for 3 GB on a box whose driver clears the memory first), out[2] place pass. */
int ss_reads_order_timing(double out_ms[3]);
/* Slabs this process has binned so far: out[0] through the passes for records of ONE length (every record of the slab as long
 * as its first: a sequencer's 150-base reads; the count pass checks every record and a single exception sends the slab through
 * the general passes), out[1] through the general passes (ragged records).  SS_ORDER_FIXED=0 switches the former off. */
int ss_reads_order_counters(uint64_t out[2]);
/* The resident flat blocks copied back to the host, slab after slab (host = NULL: only *len); for tests and debugging. */
int ss_reads_read_back(const ss_reads *r, char *host, uint64_t cap, uint64_t *len);
/* Lifetime: ss_scan_reads / ss_scan_reads_multi are asynchronous on the caller's stream and read the set's slabs.  Destroying
 * a set WAITS for the device (hipDeviceSynchronize) before its slabs are given up -- the large ones are kept for the next load
 * of this process (ss_dev_big_blocks), the rest go back to the driver -- so a scan still in flight on any stream finishes on
 * intact bases, as it did when the slabs were hipFree'd.  The counters of the scan belong to the ss_db, not to the read set. */
int ss_reads_destroy(ss_reads *r);
int ss_reads_info(const ss_reads *r, uint64_t *n_records, uint64_t *n_bases, uint64_t *n_blocks,
                  uint64_t *device_bytes);
/* Asynchronous on `stream`, with one exception: the FIRST scan of a (read set, table) pair whose table was not announced
 * with ss_db_expect_hits probes the set's first tiles and waits for their answer (one stream synchronisation, ~40 us of
 * kernel) before it launches the rest -- that call cannot be captured into a hipGraph; every later scan of the pair can. */
int ss_scan_reads(ss_db *db, const ss_reads *r, void *stream);
/* The same reads against several tables in ONE pass (the reference's loop over the identified clusters,
 * Vote_Strain_L2_Lasso_new_sp.py:295-296, re-reads the FASTQ for each, :354-372): equal to ss_scan_reads on each table;
 * k = 31 tables share the encoding and the minimizer runs of every tile, four tables per launch. */
int ss_scan_reads_multi(ss_db *const *dbs, int n_dbs, const ss_reads *r, void *stream);

/* --------------------------------------------------------------------------------------------
 * Host FASTA/FASTQ -> flat base block (what jellyfish's sequence parser feeds its counter).
 * out must hold at least len + 2 bytes.  Used by ss_scan_files and exposed for callers that
 * shard reads themselves (multi-GPU).
 * ------------------------------------------------------------------------------------------ */
int ss_fastx_to_flat(const char *text, uint64_t len, char *out, uint64_t *out_len, uint64_t *n_records);
typedef struct ss_reader ss_reader;
int ss_reader_open(const char *const *paths, int n_paths, ss_reader **out);
/* a record longer than the caller's buffer is cut and its last `overlap` bases are repeated at
 * the start of the next block; must be k-1 for exact counts (default 30) */
int ss_reader_set_overlap(ss_reader *r, int overlap);
/* fills `out` with whole records up to cap bytes; *out_len == 0 at end of input */
int ss_reader_next(ss_reader *r, char *out, uint64_t cap, uint64_t *out_len, uint64_t *n_records);
int ss_reader_close(ss_reader *r);

/* --------------------------------------------------------------------------------------------
 * Per-node reductions = match_node + del_outlier for every tree node at once
 * (library/identify.py:106-127; identify_low_depth.py:77-101).
 * Node lists are the files <db>/kmers/<id> (0-based rows of kmer.fa), de-duplicated by the
 * caller (the reference turns them into a set, identify.py:118).
 * For node j with rows R_j:
 *   length  = |{r in R_j : row_valid[r]}|                      (identify.py:119,127)
 *   n_pos   = |{r : valid, counts[r] > 0}|                      (:121-124)
 *   median2 = 2 * np.median(profile)  (exact integer)           (:107)
 *   n_kept / sum_kept = size / sum of {c : c < 100*median}      (:106-112)
 * ------------------------------------------------------------------------------------------ */
typedef struct ss_nodes ss_nodes;
typedef struct {
    uint32_t length;
    uint32_t n_pos;
    uint32_t n_kept;
    uint32_t reserved;
    uint64_t sum_kept;
    uint64_t median2;
} ss_node_stat;
int ss_nodes_create(const uint32_t *rows, const uint64_t *offsets, uint32_t n_nodes, ss_nodes **out);
int ss_nodes_destroy(ss_nodes *ns);
int ss_nodes_reduce_dev(const ss_nodes *ns, const uint32_t *counts_rows_dev, const uint8_t *row_valid_dev,
                        ss_node_stat *stats_dev, void *stream);
int ss_nodes_reduce(const ss_nodes *ns, const ss_db *db, ss_node_stat *stats /* host, n_nodes */);
/* Harvest path: the same statistics without touching every row of every node (ss_nodes.hip).  match_node reads
 * <db>/kmers/<id> and looks every row up in the dump (identify.py:116-124) for each visited node; a sample has
 * hits in a few dozen of an E. coli database's 1645 nodes.
 *   ss_nodes_bind               once per (node set, database): (counter, list position) pairs sorted by counter
 *   ss_nodes_harvest_dev        after a scan: one streaming pass moves the non-zero counters to their list positions
 *                               in a node-major buffer and flags their nodes
 *   ss_nodes_reduce_touched_dev statistics of all nodes (untouched ones: length, zeros), buffer and flags cleared
 * Several GPUs, between the two (dist.py): touched flags MAX-all-reduced (get/set), the touched nodes' segments packed
 * (ss_nodes_pack_dev; packed_dev = NULL only returns the size; needs a host sync for that size), summed, unpacked.
 * The product path uses the CAPPED forms, which never wait for the host: the first `cap` packed counts go into the
 * buffer, the full packed size is left in *total_dev for the caller to read when it reads the statistics (total > cap:
 * harvest again and exchange with a larger cap; dist.exchange_touched keeps cap at about twice the last total).
 * ss_nodes_clear_dev zeroes the buffer and the flags (after a failed exchange, before the next harvest). */
int ss_nodes_bind(ss_nodes *ns, const ss_db *db);
int ss_nodes_harvest_dev(ss_nodes *ns, const ss_db *db, void *stream);
int ss_nodes_reduce_touched_dev(ss_nodes *ns, ss_node_stat *stats_dev, void *stream);
int ss_nodes_touched_get_dev(const ss_nodes *ns, uint32_t *flags_dev, void *stream);
int ss_nodes_touched_set_dev(ss_nodes *ns, const uint32_t *flags_dev, void *stream);
int ss_nodes_pack_dev(ss_nodes *ns, uint32_t *packed_dev, uint64_t cap, uint64_t *n_packed, void *stream);
int ss_nodes_unpack_dev(ss_nodes *ns, const uint32_t *packed_dev, void *stream);
int ss_nodes_pack_capped_dev(ss_nodes *ns, uint32_t *packed_dev, uint64_t cap, uint64_t *total_dev, void *stream);
int ss_nodes_unpack_capped_dev(ss_nodes *ns, const uint32_t *packed_dev, uint64_t cap, void *stream);
int ss_nodes_clear_dev(ss_nodes *ns, void *stream);
/* one ad-hoc row list (adjust_profile's `remain` set, identify.py:181-189) */
int ss_rows_reduce(const ss_db *db, const uint32_t *rows, uint64_t n, ss_node_stat *stat);

/* --------------------------------------------------------------------------------------------
 * Layer 2: the k-mer x strain matrix of one cluster as bit planes
 * (library/identify_strains_L2_Enet_Pscan_new_sp.py:191-201 densifies all_strains_re.npz into
 * int8; here one bit per entry, plane-major, K bits per strain padded to 128-bit words).
 * Bit vectors passed in (A, B, nu) are device arrays of words_per_plane dwords, bit k of row k
 * at word k/32, bit k%32.
 * ------------------------------------------------------------------------------------------ */
typedef struct ss_l2 ss_l2;
/* CSR of the K x S binary matrix (scipy.sparse.load_npz of all_strains_re.npz, :200) */
int ss_l2_create(const int64_t *indptr, const int32_t *indices, uint64_t K, uint32_t S, ss_l2 **out);
/* The same with the column indices already in DEVICE memory (4-byte aligned; read, not kept).  With ss_npz_member_dev: the
 * `indices.npy` member of all_strains_re.npz inflated on the device and packed from there -- a first run against a database
 * spent 2.3 of its 3.5 s inflating these members on one host thread (np.load), 2.5 GB for a 5 M x 300 cluster. */
int ss_l2_create_dev(const int64_t *indptr, const int32_t *indices_dev, uint64_t K, uint32_t S, ss_l2 **out);
/* A member of a ZIP archive (scipy.sparse.save_npz writes one .npy array per member: Recls_withR_new.py:110-112,
 * Build_overlap_matrix_sp.py:89-98) brought to the device: the member's data at [off, off + comp_n) of `path`, content CRC-32
 * and length as the archive's directory states them (both are checked).  method 8 (deflated): inflated on the device;
 * method 0 (stored): uploaded through pinned pieces whose CRCs are taken by the reading threads.  *d_data (64-byte aligned,
 * *n == usize bytes: the .npy header, then the array) is lent until ss_npz_member_done(*lease).  SS_ERANGE: the device
 * inflater declined (no dynamic-Huffman block to enter, an extreme ratio, too small) -- read the member on the host. */
int ss_npz_member_dev(const char *path, uint64_t off, uint64_t comp_n, uint32_t crc, uint64_t usize, int method, void **d_data,
                      uint64_t *n, void **lease);
int ss_npz_member_done(void *lease);
/* zlib's CRC-32 of (a prefix whose CRC-32 is prefix_crc) followed by n copies of `byte`, in O(log n): what the `data.npy`
 * member of a binary matrix must have for its content to be nnz ones -- known without inflating it. */
int ss_crc32_repeat(uint32_t prefix_crc, int byte, uint64_t n, uint32_t *out);
/* The same matrix as ready-made bit planes: planes[s * W + k / 32] bit k % 32, W = words_per_plane =
 * ((K + 31) / 32 rounded up to a multiple of 4), bits beyond K zero.  ss_l2_export_planes writes that
 * array (S * W dwords) from a handle: the host keeps it as a per-cluster image so that later runs skip
 * decompressing and re-packing all_strains_re.npz. */
int ss_l2_create_planes(const uint32_t *planes, uint64_t K, uint32_t S, ss_l2 **out);
/* A cluster image file of this package (the cache strainscan_amd writes beside SS_IMAGE_CACHE: bit planes + the overlap matrix's
 * CSR arrays at the given 64-byte-aligned offsets) straight to the device: planes AND overlap in one upload through pinned
 * buffers; equivalent to ss_l2_create_planes + ss_l2_set_overlap on the same arrays, same checks.  SS_EINVAL: not such a file.
 * (The handle's overlap arrays live inside that one allocation: ss_l2_set_overlap on an imported handle is SS_EINVAL.) */
int ss_l2_import(const char *path, uint64_t K, uint32_t S, uint64_t off_planes, uint64_t off_ptr, uint64_t off_idx, uint64_t off_val,
                 uint64_t nnz, uint32_t n_cols, ss_l2 **out);
int ss_l2_export_planes(const ss_l2 *h, uint32_t *planes);
int ss_l2_destroy(ss_l2 *h);
int ss_l2_info(const ss_l2 *h, uint64_t *K, uint32_t *S, uint64_t *words_per_plane);
/* The O(K) vector bookkeeping of detect_strains on the device (round 4; it was a dozen single-threaded numpy passes over
 * K rows: 83 ms of a 5 M-row cluster's solve).  ss_l2_set_overlap: overlap_matrix.npz as CSR (int8 K x n_cols), once per
 * cluster.  ss_l2_prepare (:191-197, 36-38, 402-415): y_host = input_y as int64[K]; col_sel[c] = how often column c is
 * among the identified clusters' columns (`overlap.A[:, all_cls - 1]`); writes y and y_u = y * ln as uint32[K], the bit
 * vectors [y > 1], [y_u > 1], [row kept: npp25 <= y <= min(npp75, npp_out)] (W words each) and y of the kept rows (0
 * elsewhere); out = {kept rows, any y_u > 0, a count negative or beyond 32 bits}.  ss_l2_fold: the fold words of
 * ss_l2_pattern_stats from the kept-row bit vector and ShuffleSplit's test bits per kept row (host, n_keep words).
 * ss_l2_count_keep: out[0] of ss_l2_prepare from y alone, on host threads (2 ms for 5 M rows): ShuffleSplit depends on the
 * number of kept rows only and is the longest step of a large cluster's solve -- it starts before y is on the device. */
int ss_l2_count_keep(const int64_t *y_host, uint64_t K, double npp25, double npp75, double npp_out, uint64_t *n_keep);
int ss_l2_set_overlap(ss_l2 *h, const int64_t *indptr, const int32_t *indices, const int8_t *data, uint32_t n_cols);
int ss_l2_prepare(const ss_l2 *h, const int64_t *y_host, const uint8_t *col_sel, double npp25, double npp75, double npp_out,
                  uint32_t *y_dev, uint32_t *yu_dev, uint32_t *G_dev, uint32_t *Gu_dev, uint32_t *keep_dev, uint32_t *ykeep_dev,
                  uint64_t out[3]);
int ss_l2_fold(const ss_l2 *h, const uint32_t *keep_dev, const uint32_t *split_bits, uint64_t n_keep, uint32_t *fold_dev);
/* ShuffleSplit(n_splits, test_size, random_state = seed) of scikit-learn (identify_strains_L2_Enet_Pscan_new_sp.py:433-442) with the
 * swaps on the device: the host walks the one sequential MT19937 word stream of the 20 permutations on a thread of its own and
 * hands each split's partners of the rows n - 1 .. n_test to the device, which finds every row's final element without
 * replaying the swap chain (ss_host.hip).  start returns at once; wait joins, synchronises and gives uint32[n] ON THE
 * DEVICE: bit f of entry e = row e is in the TRAINING half of split f (the test half is its complement).  1 <= n_test < n.
 * ss_l2_fold_train is ss_l2_fold with these bits (no host copy of the splits at all). */
typedef struct ss_split ss_split;
int ss_split_dev_start(uint64_t n, int n_splits, uint64_t n_test, uint32_t seed, ss_split **out);
int ss_split_dev_wait(ss_split *s, const uint32_t **train_bits_dev, double *walk_ms);
int ss_split_dev_free(ss_split *s);
int ss_l2_fold_train(const ss_l2 *h, const uint32_t *keep_dev, const uint32_t *train_bits_dev, uint64_t n_keep, int n_splits, uint32_t *fold_dev);
/* out1[s] = |X_s & A|, out2[s] = |X_s & A & B|; NULL = all ones.  Serves stat_cov/cal_cov_all
 * (:33-49: A = NULL, B = [y > 1]), get_remainc (:94-108) and get_candidate_arr (:121-134):
 * A = not-yet-used k-mers, B = [y_u > 1] or [y > 1]. */
int ss_l2_popc2(const ss_l2 *h, const uint32_t *A_dev, const uint32_t *B_dev, uint64_t *out1, uint64_t *out2);
/* nu &= ~X_col : "used_kmer = used_kmer + pXt[candidate]; used_kmer[used_kmer>1]=1" (:367-369) */
int ss_l2_andnot_col(const ss_l2 *h, uint32_t col, uint32_t *nu_dev);
/* For each listed column: over rows with X = 1 and y != 0, n_nz = how many, v_lo / v_hi =
 * np.percentile(.., q_lo / q_hi, interpolation='nearest'), cnt_in / sum_in = size and sum of the
 * values inside [v_lo, v_hi].  optimize_dominat_y (:136-175, q = 5/95, uses sum_in) and
 * get_avg_depth (:110-120, q = 25/75, uses sum_in / cnt_in). */
int ss_l2_quantile_sums(const ss_l2 *h, const uint32_t *y_dev, const uint32_t *cols, uint32_t ncols, double q_lo,
                        double q_hi, uint64_t *n_nz, uint32_t *v_lo, uint32_t *v_hi, uint64_t *cnt_in,
                        uint64_t *sum_in);
/* Sufficient statistics of the elastic net over the p <= 16 selected columns: for every p-bit row
 * pattern m, {rows, sum y, sum y^2} over (f < n_folds) the kept rows in the TEST half of fold f,
 * and (f = n_folds) all kept rows.  fold_dev[k]: bit 31 = row kept by the filter of :402-415,
 * bit f = row in the test half of ShuffleSplit fold f.  stats: host, [(n_folds+1)][2^p][3]. */
int ss_l2_pattern_stats(const ss_l2 *h, const uint32_t *cols, int p, const uint32_t *y_dev,
                        const uint32_t *fold_dev, int n_folds, uint64_t *stats);

/* --------------------------------------------------------------------------------------------
 * Positive elastic-net coordinate descent along a descending alpha path, warm-started, on Gram
 * statistics -- scikit-learn's _cd_fast.enet_coordinate_descent_gram inside enet_path
 * (ElasticNetCV at identify_strains...:437-442) and, with F = 1 and one alpha, the refit of
 * ElasticNet(...).fit (:451-455).  One workgroup per problem f < F.  All pointers are host
 * memory: Q [F][p][p], q [F][p], yy / n_train / n_test [F]; test_stats [F][2^p][3] (from
 * ss_l2_pattern_stats) or NULL; out: mse [n_alphas][F] (needs test_stats), coefs
 * [F][n_alphas][p], iters / gaps [F][n_alphas] (any may be NULL).
 * ------------------------------------------------------------------------------------------ */
int ss_enet_path_gram(const double *Q, const double *q, const double *yy, const double *n_train,
                      const double *n_test, int F, int p, const double *alphas, int n_alphas, double l1_ratio,
                      int max_iter, double tol, int positive, const uint64_t *test_stats, double *mse,
                      double *coefs, int *iters, double *gaps);

/* The same coordinate descent in its RESIDUAL form -- scikit-learn's _cd_fast.enet_coordinate_descent, what
 * ElasticNet(alpha, precompute=False).fit runs at identify_strains...:451-455 -- for callers that hold the design matrix
 * itself: R = y - X w lives on the device beside X, one pass over the rows per coordinate (32 B per row), block sums added
 * in a fixed order.  X: host, column-major [p][N] doubles (Fortran order, as scikit-learn holds it), p <= 64; y: host [N];
 * w [p]: in = the start, out = the coefficients; l1 = alpha * l1_ratio * N, l2 = alpha * (1 - l1_ratio) * N as
 * ElasticNet.fit passes them; gap / n_iter as the Cython routine returns them (may be NULL).  [The product's refit uses
 * ss_enet_path_gram with F = 1: for binary columns the Gram statistics are exact and p x p.] */
int ss_enet_cd(const double *X, const double *y, uint64_t N, int p, double l1, double l2, int max_iter, double tol,
               int positive, double *w, double *gap, int *n_iter);

#ifdef __cplusplus
}
#endif
#endif /* STRAINSCAN_HIP_H */
