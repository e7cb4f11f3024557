"""Layer-2 driver -- drop-in for library/Vote_Strain_L2_Lasso_new_sp.py.

vote_strain_L2_batch(input_fq, fq2, db_dir, out_dir, ksize, res, l2, msn, pmode, emode) keeps
the reference's signature (:247) and writes the same files: <out_dir>/C<id>/StrainVote.report
(12 columns, :423-437) and <out_dir>/final_report.txt (:116-170, :232-244).  Per identified
multi-strain cluster the reference spawns `jellyfish count -m <k> --if C<id>/all_kmer.fasta`
over the whole read set (:354-372); here that is one device scan against the cluster's k-mer
table, followed by strainscan_amd.identify_strains_L2_Enet_Pscan_new_sp.detect_strains.
"""
import os
import shutil
from collections import defaultdict

import numpy as np

from . import _lib
from . import identify_strains_L2_Enet_Pscan_new_sp

STRAINVOTE_HEADER = ("Strain_ID\tStrain_Name\tCluster_ID\tRelative_Abundance_Inside_Cluster\tPredicted_Depth (Enet)\t"
                     "Predicted_Depth (Ab*cls_depth)\tCoverage\tCoverd/Total_kmr\tValid_kmr\tRemain_Coverage\tCV\t"
                     "Exist_Evidence\n")
FINAL_HEADER = ("ID\tStrain_Name\tCluster_ID\tRelative_Abundance\tPredicted_Depth (Enet)\t"
                "Predicted_Depth (Ab*cls_depth)\tCoverage\tCoverd/Total_kmr\n")
SINGLE_HEADER = ("Strain_ID\tStrain_Name\tCluster_ID\tRelative_Abundance_Inside_Cluster\tPredicted_Depth\tCoverage\t"
                 "Covered/Total_kmr\n")


def build_dir(idir):
    os.makedirs(idir, exist_ok=True)


def check_L1_res(res):
    """:68-74 -- 1 when every identified cluster is a single-strain cluster."""
    print("- Check L1 identification result firstly ...")
    check = 1
    for r in res:
        if res[r]["strain"] == 0:
            check = 0
    return check


def generate_single_report(in_dict, out_dir):
    """:232-244."""
    res_tem, sc = {}, {}
    for c in in_dict:
        res_tem[in_dict[c]["strain"]] = in_dict[c]["cls_per"]
        sc[in_dict[c]["strain"]] = c
    ranked = sorted(res_tem.items(), key=lambda d: d[1], reverse=True)
    with open(out_dir + "/final_report.txt", "w+") as o:
        o.write(SINGLE_HEADER)
        for c, r in enumerate(ranked, 1):
            e = in_dict[sc[r[0]]]
            o.write(str(c) + "\t" + r[0] + "\t" + "C" + str(sc[r[0]]) + "\t" + str(e["cls_per"]) + "\t" +
                    str(e["cls_ab"]) + "\t" + str(e["cls_cov"]) + "\t" + str(e["cls_covered_num"]) + "/" +
                    str(e["cls_total_num"]) + "\n")


def merge_res(out_dir, res):
    """:116-170 -- merge the per-cluster reports and single-strain clusters into final_report.txt."""
    dab = {}
    total_depth = 0
    dinfo = defaultdict(lambda: {})
    for r in res:
        if not res[r]["strain"] == 0:
            s = res[r]["strain"]
            total_depth += float(res[r]["s_ab"])
            dinfo[s]["cid"] = "C" + str(r)
            dinfo[s]["pde"] = "NA"
            dinfo[s]["pda"] = float(res[r]["s_ab"])
            dinfo[s]["cov"] = float(res[r]["cls_cov"])
            dinfo[s]["ct"] = str(res[r]["cls_covered_num"]) + "/" + str(res[r]["cls_total_num"])
            dinfo[s]["percent"] = float(res[r]["cls_per"])
        else:
            path = out_dir + "/C" + str(r) + "/StrainVote.report"
            if not os.path.exists(path):
                continue
            total_pda = 0
            total_pde = 0
            tem = []
            with open(path, "r") as f:
                f.readline()
                while True:
                    line = f.readline().strip()
                    if not line:
                        break
                    ele = line.split("\t")
                    total_pda += float(ele[5])
                    total_pde += float(ele[4])
                    dinfo[ele[1]]["cid"] = ele[2]
                    dinfo[ele[1]]["pde"] = str(ele[4])
                    dinfo[ele[1]]["pda"] = float(ele[5])
                    dinfo[ele[1]]["cov"] = str(ele[6])
                    dinfo[ele[1]]["ct"] = str(ele[7])
                    dinfo[ele[1]]["percent"] = float(res[r]["cls_per"]) * float(ele[3])
                    tem.append(ele[1])
            if len(tem) == 1:
                total_depth += total_pde
                dinfo[tem[0]]["pda"] = float(dinfo[tem[0]]["pde"])
            else:
                total_depth += total_pda
    for s in dinfo:
        dab[s] = dinfo[s]["pda"] / total_depth
    ranked = sorted(dab.items(), key=lambda d: d[1], reverse=True)
    with open(out_dir + "/final_report.txt", "w+") as o:
        o.write(FINAL_HEADER)
        for c, r in enumerate(ranked, 1):
            d = dinfo[r[0]]
            o.write(str(c) + "\t" + r[0] + "\t" + d["cid"] + "\t" + str(r[1]) + "\t" + str(d["pde"]) + "\t" +
                    str(d["pda"]) + "\t" + str(d["cov"]) + "\t" + d["ct"] + "\n")


def remove_1(counts):
    """:312-322 on the array form: counts ordered by k-mer id, singletons zeroed."""
    py = np.asarray(counts).astype(np.int64)
    py[py == 1] = 0
    return py


def cluster_counts(input_fq, fq2, cls_db_dir, ksize):
    """The `jellyfish count -m ksize --if all_kmer.fasta` + dump + load_kmer_count + ordering by
    k-mer id of :354-389, as one device scan.  Row r of all_kmer.fasta is k-mer id r+1
    (Build_kmer_sets_unique_region_lasso_test_allinone_sp.py:397-399,409-410)."""
    from .db import fasta_index, scan_into
    db = fasta_index(os.path.join(cls_db_dir, "all_kmer.fasta"), int(ksize), 2)
    db.expect_hits()                         # every k-mer of the cluster's strains, and the sample holds the cluster
    try:
        scan_into(db, [input_fq, fq2])       # resident reads: no second parse, no second PCIe trip
        _lib.check(_lib.lib().ss_device_sync(), "ss_device_sync")
        return db.counts_rows()
    finally:
        db.close()


def cluster_counts_many(input_fq, fq2, cls_db_dirs, ksize, group=16):
    """cluster_counts for several clusters with ONE pass over the resident reads per `group` tables (the reference's loop
    :295-296 runs jellyfish over the whole FASTQ once per cluster, :354-372): ss_scan_reads_multi.  -> list of count arrays,
    in the order of cls_db_dirs.  Without a resident read set (too large for the device) the clusters are scanned one by one."""
    from .db import fasta_index, resident_reads
    from . import dist
    rs = resident_reads([input_fq, fq2]) if len(cls_db_dirs) > 1 else None
    if rs is None:
        return [cluster_counts(input_fq, fq2, d, ksize) for d in cls_db_dirs]
    out = []
    for g0 in range(0, len(cls_db_dirs), group):
        dbs = []
        try:
            for d in cls_db_dirs[g0:g0 + group]:
                dbs.append(fasta_index(os.path.join(d, "all_kmer.fasta"), int(ksize), 2).expect_hits())
            for db in dbs:
                db.reset()
            rs.scan_into_many(dbs)
            if dist.is_distributed():
                for db in dbs:                       # same order on every rank
                    dist.allreduce_table(db)
            _lib.check(_lib.lib().ss_device_sync(), "ss_device_sync")
            out.extend(db.counts_rows() for db in dbs)
        finally:
            for db in dbs:
                db.close()
    return out


def vote_strain_L2(item, counts=None):
    """:334-438.  item = [input_fq, cluster db dir, out dir, ksize, cls_ab, 'C<id>', cls_cov,
    all identified cluster ids, l2, msn, pmode, emode, fq2]; counts: the cluster's k-mer counts when the caller has
    scanned already (cluster_counts_many)."""
    (input_fq, db_dir, out_dir, ksize, cls_ab, cls, cls_cov, all_cls, l2, msn, pmode, emode, fq2) = item[:13]
    py_o = remove_1(cluster_counts(input_fq, fq2, db_dir, ksize) if counts is None else counts)
    npp = py_o[py_o != 0]
    npp25 = 0
    with np.errstate(invalid="ignore"):
        npp_outlier = np.median(npp) * 1000 if npp.size else float("nan")
    npp75 = npp_outlier
    res, res2, strain_cov, strain_val, final_src = identify_strains_L2_Enet_Pscan_new_sp.detect_strains(
        db_dir + "/all_strains_re.npz", py_o, db_dir + "/id2strain_re.pkl", int(ksize), npp25, npp75, npp_outlier,
        cls_cov, db_dir + "/overlap_matrix.npz", all_cls, l2, msn, pmode, emode)
    if len(res) == 0:
        return
    nr = sorted(res.items(), key=lambda d: d[1], reverse=True)
    tdep = 0
    for n in nr:
        tdep += res2[n[0]]
    with open(out_dir + "/StrainVote.report", "w+") as o:
        o.write(STRAINVOTE_HEADER)
        for c, n in enumerate(nr, 1):
            name = n[0]
            sc = strain_cov[name]
            body = ("\t" + cls + "\t" + str(n[1]) + "\t" + str(res2[name]) + "\t" + str((res2[name] / tdep) * cls_ab) +
                    "\t" + str(sc[0]) + "\t" + str(sc[1]) + "/" + str(sc[2]) + "\t" + str(strain_val[name]) + "\t" +
                    str(final_src[name]))
            if n[1] > 0.02 and sc[0] > 0.7:
                o.write(str(c) + "\t" + name + body + "\t*\n")
            elif emode == 1:
                o.write(str(c) + "\t" + name + " (With_ExtraRegion_covered)" + body + "\t\n")
            else:
                o.write(str(c) + "\t" + name + body + "\t\n")


def vote_strain_L2_batch(input_fq, fq2, db_dir, out_dir, ksize, res, l2, msn, pmode, emode):
    """:247-311."""
    check = check_L1_res(res)
    if check == 1:
        print("- Only single cluster is identified, will not go to the 2nd layer identification ...")
        generate_single_report(res, out_dir)
        raise SystemExit            # the reference calls exit() here (:257)
    if len(res) == 1:
        print("- Only 1 cluster is identified ...")
        for r in res:
            cls = "C" + str(r)
            nd = db_dir + "/Kmer_Sets_L2/Kmer_Sets/C" + str(r)
            cls_out = out_dir + "/" + cls
            build_dir(cls_out)
            item = [input_fq, nd, cls_out, ksize, res[r]["cls_ab"], cls, res[r]["cls_cov"], list(res.keys()), l2, msn,
                    pmode, emode, fq2]
            vote_strain_L2(item)
            if os.path.exists(cls_out + "/StrainVote.report"):   # `cp` at :273
                shutil.copyfile(cls_out + "/StrainVote.report", out_dir + "/final_report.txt")
    else:
        print("- " + str(len(res)) + " clusters are identified ...")
        todo = []
        for r in res:
            if not res[r]["strain"] == 0:
                continue
            cls = "C" + str(r)
            nd = db_dir + "/Kmer_Sets_L2/Kmer_Sets/C" + str(r)
            cls_out = out_dir + "/" + cls
            build_dir(cls_out)
            todo.append([input_fq, nd, cls_out, ksize, res[r]["cls_ab"], cls, res[r]["cls_cov"], list(res.keys()), l2,
                         msn, pmode, emode, fq2])
        print("- Parallel strain-level identification ...")
        # clusters are independent (the reference ran them in a Pool(5) at :298-305 before it went serial at
        # :295-296): a few host threads keep the device busy while another cluster's files are read and its
        # reports written.  Under torch.distributed the per-cluster all-reduces must be issued in the same order
        # on every rank, so the loop stays serial there.  SS_L2_THREADS=1 forces the serial loop.
        from . import dist
        nthreads = 1 if dist.is_distributed() else max(1, min(len(todo), int(os.environ.get("SS_L2_THREADS", "4"))))
        # all clusters' tables in ONE pass over the resident reads (SS_L2_ONE_PASS=0: a scan per cluster, as the
        # reference does)
        counts = [None] * len(todo)
        if len(todo) > 1 and os.environ.get("SS_L2_ONE_PASS", "1") != "0":
            try:
                counts = cluster_counts_many(input_fq, fq2, [item[1] for item in todo], ksize)
                _lib.cli_clock("cluster tables scanned (%d)" % len(todo))
            except Exception:                         # noqa: B902
                # a cluster whose k-mer set cannot be read: the reference's serial loop (:295-296) has written the reports of the
                # clusters in front of it when it dies there -- so does the loop below, cluster by cluster.  (Every exception, not
                # only OSError: under a process group the ranks behind rank 0 learn of its failure as a RuntimeError, db.rank0_first,
                # and all of them must take this turn together.)
                counts, nthreads = [None] * len(todo), 1
        if nthreads == 1:
            for item, c in zip(todo, counts):
                vote_strain_L2(item, c)
        else:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=nthreads, thread_name_prefix="ss-l2")
            futs = [pool.submit(vote_strain_L2, item, c) for item, c in zip(todo, counts)]
            failed = None
            for j, fut in enumerate(futs):
                try:
                    fut.result()
                except BaseException as e:            # noqa: B902 -- the first failure in submission order is the one the serial loop meets
                    failed = (j, e)
                    break
            pool.shutdown(wait=True, cancel_futures=True)
            if failed is not None:                    # ... and what the threads behind it wrote meanwhile, the serial loop never wrote
                for item in todo[failed[0] + 1:]:
                    if os.path.exists(item[2] + "/StrainVote.report"):
                        os.unlink(item[2] + "/StrainVote.report")
                raise failed[1]
        _lib.cli_clock("clusters solved (%d)" % len(todo))
        print("- Generate final report ...")
        merge_res(out_dir, res)
