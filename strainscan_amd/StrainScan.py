"""`strainscan` command line -- drop-in for the identification CLI StrainScan.py:113-271.

Same flags (-i -j -d -o -k -l -b -p -r -e -s), same cutoff ladder, same output files
(final_report.txt, C<id>/StrainVote.report, strain_prob.txt).  Plasmid mode (-p 1/2) rebuilds a
database with the reference's offline builder (StrainScan.py:235) and is out of scope here.
"""
import argparse
import os
import re
import sys

if __name__ == "__main__" or os.path.basename(sys.argv[0] or "") in ("strainscan", "StrainScan.py"):
    # the command itself: the library is loaded and the HIP runtime started on a worker thread (0.15-0.2 s) while this
    # thread imports numpy and the modules below and reads the arguments
    import threading

    def _early():
        from . import _lib
        # (plain-text reads in a single process: the parse threads' pinned buffers too; .gz inputs never use them, they have
        #  upload buffers of their own)
        n_gz = sum(a.endswith(".gz") for a in sys.argv[1:])
        _lib.warm_up(ingest=int(os.environ.get("WORLD_SIZE", "1")) <= 1 and not n_gz, gz=min(n_gz, 2))

    threading.Thread(target=_early, name="ss-gpu-warm-up", daemon=True).start()

from . import identify, identify_low_mem, identify_low_depth, Vote_Strain_L2_Lasso_new_sp

usage = "StrainScan - A kmer-based strain-level identification tool (MI355X-native identification path)."


def generate_prob_report(prob_dict, out_dir, db_dir):
    """StrainScan.py:98-111."""
    d = {}
    with open(db_dir + "/hclsMap_95_recls.txt", "r") as f:
        while True:
            line = f.readline().strip()
            if not line:
                break
            ele = line.split("\t")
            d[int(ele[0])] = ele[-1]
    with open(out_dir + "/strain_prob.txt", "w+") as op:
        op.write("Cluster_ID\tProbability\tNumber_of_strains\tStrains_in_the_cluster\n")
        for p in prob_dict:
            st = re.split(",", d[p[0]])
            op.write("C" + str(p[0]) + "\t" + str(p[1]) + "\t" + str(len(st)) + "\t" + d[p[0]] + "\n")


def _l1(mdb, in_fq, tdb, cutoff):
    mod = identify_low_mem if mdb == 1 else identify
    return mod.identify_cluster(in_fq, tdb, cutoff)


def identify_with_ladder(in_fq, tdb, ldep, mdb):
    """StrainScan.py:192-217 -> (cls_dict, l2)."""
    l2 = 0
    if ldep == 0:
        cls_dict = _l1(mdb, in_fq, tdb, [0.1, 0.4, 1])
        if len(cls_dict) == 0:
            cls_dict = _l1(mdb, in_fq, tdb, [0.05, 0.05, 1])
            l2 = 1
        if len(cls_dict) == 0:
            print("Warning: No clusters can be detected!")
            raise SystemExit
    elif ldep == 1:
        cls_dict = _l1(mdb, in_fq, tdb, [0.01, 0.05, 1])
        l2 = 1
    elif ldep == 2:
        cls_dict = _l1(mdb, in_fq, tdb, [0.005, 0.01, 1])
        l2 = 1
    else:
        raise ValueError("-l must be 0, 1 or 2")
    return cls_dict, l2


def _clock(what):
    """SS_CLI_TRACE=1: seconds since the interpreter started, at the CLI's milestones."""
    from ._lib import cli_clock
    cli_clock(what)


def main(argv=None):
    _clock("main() entered")
    pwd = os.getcwd()
    ap = argparse.ArgumentParser(prog="StrainScan.py", description=usage)
    ap.add_argument("-i", "--input_fastq", dest="input_fq", type=str, required=True,
                    help="The dir of input fastq data --- Required")
    ap.add_argument("-j", "--input_fastq_2", dest="input_fq2", type=str,
                    help="The dir of input fastq data (for pair-end data).")
    ap.add_argument("-d", "--database_dir", dest="db_dir", type=str, required=True,
                    help="The dir of your database --- Required")
    ap.add_argument("-o", "--output_dir", dest="out_dir", type=str,
                    help="Output dir (default: current dir/StrainScan_Result)")
    ap.add_argument("-k", "--kmer_size", dest="ksize", type=str,
                    help="The size of kmer, should be odd number. (default: k=31)")
    ap.add_argument("-l", "--low_dep", dest="ldep", type=str,
                    help='"1" for low depth (< 10x), "2" for super low depth (< 1x) (default: -l 0)')
    ap.add_argument("-b", "--strain_prob", dest="sprob", type=str,
                    help="1: also output the probability of detecting a strain (or cluster) in low-depth samples")
    ap.add_argument("-p", "--plasmid_mode", dest="pmode", type=str,
                    help="plasmid / reference-genome search modes of the reference (not supported here)")
    ap.add_argument("-r", "--ref_genome", dest="rgenome", type=str, help="reference genomes for -p")
    ap.add_argument("-e", "--extraRegion_mode", dest="emode", type=str,
                    help="1: also return strains with extra regions covered (default: -e 0)")
    ap.add_argument("-s", "--minimum_snv_num", dest="msn", type=str,
                    help="The minimum number of SNV at Layer-2 identification. (default: 40)")
    args = ap.parse_args(argv)

    fq_dir = args.input_fq
    fq2 = args.input_fq2 or ""
    db_dir = args.db_dir
    ksize = args.ksize if args.ksize else 31
    ldep = int(args.ldep) if args.ldep else 0
    sprob = int(args.sprob) if args.sprob else 0
    pmode = int(args.pmode) if args.pmode else 0
    emode = int(args.emode) if args.emode else 0
    msn = int(args.msn) if args.msn else 40
    if pmode in (1, 2):
        print("Warning: plasmid / reference-genome mode (-p) needs the reference's database builder "
              "(StrainScan.py:235) and is not part of this identification path.")
        raise SystemExit(2)
    out_dir = args.out_dir if args.out_dir else pwd + "/StrainScan_Result"
    if not re.search("/", out_dir):
        out_dir = pwd + "/" + out_dir
    os.makedirs(out_dir, exist_ok=True)     # (exist_ok: under torchrun every rank arrives here with the same -o at the same moment)

    from . import dist
    rank, world = dist.init_from_env()      # torchrun: one process per GPU, reads shard across ranks
    if rank != 0:                           # every rank computes; rank 0 owns the output directory
        import atexit
        import shutil
        import tempfile
        out_dir = tempfile.mkdtemp(prefix="strainscan_rank%d_" % rank)
        atexit.register(shutil.rmtree, out_dir, True)      # the other ranks' reports are scratch
    in_fq = (fq_dir, fq2)
    tdb = db_dir + "/Tree_database"
    if sprob == 1:
        prob = identify_low_depth.identify_ranks(in_fq, tdb)
        generate_prob_report(prob, out_dir, tdb)
    mdb = 1 if os.path.exists(db_dir + "/Memory_DB") else 0
    _clock("arguments read")
    cls_dict, l2 = identify_with_ladder(in_fq, tdb, ldep, mdb)
    _clock("clusters identified")
    print(cls_dict)
    if len(cls_dict) == 0:
        print("Warning: No clusters can be detected!")
        raise SystemExit
    Vote_Strain_L2_Lasso_new_sp.vote_strain_L2_batch(fq_dir, fq2, db_dir, out_dir, ksize, dict(cls_dict), l2, msn,
                                                     pmode, emode)


def cli():
    """Entry of the `strainscan` command (pyproject.toml, bin/strainscan, python -m): main(), then the process ENDS -- pending
    cache images are written, the streams flushed, and nothing is torn down piece by piece (unpinning the parse buffers,
    freeing GBs of device memory and joining the thread pools cost a fresh process 0.1 s of its 0.65 s; the driver reclaims
    everything at once)."""
    try:
        rc = main()
    except SystemExit as e:
        rc = e.code
    _clock("reports written")
    try:
        from . import db
        db.wait_cache_writes()
    except Exception:                       # noqa: B902
        pass
    if rc is not None and not isinstance(rc, int):      # sys.exit("message"): the message goes to stderr, the code is 1
        print(rc, file=sys.stderr)
        rc = 1
    sys.stdout.flush()
    sys.stderr.flush()
    # several ranks: the ordinary way out (process group, RCCL); SS_FAST_EXIT=0: the ordinary way out for one rank as well
    # (atexit handlers, logging shutdown, temporary files of the caller's own)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("SS_FAST_EXIT", "1") == "0":
        sys.exit(rc)
    import logging
    logging.shutdown()
    os._exit(rc or 0)


if __name__ == "__main__":
    cli()
