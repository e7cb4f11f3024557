"""strainscan_amd -- MI355X-native identification path of StrainScan.

The compute lives in hand-written gfx950 HIP kernels behind a C ABI
(include/strainscan_hip.h, strainscan_amd/lib/libstrainscan_hip.so); this package is the
host-side mirror of the reference's Python interface for that path:

    strainscan_amd.identify.identify_cluster                  (library/identify.py:402)
    strainscan_amd.identify_low_mem.identify_cluster          (library/identify_low_mem.py)
    strainscan_amd.identify_low_depth.identify_ranks          (library/identify_low_depth.py:104)
    strainscan_amd.Vote_Strain_L2_Lasso_new_sp.vote_strain_L2_batch   (library/Vote_...:247)
    strainscan_amd.identify_strains_L2_Enet_Pscan_new_sp.detect_strains  (library/identify_strains...:177)
    strainscan_amd.seqpy.revcomp                              (library/seqpy.c:24)
    strainscan_amd.StrainScan.main                            (StrainScan.py:113, the `strainscan` CLI)

There is no CPU fallback: every entry point raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"
