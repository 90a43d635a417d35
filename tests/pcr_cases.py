"""The calls of the reference's PCR-duplicate unit test (src/tests/build_graph_tests.c:19-148):
k = 19, one colour, fq_cutoff 9 (no qualities given), hp_cutoff 9.  Each step is one
build_graph_from_reads_mt call: (read 1, read 2 or None, mate orientation, remove_pcr_dups) and
the coverage the test asserts afterwards for K1 / K2 (None = not asserted)."""
K = 19
FQ_CUTOFF, HP_CUTOFF = 9, 9
K1, K2 = "CTACGATGTATGCTTAGCT", "TAGAACGTTCCCTACACGT"
STEPS = [
    ("", "", "FF", True, (None, None)),                                                   # empty reads are fine
    ("CTACGATGTATGCTTAGCTGTTCCG", "TAGAACGTTCCCTACACGTCCTATG", "FF", True, (1, 1)),      # loaded
    ("CTACGATGTATGCTTAGCTAATGAT", "TAGAACGTTCCCTACACGTTGTTTG", "FF", True, (1, 1)),      # duplicate FF
    ("CTACGATGTATGCTTAGCTCCGAAG", "AGACTAAGCTAAGCATACATCGTAG", "FR", True, (1, 1)),      # duplicate FR
    ("AGGAGTTGTCTTCTAAGGAAACGTGTAGGGAACGTTCTA", "TAGAACGTTCCCTACACGTTTTCCACGAGTTAATCTAAG", "RF", True, (1, 1)),
    ("AACCCTAAAAACGTGTAGGGAACGTTCTA", "AATGCGTGTTAGCTAAGCATACATCGTAG", "RR", True, (1, 1)),
    ("CTACGATGTATGCTTAGCTAATGAT", "TAGAACGTTCCCTACACGTTGTTTG", "FF", False, (2, 2)),     # filter off: loaded
    ("CTACGATGTATGCTTAGCTAGTGTGATATCCTCC", None, "FF", True, (2, None)),                  # SE duplicate, FF
    ("GCGTTACCTACTGACAGCTAAGCATACATCGTAG", None, "RR", True, (None, 2)),                  # SE duplicate, RR
    ("ACGTGTAGGGAACGTTCTACTTCTACCGGAGGAT", "AGCTAAGCATACATCGTAGTACAATGCACCCTCC", "FF", True, (3, 3)),  # other strand: loaded
    ("ACGTGTAGGGAACGTTCTACTTCTACCGGAGGAT", "AGCTAAGCATACATCGTAGTACAATGCACCCTCC", "FF", True, (3, 3)),  # not a second time
]
TOTAL_SEQ, CONTIGS = 168, 6   # build_graph_tests.c:50-51,93-94,125-126
