"""Edge bytes pinned to what the reference's own tests state.

1. tests/inferedges/Makefile (k = 5, the "CAAGG" case): its five `build` inputs and the diagram in
   that Makefile (":CCAAG* -> CAAGG", "CAAGG -> *AAGGT": the starred k-mers are the only ones with
   an edge) fix every edge byte of the five graphs; the bit positions are db_node.h:180
   (nuc_orient_to_edge = 1 << (nuc + 4 * orient)) and db_graph.c:152-166 (an edge A -> B sets B's
   last base on A in A's orientation and the complement of A's first base on B in B's reversed
   orientation).  Expectations below are written out by hand, not computed.
2. src/tests/node_tests.c:10-51 (test_db_graph_next_nodes): a 2-colour k = 11 graph from a random and
   a shared 59-mer per colour; for every node, colour and orientation, every edge bit leads to a
   k-mer that is in the graph (db_graph_next_nodes asserts the lookup succeeds, db_graph.c:231-258).
   Checked here on the exported records, plus the reciprocal bit on the neighbour.

3. Two-word keys (k = 33 and k = 63), record BYTES written out by hand from binary_kmer.c:156-186
   (b[0] is the top word and holds k & 31 bases in its low bits; b[1] the last 32 bases),
   graph_writer.c:116-127 (key words in index order, each little-endian; u32 coverage; u8 edges) and
   the sort order of `build --sort` (hash_table.c:371: k-mers compared word 0 first).  Nothing in
   these expectations comes from the oracle or from SURVEY.md.

All run against the oracle (CPU) and, marked gpu, against the HIP path.
"""
import numpy as np
import pytest

NUC = {"A": 0, "C": 1, "G": 2, "T": 3}
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def kmer_int(s):
    v = 0
    for c in s:
        v = v << 2 | NUC[c]
    return v


def kmer_str(v, k):
    return "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def records(body, k, ncols):
    """.ctx body -> {kmer string: (covgs, edges)}"""
    W = (2 * k + 63) // 64
    rs = 8 * W + 5 * ncols
    a = np.frombuffer(body, np.uint8).reshape(-1, rs)
    out = {}
    for row in a:
        words = np.frombuffer(row[:8 * W].tobytes(), "<u8")
        v = 0
        for w in words:
            v = v << 64 | int(w)
        covg = tuple(int(x) for x in np.frombuffer(row[8 * W:8 * W + 4 * ncols].tobytes(), "<u4"))
        edges = tuple(int(x) for x in row[8 * W + 4 * ncols:])
        out[kmer_str(v, k)] = (covg, edges)
    return out


# --- 1. tests/inferedges/Makefile, k = 5 ---------------------------------------------------------
# canonical k-mer (smaller of the k-mer and its reverse complement), coverage, edge byte
INFEREDGES_K5 = {
    # CAAGG.left.kmers.k5.ctx: single-k-mer reads, no edges at all
    "left.kmers": (["CAAGG", "ACAAG", "CCAAG", "GCAAG", "TCAAG"],
                   {"CAAGG": (1, 0x00), "ACAAG": (1, 0x00), "CCAAG": (1, 0x00), "CTTGC": (1, 0x00), "CTTGA": (1, 0x00)}),
    # CAAGG.right.kmers.k5.ctx
    "right.kmers": (["CAAGG", "AAGGA", "AAGGC", "AAGGG", "AAGGT"],
                    {"CAAGG": (1, 0x00), "AAGGA": (1, 0x00), "AAGGC": (1, 0x00), "AAGGG": (1, 0x00), "AAGGT": (1, 0x00)}),
    # CAAGG.left.edges.k5.ctx: CCAAG -> CAAGG.  CCAAG is its own key (orientation 0): next base G = bit 2.
    # CAAGG is its own key: previous base C, stored as its complement G in the reverse nibble = bit 4 + 2.
    "left.edges": (["ACAAG", "CCAAGG", "GCAAG", "TCAAG"],
                   {"ACAAG": (1, 0x00), "CCAAG": (1, 0x04), "CAAGG": (1, 0x40), "CTTGC": (1, 0x00), "CTTGA": (1, 0x00)}),
    # CAAGG.right.edges.k5.ctx: CAAGG -> AAGGT.  CAAGG: next base T = bit 3.  AAGGT (own key): previous
    # base C -> complement G in the reverse nibble = bit 6.
    "right.edges": (["AAGGA", "AAGGC", "AAGGG", "CAAGGT"],
                    {"AAGGA": (1, 0x00), "AAGGC": (1, 0x00), "AAGGG": (1, 0x00), "CAAGG": (1, 0x08), "AAGGT": (1, 0x40)}),
    # empty.k5.ctx: one empty line
    "empty": ([""], {}),
}


def _check_inferedges(build):
    for name, (reads, want) in INFEREDGES_K5.items():
        got = records(build(5, 1, [(0, reads)]), 5, 1)
        assert got == {km: ((c,), (e,)) for km, (c, e) in want.items()}, name
    # the join of the Makefile's colour 0 (left.edges + right.edges in one colour): CAAGG carries both bits
    got = records(build(5, 1, [(0, INFEREDGES_K5["left.edges"][0]), (0, INFEREDGES_K5["right.edges"][0])]), 5, 1)
    assert got["CAAGG"] == ((2,), (0x48,)) and got["CCAAG"] == ((1,), (0x04,)) and got["AAGGT"] == ((1,), (0x40,))
    # the same two read sets in two colours keep their bits apart (per-colour edges, db_node.h:273-274)
    got = records(build(5, 2, [(0, INFEREDGES_K5["left.edges"][0]), (1, INFEREDGES_K5["right.edges"][0])]), 5, 2)
    assert got["CAAGG"] == ((1, 1), (0x40, 0x08))
    assert got["CCAAG"] == ((1, 0), (0x04, 0x00)) and got["AAGGT"] == ((0, 1), (0x00, 0x40))


# --- 2. src/tests/node_tests.c:10-51 ---------------------------------------------------------------
SHARED = "CTTTCTTATCTGGAACCAGCTTTGCGGGGATGGAGTGTAACCTTGACAATGGGTCCTGC"


def _check_next_nodes(build):
    k, ncols = 11, 2
    rng = np.random.default_rng(11)
    jobs = []
    for col in range(ncols):
        jobs.append((col, ["".join("ACGT"[i] for i in rng.integers(0, 4, 59))]))
        jobs.append((col, [SHARED]))
    recs = records(build(k, ncols, jobs), k, ncols)
    assert len(recs) >= 2 * 49 - 49
    n_edges = 0
    for km, (covg, edges) in recs.items():
        assert km <= revcomp(km)
        for col in range(ncols):
            if edges[col]:
                assert covg[col] > 0
            for orient in (0, 1):
                nib = (edges[col] >> (4 * orient)) & 0xF
                for nuc in range(4):
                    if not nib >> nuc & 1:
                        continue
                    n_edges += 1
                    # db_graph_next_nodes: forward = shift left, append nuc; reverse = shift right, prepend complement(nuc)
                    nxt = km[1:] + "ACGT"[nuc] if orient == 0 else COMP["ACGT"[nuc]] + km[:-1]
                    key = min(nxt, revcomp(nxt))
                    assert key in recs, (km, col, orient, nuc)
                    # reciprocal edge on the neighbour, same colour (db_graph_add_edge_mt sets both ends)
                    n_or = (0 if key == nxt else 1) ^ orient  # orientation in which the walk traverses the neighbour
                    back_or = n_or ^ 1
                    lost = km[0] if orient == 0 else km[-1]   # the base that leaves the window
                    back_nuc = NUC[COMP[lost]] if orient == 0 else NUC[lost]
                    assert recs[key][1][col] >> (back_nuc + 4 * back_or) & 1, (km, key, col, orient, nuc)
    assert n_edges >= 4 * 48  # two 59-mers per colour: at least 48 adjacent pairs each, both ends


# --- 3. two-word keys: record bytes by hand ----------------------------------------------------------
def _le64(v):
    return bytes((v >> (8 * i)) & 0xFF for i in range(8))


def _check_two_word_records(build):
    """Read S = C A..A G T (k + 1 bases) holds two k-mers:
         K1 = C A^(k-2) G     rc(K1) = C T^(k-2) G : equal first base, then A < T   -> key K1, forward
         K2 = A^(k-2) G T     rc(K2) = A C T^(k-2) : equal first base, then A < C   -> key K2, forward
       Edge K1 -> K2 (db_graph.c:152-166): on K1 (forward) the next base T = bit 3 -> 0x08; on K2 the
       complement of K1's first base C, i.e. G = 2, in the reverse nibble -> bit 6 -> 0x40.
       k = 33: b[0] holds 1 base, b[1] 32: K1 = {C = 1, A^31 G = 2}; K2 = {A = 0, A^30 G T = 0b1011}.
       k = 63: b[0] holds 31 bases: K1 = {C A^30 = 1 << 60, A^31 G = 2}; K2 = {0, 0b1011}.
       K2 < K1 (word 0 decides), so the sorted body is K2's record, then K1's.
       The reverse complement of S holds rc(K2) -> rc(K1): same keys, both in reverse orientation, and
       the same two edge bits (on K2, reverse: next base G -> bit 2 + 4; on K1, forward nibble: the
       complement of rc(K2)'s first base A, T -> bit 3): coverage 2, edges unchanged."""
    for k, k1w0 in ((33, 1), (63, 1 << 60)):
        S = "C" + "A" * (k - 2) + "GT"
        assert len(S) == k + 1
        one = (_le64(0) + _le64(0xB) + bytes([1, 0, 0, 0]) + bytes([0x40]) +
               _le64(k1w0) + _le64(2) + bytes([1, 0, 0, 0]) + bytes([0x08]))
        assert len(one) == 2 * 21
        assert build(k, 1, [(0, [S])]) == one, k
        two = (_le64(0) + _le64(0xB) + bytes([2, 0, 0, 0]) + bytes([0x40]) +
               _le64(k1w0) + _le64(2) + bytes([2, 0, 0, 0]) + bytes([0x08]))
        assert build(k, 1, [(0, [S, revcomp(S)])]) == two, k
        # two colours: W words, then covg[0], covg[1] (u32 each), then edges[0], edges[1]
        got = build(k, 2, [(1, [S])])
        assert got == (_le64(0) + _le64(0xB) + bytes(4) + bytes([1, 0, 0, 0]) + bytes([0, 0x40]) +
                       _le64(k1w0) + _le64(2) + bytes(4) + bytes([1, 0, 0, 0]) + bytes([0, 0x08])), k


def _oracle_build(orc):
    def build(k, ncols, jobs):
        g = orc.Graph(k, ncols, 1 << 12)
        for col, reads in jobs:
            b, o = orc.pack_reads(reads)
            g.add_reads(col, b, o)
        return g.ctx_bytes(True)[g.header_size():]
    return build


def _gpu_build(mcx, orc, defer):
    def build(k, ncols, jobs):
        g = mcx.Graph(k, ncols, 1 << 12)
        g.configure("defer", defer)
        for col, reads in jobs:
            b, o = orc.pack_reads(reads)
            g.add_reads(col, b, o)
        body = g.export(True)
        g.close()
        return body
    return build


def test_inferedges_k5_edges_oracle(orc):
    _check_inferedges(_oracle_build(orc))


def test_node_tests_next_nodes_oracle(orc):
    _check_next_nodes(_oracle_build(orc))


def test_two_word_record_bytes_oracle(orc):
    _check_two_word_records(_oracle_build(orc))


@pytest.mark.gpu
@pytest.mark.parametrize("defer", [0, 1])
def test_two_word_record_bytes_gpu(mcx, orc, defer):
    _check_two_word_records(_gpu_build(mcx, orc, defer))


@pytest.mark.gpu
@pytest.mark.parametrize("defer", [0, 1])
def test_inferedges_k5_edges_gpu(mcx, orc, defer):
    _check_inferedges(_gpu_build(mcx, orc, defer))


@pytest.mark.gpu
@pytest.mark.parametrize("defer", [0, 1])
def test_node_tests_next_nodes_gpu(mcx, orc, defer):
    _check_next_nodes(_gpu_build(mcx, orc, defer))
