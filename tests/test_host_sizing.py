"""Row I of SURVEY.md 8(a): the product's -n / -m table sizing (mccortex_amd/host/host_util.c) against
the capacity table the survey recorded from the reference (tests/golden/reference_kats.json,
src/basic/hash_mem.c:5-51) and against the decision rules of cmd_get_kmers_in_hash as `build`
calls it (src/graph/cmd_mem.c:38-130, src/commands/ctx_build.c:317-322).  The helpers are built
into mccortex_amd/bin/libmcxhost.so by the host Makefile; no GPU needed."""
import ctypes as C
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Plan(C.Structure):
    _fields_ = [("nbuckets", C.c_uint64), ("bucket_size", C.c_uint64), ("capacity", C.c_uint64), ("bytes", C.c_size_t)]


@pytest.fixture(scope="module")
def host(mcx):
    L = C.CDLL(os.path.join(ROOT, "mccortex_amd", "bin", "libmcxhost.so"))
    L.table_plan_for_kmers.restype = Plan
    L.table_plan_for_kmers.argtypes = [C.c_uint64, C.c_size_t]
    L.table_plan_for_memory.restype = Plan
    L.table_plan_for_memory.argtypes = [C.c_size_t, C.c_size_t]
    L.table_plan_for_build.restype = C.c_char_p
    L.table_plan_for_build.argtypes = [C.c_size_t, C.c_bool, C.c_size_t, C.c_bool, C.c_size_t, C.c_int64,
                                       C.POINTER(Plan), C.c_char_p, C.c_size_t]
    L.mem_to_integer.restype = C.c_bool
    L.mem_to_integer.argtypes = [C.c_char_p, C.POINTER(C.c_size_t)]
    return L


def _build_plan(host, mem, mem_set, n, n_set, bits, max_kmers=-1):
    p = Plan()
    buf = C.create_string_buffer(256)
    err = host.table_plan_for_build(mem, mem_set, n, n_set, bits, max_kmers, C.byref(p), buf, 256)
    return p, (err.decode() if err else None)


def test_capacity_kats_of_the_reference(host, orc):
    kats = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))["hash_table_cap"]
    for nkmers, nbuckets, bucket in kats:
        p = host.table_plan_for_kmers(nkmers, 104)
        assert (p.nbuckets, p.bucket_size, p.capacity) == (nbuckets, bucket, nbuckets * bucket), nkmers
        # memory figure: entries * bits / 8 + 2 bytes per bucket (hash_mem.h: ht_mem)
        assert p.bytes == (bucket * nbuckets * 104) // 8 + nbuckets * 2
        # ... and the oracle's copy of the same function agrees
        nb, bs = C.c_uint64(), C.c_uint8()
        assert orc.lib().orc_hash_table_cap(nkmers, C.byref(nb), C.byref(bs)) == p.capacity
        assert (nb.value, bs.value) == (nbuckets, bucket)


def test_constants_and_memory_formula_of_the_reference_header(host, orc):
    """src/basic/hash_mem.h compiled unmodified into oracle/_ref: REHASH_LIMIT and MAX_BUCKET_SIZE against the oracle's,
    ht_mem() against the bytes the product's sizing reports, the occupancy thresholds `build` warns at
    (cmd_mem.c:66, ctx_build.c) against the literals the host program uses."""
    R = orc.ref_hashmem()
    if R is None:
        pytest.skip("oracle/_ref/libhashmemref.so not built (needs /root/reference at build time)")
    L = orc.lib()
    assert (R.ref_rehash_limit(), R.ref_max_bucket_size()) == (L.orc_rehash_limit(), L.orc_max_bucket_size()) == (20, 48)
    assert (R.ref_ideal_occupancy(), R.ref_warn_occupancy()) == (0.75, C.c_float(0.9).value)
    src = open(os.path.join(ROOT, "mccortex_amd", "host", "cmd_build.c")).read()
    assert "#define IDEAL_OCCUPANCY 0.75f" in src and "#define WARN_OCCUPANCY 0.9f" in src
    import random
    rnd = random.Random(3)
    for bits in (104, 168, 288):
        for _ in range(200):
            nk = rnd.randrange(1, 1 << rnd.randrange(4, 40))
            p = host.table_plan_for_kmers(nk, bits)
            assert 1 <= p.bucket_size <= R.ref_max_bucket_size() + 1   # (hash_mem.c:9-11: floor(n / buckets) <= 48, the bucket takes the ceiling)
            assert p.bytes == R.ref_ht_mem(p.bucket_size, p.nbuckets, bits)
        for mem in (1 << 20, 512 << 20, 70 << 30):
            p = host.table_plan_for_memory(mem, bits)
            assert p.bytes == R.ref_ht_mem(p.bucket_size, p.nbuckets, bits) <= mem


def test_memory_limit_fills_but_never_exceeds(host):
    for bits in (104, 64 + 40 + 64, 128 + 40 * 4):        # k=31 1 colour; --sort; k=63 4 colours
        for mem in (1 << 20, 512 << 20, 3 * (1 << 30) + 12345, 70 * (1 << 30)):
            p = host.table_plan_for_memory(mem, bits)
            assert p.bytes <= mem and 1 <= p.bucket_size <= 48
            assert p.nbuckets & (p.nbuckets - 1) == 0 and p.nbuckets >= 1024
            assert p.capacity == p.nbuckets * p.bucket_size
            # one more entry per bucket would not fit (or the buckets are at their 48-entry limit)
            if p.bucket_size < 48:
                assert ((p.bucket_size + 1) * p.nbuckets * bits) // 8 + p.nbuckets * 2 > mem
    # the reference's default: -m 512MB, k=31, one colour (104 bits per k-mer)
    p = host.table_plan_for_memory(512 << 20, 104)
    assert (p.nbuckets, p.bucket_size) == (1 << 20, 39) and p.capacity == 40894464


def test_build_decision_rules(host):
    bits = 104
    # -n wins over -m; the table holds at least n
    p, err = _build_plan(host, 16 << 30, True, 10 ** 9, True, bits)
    assert err is None and p.capacity == 33554432 * 30
    # ... but must fit -m, given or not (cmd_mem.c:120-123: `-n 1G` alone dies against the 512 MB default)
    p, err = _build_plan(host, 512 << 20, False, 10 ** 9, True, bits)
    assert err and "Not enough memory for requested graph" in err
    # -n that does not fit an explicit -m: the reference's message
    p, err = _build_plan(host, 1 << 30, True, 10 ** 9, True, bits)
    assert err and "Not enough memory for requested graph" in err
    # neither: the 512 MB default of -m is filled ...
    p, err = _build_plan(host, 512 << 20, False, 0, False, bits)
    assert err is None and p.capacity == 40894464
    # ... unless the inputs cannot hold that many k-mers (5 k-mers per input byte at occupancy 0.75)
    p, err = _build_plan(host, 512 << 20, False, 0, False, bits, max_kmers=3_000_000)
    assert err is None and p.capacity == host.table_plan_for_kmers(4_000_000, bits).capacity
    # an estimate never enlarges the table, and -n ignores it
    p, err = _build_plan(host, 512 << 20, False, 0, False, bits, max_kmers=10 ** 12)
    assert p.capacity == 40894464
    p, err = _build_plan(host, 512 << 20, False, 5000, True, bits, max_kmers=10)
    assert p.capacity == host.table_plan_for_kmers(5000, bits).capacity == 5120
    # never fewer than 1024 entries
    # a -m below the smallest table dies with the reference's message (its size_t arithmetic included)
    p, err = _build_plan(host, 1000, True, 0, False, bits)
    assert err and "Not enough memory for requested graph" in err
    p, err = _build_plan(host, 512 << 20, False, 0, False, bits, max_kmers=3)
    assert p.capacity == 1024


def test_memory_argument_units(host):
    """cmd.c:206-214 / util.c:206-222: 1024, 2MB, 1G ... binary units"""
    for text, want in [(b"1024", 1024), (b"2MB", 2 << 20), (b"1G", 1 << 30), (b"12K", 12 << 10), (b"3gb", 3 << 30)]:
        v = C.c_size_t()
        assert host.mem_to_integer(text, C.byref(v)) and v.value == want, text
    v = C.c_size_t()
    assert not host.mem_to_integer(b"12XB", C.byref(v))


def test_fastq_offset_table(host):
    """seq_file's seq_guess_fastq_format as the reference uses it (src/basic/seq_reader.c:252-288):
    the Sanger range wins for uniformly high Phred+33 qualities; Phred+64 families by their minimum."""
    host.fq_offset_from_range.restype = C.c_int
    host.fq_offset_from_range.argtypes = [C.c_int, C.c_int]
    cases = [((73, 73), 33),     # all 'I' (simulated reads): Sanger, not Phred+64
             ((70, 70), 33),     # all 'F' (binned NovaSeq qualities)
             ((33, 73), 33), ((35, 74), 33), ((40, 126), 33),
             ((64, 104), 64), ((66, 104), 64), ((67, 105), 64), ((59, 100), 64),
             ((64, 72), 33),     # inside the Sanger range: read as Sanger, as seq_file does
             ((80, 110), 33),    # nothing fits: offset 33
             ((255, 0), 0)]      # no qualities
    for (lo, hi), want in cases:
        assert host.fq_offset_from_range(lo, hi) == want, (lo, hi)
