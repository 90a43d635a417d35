"""CPU tests of the boundary: the C-ABI library builds for gfx950, loads, exports every symbol
include/mcx_gpu.h declares, fails loudly without a GPU, and its host primitives (same templates
the kernels use) agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mcx_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcx_[a-z0-9_]+)\s*\(", src)) - {"mcx_sink_fn"})


def test_library_exports_every_declared_symbol(mcx):
    L = mcx.lib()
    syms = _declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(L, s), "libmcxgpu.so does not export %s" % s
    from mccortex_amd import graph
    assert sorted(graph.SYMBOLS) == syms


def test_no_cpu_fallback(mcx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mcx.McxError) as ei:
        mcx.Graph(31, 1, 1024)
    assert ei.value.code == -2


def test_product_does_not_touch_oracle():
    """The shipped path must not import/link/execute anything under oracle/."""
    bad = []
    for base in ("mccortex_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".c", ".cpp", ".hpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle|liborc|orc_", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


@pytest.mark.parametrize("k", [3, 11, 31, 33, 39, 51, 63])
def test_host_primitives_match_oracle(mcx, orc, k):
    L = orc.lib()
    W = L.orc_words_for_k(k)
    rng = np.random.default_rng(k)
    for _ in range(300):
        s = "".join("ACGTacgt"[i] for i in rng.integers(0, 8, k))
        x = L.orc_kmer_from_str(s.encode(), k)
        w = mcx.kmer_from_str(s, k)
        assert w == [int(x.b[i]) for i in range(W)]
        key = L.orc_kmer_get_key(x, k)
        kw, o = mcx.kmer_canonical(w, k)
        assert kw == [int(key.b[i]) for i in range(W)]
        assert o == (0 if kw == w else 1)
        iv = int(rng.integers(0, 2**32))
        assert mcx.kmer_hash(kw, k, iv) == L.orc_kmer_hash(key, k, iv)
        assert 0 <= mcx.key_owner(kw, k, 8) < 8


def test_ctx_header_matches_oracle(mcx, orc):
    import synth
    g = synth.genome(20000, 1)
    og = orc.Graph(31, 3, 1 << 16)
    hdr = mcx.CtxHeader(31, 3)
    names = ["a", "sample_two", "undefined"]
    for c in range(3):
        og.set_sample(c, names[c]); hdr.names[c] = names[c]
    for c, n, ln in [(0, 500, 100), (0, 300, 77), (2, 200, 150), (0, 1, 31)]:
        b, o = synth.reads(n, ln, seed=n, g=g, n_frac=0.2)
        st = og.add_reads(c, b, o)
        og.update_stats(c, st)
        hdr.update_stats(c, st.total_bases_loaded, st.contigs_parsed)
    assert mcx.ctx_header_bytes(hdr) == og.ctx_bytes(True)[:og.header_size()]


def test_host_packer_matches_the_base_codes(mcx):
    """mcx_pack_bases (the staging path's packer: AVX-512, AVX2 and the portable SWAR version): codes as
    src/basic/dna.c:8-25, validity = one of ACGTacgt; every byte value, random mixes."""
    import ctypes as C
    import os
    import subprocess
    import sys
    L = mcx.lib()
    L.mcx_pack_bases.restype = None
    L.mcx_pack_bases.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(3)
    src = np.concatenate([np.arange(256, dtype=np.uint8),
                          np.frombuffer(b"ACGTacgtNn\n\r .-*", dtype=np.uint8)[rng.integers(0, 16, 4096 + 32 - 256)]])  # (an odd number of 32-byte blocks: the AVX-512 tail)
    code = np.zeros(len(src) // 16, dtype=np.uint32)
    inv = np.zeros(len(src) // 16, dtype=np.uint16)
    L.mcx_pack_bases(src.ctypes.data, len(src), code.ctypes.data, inv.ctypes.data)
    lut = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("a"): 0, ord("c"): 1, ord("g"): 2, ord("t"): 3}
    for c in range(len(src) // 16):
        for j in range(16):
            ch = int(src[16 * c + j])
            bad = (int(inv[c]) >> (15 - j)) & 1
            assert bad == (ch not in lut), (c, j, ch)
            if ch in lut:
                assert (int(code[c]) >> (30 - 2 * j)) & 3 == lut[ch], (c, j, ch)
    # the portable version gives the same words
    prog = ("import sys, numpy as np, ctypes as C; sys.path.insert(0, %r); import mccortex_amd as m; L = m.lib(); "
            "L.mcx_pack_bases.restype = None; L.mcx_pack_bases.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]; "
            "s = np.frombuffer(sys.stdin.buffer.read(), np.uint8).copy(); c = np.zeros(len(s) // 16, np.uint32); i = np.zeros(len(s) // 16, np.uint16); "
            "L.mcx_pack_bases(s.ctypes.data, len(s), c.ctypes.data, i.ctypes.data); sys.stdout.buffer.write(c.tobytes() + i.tobytes())") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for off in ("MCX_NO_AVX2", "MCX_NO_AVX512"):   # SWAR; AVX2 where the host's default is AVX-512
        out = subprocess.run([sys.executable, "-c", prog], input=src.tobytes(), stdout=subprocess.PIPE, env=dict(os.environ, **{off: "1"})).stdout
        assert out == code.tobytes() + inv.tobytes(), off


def test_one_pass_read_packer_matches_assemble_then_pack(mcx):
    """mcx_pack_reads_host: the staging threads' one-pass packer (reads -> 2-bit codes + invalid flags, round 4)
    against the stream assembled as ASCII and packed block by block, on ragged reads of arbitrary bytes: empty reads,
    reads shorter than one step, lengths around the 32 / 64 / 128 boundaries, every byte value."""
    import ctypes as C
    L = mcx.lib()
    L.mcx_pack_reads_host.restype = C.c_uint64
    L.mcx_pack_reads_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"ACGTacgtNn\n\r .-*\x00\xff", dtype=np.uint8)
    for case in range(12):
        if case == 0:
            lens = np.array([0, 1, 0, 31, 32, 33, 63, 64, 65, 127, 128, 129, 150, 0, 0, 191, 192, 193, 1000, 0], dtype=np.int64)
        elif case == 1:
            lens = np.full(3000, 150, dtype=np.int64)
        elif case == 2:
            lens = np.array([200000], dtype=np.int64)
        elif case == 3:
            lens = np.zeros(0, dtype=np.int64)
        else:
            lens = rng.integers(0, [4, 40, 151, 300, 70, 2000, 10, 129][case % 8], size=int(rng.integers(1, 2500))).astype(np.int64)
        off = np.zeros(len(lens) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
        n = int(off[-1])
        bases = (alphabet[rng.integers(0, len(alphabet), n)] if case % 2 else rng.integers(0, 256, n).astype(np.uint8))
        bases = np.concatenate([bases, np.zeros(1, np.uint8)])[:n].copy()  # exactly n bytes: a step's load must not run past the end
        cap = 128 + ((n + len(lens) + 63) // 64) * 64
        out = []
        for fused in (0, 1):
            code = np.zeros(cap // 16, dtype=np.uint32)
            inv = np.zeros(cap // 16, dtype=np.uint16)
            total = L.mcx_pack_reads_host(bases.ctypes.data if n else None, off.ctypes.data, len(lens), code.ctypes.data, inv.ctypes.data, cap, fused)
            if fused and total == 0:
                pytest.skip("no AVX-512 on this host: the one-pass packer is not used")
            assert total == cap
            out.append((code, inv))
        assert np.array_equal(out[0][1], out[1][1]), "invalid flags, case %d" % case
        # codes only have to agree where a position is valid; the packers agree everywhere, which is simpler to state
        assert np.array_equal(out[0][0], out[1][0]), "codes, case %d" % case


def test_multi_exchange_bytes_sizing(mcx):
    """mcx_multi_exchange_bytes (no GPU needed): what the build command adds to the table when it checks -m / -n against
    the free HBM of every device of a split table"""
    assert mcx.multi_exchange_bytes(31, 1, 1 << 30) == 0
    v3 = mcx.multi_exchange_bytes(31, 8, 1 << 30)
    assert 8e9 < v3 < 11e9                                   # 3 send + 3 x 8 receive sets of a 128 Mi-position piece
    assert mcx.multi_exchange_bytes(63, 8, 1 << 30) > 1.8 * v3  # 32-byte records
    assert 9e9 < mcx.multi_exchange_bytes(21, 8, 1 << 30) < 13e9  # k < 29: format v2 (packed tuples)
    with pytest.raises(mcx.McxError):
        mcx.multi_exchange_bytes(31, 3, 1 << 30)
