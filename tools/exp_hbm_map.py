#!/usr/bin/env python3
"""round 6: where in HBM is the split's write pattern fast?  Chunks of CHUNK GB allocated one after the other until the
part is nearly full, each probed (k_l2_probe: 16 K write fronts 481 KB apart, 128-byte runs, the whole depth); then all
freed and the same again, in reverse probing order: is the time a property of the place or of the moment?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mccortex_amd as mcx
L = mcx.lib()
L.mcx_debug_probe.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
jit = int(os.environ.get("JIT", "0"))  # words: every bin starts up to this much later (power of two minus one)
dev = torch.device("cuda", 0)
chunk_gb = float(os.environ.get("CHUNK", "16"))
cap = int(os.environ.get("CAP", "60156"))          # words per bin (12.42 G window: 60156)
nreg, spb = 32, 512
need = nreg * spb * cap * 8
iters_words = cap - jit - 1
words = int(chunk_gb * (1 << 30)) // 8
assert words * 8 >= need, (words * 8, need)
iters = (cap - int(os.environ.get("JITMAX", "0")) - 1) // 1024  # (the same for every jitter of a comparison)


def probe(t):
    ms = C.c_float(0)
    assert L.mcx_debug_probe(C.c_void_p(t.data_ptr()), cap, nreg, spb, iters, jit, C.byref(ms)) == 0
    return ms.value


for rnd in range(2):
    chunks = []
    while True:
        free, tot = torch.cuda.mem_get_info()
        if free < words * 8 + (6 << 30):
            break
        chunks.append(torch.empty(words, dtype=torch.int64, device=dev))
    order = range(len(chunks)) if rnd == 0 else reversed(range(len(chunks)))
    res = {}
    for i in order:
        res[i] = probe(chunks[i])
    print("round %d: %d chunks of %.0f GB: %s" % (rnd, len(chunks), chunk_gb, "  ".join("%d:%x:%.2f" % (i, chunks[i].data_ptr() >> 30, res[i]) for i in sorted(res))), flush=True)
    # twice more for the first four: repeatable?
    print("   again: %s" % "  ".join("%d:%.2f" % (i, probe(chunks[i])) for i in range(min(4, len(chunks)))), flush=True)
    del chunks
    torch.cuda.empty_cache()
