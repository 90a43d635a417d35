"""Build the gfx950 extension in-tree: mccortex_amd/libmcxgpu.so (hipcc cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmcxgpu.so")
SOURCES = ["mcx_api.hip", "mcx_kernels.h", "mcx_kmer.h", "mcx_defer.h", "mcx_superk.h", "mcx_streamfc.h", "mcx_multi.h", "mcx_ubench.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "..", "include", "mcx_gpu.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           os.path.join(CSRC, "mcx_api.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
