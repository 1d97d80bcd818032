"""Builds libpfotgn.so (HIP, gfx950) in-tree with hipcc.  No torch extension machinery: the library
is a plain C-ABI shared object (include/pfotgn.h) that Python binds with ctypes."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.environ.get("PFO_CSRC") or os.path.join(HERE, "csrc")     # override: A/B build of another checkout's kernels
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpfotgn.so")
SOURCES = ["sampler.hip", "gemm.hip", "attn.hip", "memory.hip", "misc.hip", "csr.hip", "tgn.hip"]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, defines=(), tag=""):
    """``defines``/``tag`` build an A/B variant (lib/libpfotgn_<tag>.so) next to the default library."""
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj" + ("_" + tag if tag else ""))
    lib = os.path.join(LIBDIR, "libpfotgn%s.so" % ("_" + tag if tag else ""))
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "pfotgn.h"))
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if force or _newer(obj, [path] + headers):
            cmd = [hipcc, "-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17"] + ["-D" + d for d in defines] + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(7, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _newer(lib, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    tag = next((a[6:] for a in sys.argv[1:] if a.startswith("--tag=")), "")
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, defines=defs, tag=tag))
