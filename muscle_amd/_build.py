"""Build libmuscle_hip.so (gfx950) in-tree with hipcc.  Called by __graft_entry__.build()."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmuscle_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value"]


HEADER = os.path.join(os.path.dirname(HERE), "include", "muscle_hip.h")


def abi_hash(path: str = HEADER) -> int:
    """31-bit hash of the public header's text; compiled into the library (mx_abi_hash) and re-derived by _lib.lib()."""
    import hashlib
    with open(path, "rb") as f:
        return int.from_bytes(hashlib.sha256(f.read()).digest()[:4], "little") & 0x7FFFFFFF


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    extra = os.environ.get("MUSCLE_EXTRA_FLAGS", "").split()                                             # A/B builds
    if any(f.startswith("-DMX_LAB") for f in extra):
        # timing-only switches that change results belong to tools/hip/gemm_lab.hip, never to the shipped library
        raise RuntimeError("MUSCLE_EXTRA_FLAGS: -DMX_LAB_* switches are lab-only (tools/hip/gemm_lab.hip); refusing to build them into libmuscle_hip.so")
    flags = FLAGS + [f"-DMX_ABI_HASH={abi_hash()}"] + extra
    # the flag list (incl. the header hash) is a staleness input: a changed flag or ABI rebuilds everything
    stamp = os.path.join(objdir, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    jobs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc, *flags, "-x", "hip", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose:
            print(f"[muscle_amd] compiled {os.path.basename(src)}", file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[muscle_amd] linked {LIB}", file=sys.stderr)
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
