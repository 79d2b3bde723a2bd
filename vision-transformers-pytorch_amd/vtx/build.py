"""Build libvtx.so (gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build().

hipcc cross-compiles without a GPU; the resulting .so sits next to this file (git-ignored,
shipped to the GPU box by gpurun).  No JIT, no torch.utils.cpp_extension: plain C ABI.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libvtx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-value"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


_HIPCC_VERSION = None


def hipcc_version():
    """First line of ``hipcc --version`` (cached): part of every object's stamp -- the ISA-level guards of the library (wait states hipcc
    does not insert, profiles/round4_nondeterminism_root_cause.md, round5_mfma_branch_hazard.md) were validated against ONE compiler."""
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
            _HIPCC_VERSION = " | ".join(l.strip() for l in out.splitlines()[:2])
        except OSError:
            _HIPCC_VERSION = "unknown"
    return _HIPCC_VERSION


def _stamp(src):
    h = hashlib.sha1()
    h.update(hipcc_version().encode())
    for f in [src] + sorted(x for x in os.listdir(CSRC) if x.endswith(".h")):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stampf = obj + ".stamp"
    stamp = _stamp(src)
    if os.path.exists(obj) and os.path.exists(stampf) and open(stampf).read() == stamp:
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(stampf, "w") as fh:
        fh.write(stamp)
    return obj, True


def build(verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(_compile, _sources()))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[vtx] built {LIB} from {len(objs)} objects ({hipcc_version()})", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build()
