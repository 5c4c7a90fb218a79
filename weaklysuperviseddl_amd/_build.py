"""Build libwsdl_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

In-tree build: objects under csrc/build/, the shared library next to the sources so that it travels to
the GPU box with the repository snapshot.  Rebuilds only what is older than its sources.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libwsdl_hip.so")
ARCH = "gfx950"
SOURCES = ["common.hip", "plan.hip", "conv_igemm.hip", "norm_pool.hip", "resample_loss.hip", "layercam_optim.hip", "lovasz.hip", "components.hip"]
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "wsdl_hip.h"))
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        if force or _newer(obj, [sp] + headers):
            jobs.append((sp, obj))

    def cc(job):
        sp, obj = job
        cmd = [hipcc] + FLAGS + ["-c", sp, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (sp, r.stderr[-4000:]))
        if verbose:
            print("[wsdl build] compiled", os.path.basename(sp), flush=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(bdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _newer(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("[wsdl build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
