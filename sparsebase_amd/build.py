"""Builds sparsebase_amd/lib/libsbx.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build
container.  Objects are cached by source mtime under sparsebase_amd/lib/obj.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libsbx.so")
ARCH = "gfx950"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", os.path.join(ROOT, "include"),
         "-I", CSRC, "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "sbx.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, obj, verbose):
    cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_m = _headers_mtime()
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        dep_m = 0  # (the 64-bit twins include their 32-bit files)
        if src.endswith("sbx_rcm64.hip"):
            dep_m = os.path.getmtime(os.path.join(CSRC, "sbx_rcm.hip"))
        if src.endswith("sbx_gray64.hip"):
            dep_m = os.path.getmtime(os.path.join(CSRC, "sbx_gray.hip"))
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m, dep_m)
        if stale:
            jobs.append((src, obj))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(lambda so: _compile(so[0], so[1], verbose), jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


# Checking builds that travel with the product library (built by __graft_entry__.build(), loaded only by tests):
#   fenced: sbx_rcm.hip with -DSBX_GB_FENCED — release / acquire fences at every grid barrier and election (see gb_wait)
VARIANTS = {"fenced": (("sbx_rcm.hip", "sbx_rcm64.hip"), ["-DSBX_GB_FENCED"])}


def variant_path(name):
    return os.path.join(LIBDIR, f"libsbx_{name}.so")


def build_variant(name, verbose=False):
    """libsbx_<name>.so: the product's objects with one source recompiled under the variant's flags."""
    srcs, extra = VARIANTS[name]
    build(verbose=verbose)
    lib = variant_path(name)
    stale = False
    vobjs = []
    for src in srcs:
        srcpath = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, f"{src[:-4]}_{name}.o")
        vobjs.append(obj)
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(srcpath), _headers_mtime(),
                                                                  os.path.getmtime(os.path.join(CSRC, "sbx_rcm.hip"))):
            stale = True
            r = subprocess.run([HIPCC] + FLAGS + extra + ["-c", srcpath, "-o", obj], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src} ({name}):\n{r.stdout}\n{r.stderr}")
    if stale or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(LIB):
        objs = [os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o") for s in _sources()
                if os.path.basename(s) not in srcs]
        r = subprocess.run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs + vobjs + ["-ldl"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed ({name}):\n{r.stdout}\n{r.stderr}")
    return lib


def build_tuning(verbose=False):
    """libsbx_tuning.so: every source with -DSBX_TUNING — the build whose tuning and diagnostic environment switches
    (sbx_env_tuning in sbx_internal.h) are live.  For tools (SBX_PROBE_LIB=tuning); the product library ignores them."""
    odir = os.path.join(LIBDIR, "obj_tuning")
    os.makedirs(odir, exist_ok=True)
    hdr_m = _headers_mtime()
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(odir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        dep_m = 0  # (the 64-bit twins include their 32-bit files)
        if src.endswith("sbx_rcm64.hip"):
            dep_m = os.path.getmtime(os.path.join(CSRC, "sbx_rcm.hip"))
        if src.endswith("sbx_gray64.hip"):
            dep_m = os.path.getmtime(os.path.join(CSRC, "sbx_gray.hip"))
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m, dep_m):
            jobs.append((src, obj))

    def one(so):
        r = subprocess.run([HIPCC] + FLAGS + ["-DSBX_TUNING", "-c", so[0], "-o", so[1]], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {so[0]} (tuning):\n{r.stdout}\n{r.stderr}")
    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(one, jobs))
    lib = variant_path("tuning")
    if jobs or not os.path.exists(lib):
        r = subprocess.run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs + ["-ldl"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed (tuning):\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    if "--tuning" in sys.argv:
        print(build_tuning(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
