"""In-tree build of libdygnn_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m dyglib_amd._build [--force]

The library is built next to its sources (dyglib_amd/csrc/libdygnn_hip.so): it is git-ignored but
travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys
from typing import List

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
LIB_NAME = "libdygnn_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
SOURCES = ["csr_host.cpp", "sampler.hip", "cooccurrence.hip", "dygformer_generic.hip", "dygformer_fused3.hip", "dygformer_train.hip",
           "dygformer_api.hip", "tgat.hip", "tgat_chain.hip", "metrics.hip"]
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-ffp-contract=off"]   # contractions are written explicitly (fmaf) where the oracle has them


# "stamps": diagnostic build with in-kernel s_memtime phase stamps (tools/phase_profile.py)
# "asan":   HOST code (CSR builder, argument validation, packing / planning code of every entry point) under AddressSanitizer +
#           UndefinedBehaviorSanitizer; hipcc leaves the gfx950 code objects unsanitized (GPU ASan needs xnack+, unavailable here).
#           Driven by tests/test_sanitizers_cpu.py on the CPU box, never loaded by the product path.
SANITIZE = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-shared-libsan", "-Wno-option-ignored"]
VARIANTS = {"": [], "stamps": ["-DDYGNN_STAMPS=1"], "asan": SANITIZE,
            # A/B arm of the fused DyGFormer kernel (tools/ab_fused3.py): its build-time feature switched off
            "f3noskip": ["-DF3_KSKIP=0"]}
for _k, _v in list(os.environ.items()):          # ad-hoc arms: DYGNN_VARIANT_<name>="-DX=1 -DY=2"
    if _k.startswith("DYGNN_VARIANT_"):
        VARIANTS[_k[len("DYGNN_VARIANT_"):].lower()] = _v.split()
LINK_EXTRA = {"asan": ["-fsanitize=address,undefined", "-shared-libsan"]}


def asan_runtime() -> str:
    """clang's shared ASan runtime (to LD_PRELOAD into a process that dlopens the asan variant)."""
    import glob
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not hits:
        raise RuntimeError("libclang_rt.asan-x86_64.so not found under /opt/rocm/lib/llvm")
    return hits[0]


def lib_path(variant: str = "") -> str:
    return LIB_PATH if not variant else os.path.join(CSRC, f"libdygnn_hip_{variant}.so")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _digest() -> str:
    h = hashlib.sha256()
    for root in (CSRC, INCLUDE):
        for fn in sorted(os.listdir(root)):
            if fn.endswith((".hip", ".cpp", ".h")):
                with open(os.path.join(root, fn), "rb") as f:
                    h.update(fn.encode()), h.update(f.read())
    h.update(" ".join(CXXFLAGS).encode())
    return h.hexdigest()


def _src_digest(src: str, extra=()) -> str:
    """One source + every header + the flags: an object is reused while none of them changed."""
    h = hashlib.sha256()
    for root in (CSRC, INCLUDE):
        for fn in sorted(os.listdir(root)):
            if fn.endswith(".h") or (root == CSRC and fn == src):
                with open(os.path.join(root, fn), "rb") as f:
                    h.update(fn.encode()), h.update(f.read())
    h.update(" ".join([*CXXFLAGS, *extra]).encode())
    return h.hexdigest()


def _compile(src: str, obj_dir: str, extra=()) -> str:
    obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
    tag, dig = obj + ".digest", _src_digest(src, extra)
    if os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == dig:
        return obj
    cmd = [_hipcc(), *CXXFLAGS, *extra, "-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
    if src.endswith(".cpp"):
        cmd.insert(1, "-x"), cmd.insert(2, "hip")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip() and os.environ.get("DYGNN_BUILD_WARNINGS"):
        sys.stderr.write(r.stderr)
    with open(tag, "w") as f:
        f.write(dig)
    return obj


def build(force: bool = False, verbose: bool = True, variant: str = "") -> str:
    out = lib_path(variant)
    obj_dir = os.path.join(CSRC, "build" + ("_" + variant if variant else ""))
    stamp = os.path.join(obj_dir, "digest.txt")
    digest = _digest()
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == digest:
        return out
    os.makedirs(obj_dir, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        objs: List[str] = list(ex.map(lambda s: _compile(s, obj_dir, VARIANTS[variant]), SOURCES))
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *LINK_EXTRA.get(variant, []), *objs, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(digest)
    if verbose:
        print(f"built {out}")
    return out


if __name__ == "__main__":
    variants = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")] or [""]
    for v in variants:
        build(force="--force" in sys.argv, variant=v)
