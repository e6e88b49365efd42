"""Builds libsegearth_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m clip_decontamination_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the repo snapshot.  Objects are cached under csrc/_obj by source
mtime so a one-file edit recompiles one file.
"""
from __future__ import annotations

import argparse
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsegearth_hip.so")
ARCH = "gfx950"
SOURCES = ["capi.hip", "gemm_bf16.hip", "gemm_f32.hip", "attention.hip", "attention_f16.hip", "attention_h2.hip", "rowops.hip", "patchify.hip", "refine.hip",
           "head.hip", "jbu.hip", "ctd.hip"]
# the attention loops count vector-issue slots between MFMAs: SLP-packed v_pk_add_f32 / v_pk_mul_f32 cost several plain f32 ops there
# (MI355X_MICROARCH.md, 'price of one filler beside MFMAs'), so the scalar f32 arithmetic of those units stays scalar.
# jbu.hip: with the Keff arithmetic of jbu_conv_lowres_kernel SLP-packed into v_pk_fma_f32 the kernel was not reproducible from run to
# run (rare pixels, always the high half of lanes 48..63, only once co-resident workgroups ran different phases); the scalar build is
# bit-reproducible (tests/test_gpu_repro.py; DESIGN.md section 4 'JBU reproducibility')
_NO_SLP = ["-fno-slp-vectorize"]
EXTRA_FLAGS = {"attention.hip": _NO_SLP, "attention_f16.hip": _NO_SLP, "attention_h2.hip": _NO_SLP, "jbu.hip": _NO_SLP}
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built on this machine")
    return exe


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "segearth_hip.h"))
    hs.append(os.path.abspath(__file__))              # the compile flags live here
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    sp = os.path.join(CSRC, src)
    newest = max(os.path.getmtime(sp), _newest_header())
    with open(sp) as f:                               # a unit that re-compiles another .hip (attention_f16.hip) depends on it too
        for line in f:
            if line.startswith('#include "') and line.rstrip().endswith('.hip"'):
                newest = max(newest, os.path.getmtime(os.path.join(CSRC, line.split('"')[1])))
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj
    cmd = [hipcc(), *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB) from {len(objs)} objects")
    elif verbose:
        print(f"{LIB} is up to date")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    build(ap.parse_args().force)
