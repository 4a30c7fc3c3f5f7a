"""Build the gfx950 C-ABI library in-tree: bayeformers_amd/lib/libbayeformers_amd.so.

    python -m bayeformers_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  Objects are cached under bayeformers_amd/csrc/_obj and rebuilt when a source or header is newer.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libbayeformers_amd.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

# The product library carries only code something dispatches to.  bf_fused_ws.hip — the weight-stationary single-kernel
# variant of NS-1, measured slower than both alternatives at every M (LABBOOK.md section 4.3, profiles/r4b_mid_m_crossover.txt)
# — and the round-1 GEMM kernel are DEV_SOURCES: built into libbayeformers_amd_dev.so only (tests/test_gpu_fused_ws.py and
# the tools/ micro-benchmarks select it with BF_LIB_PATH).
SOURCES = ["bf_api.hip", "bf_sample.hip", "bf_gemm.hip", "bf_gemm256.hip", "bf_gemm256_r5.hip", "bf_backward.hip", "bf_fused_small.hip", "bf_norm.hip", "bf_attention.hip", "bf_attention_bwd.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(INCLUDE, "bayeformers_amd.h"))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# Developer build (tools/ micro-benchmarks only): -DBF_DEV keeps the ablation / A-B environment switches
# (BF_GEMM_VARIANT, BF_GEMM_ABLATE, BF_GEMM_SCHED, BF_ATTN_ABLATE) and the round-1 GEMM kernel that the product
# library compiles out.  Select it with BF_LIB_PATH=bayeformers_amd/lib/libbayeformers_amd_dev.so.
DEV_LIB = os.path.join(LIBDIR, "libbayeformers_amd_dev.so")
DEV_SOURCES = ["bf_gemm256_r1.hip", "bf_fused_ws.hip"]


def build(force=False, verbose=True, dev=False):
    """Compile every HIP source for gfx950 and link the shared library.  Returns the library path."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = _headers()
    objs = []
    procs = []
    lib = DEV_LIB if dev else LIB
    for src in SOURCES + (DEV_SOURCES if dev else []):
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".dev.o" if dev else ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + (["-DBF_DEV"] if dev else []) + ["-c", s, "-o", o]
            if verbose:
                print("[bayeformers_amd.build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, pr in procs:
        out, _ = pr.communicate()
        if out.strip() and verbose:
            print(out)
        if pr.returncode != 0:
            print(out, file=sys.stderr)
            failed = True
    if failed:
        raise RuntimeError("hipcc failed")
    if force or _stale(lib, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs
        if verbose:
            print("[bayeformers_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, dev="--dev" in sys.argv))
