"""Builds libdfdclip_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU, so the build runs in the CPU
container and the resulting .so travels to the GPU box with the source snapshot.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(PKG_DIR, "libdfdclip_hip.so")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# Kernels whose hand-counted `s_waitcnt vmcnt(N)` are exact only while the compiler adds no vector-memory operation of
# its own: a register spill (scratch_store / scratch_load counts in vmcnt) would silently turn a wait into a race.  The
# build asks the compiler for its resource report on these files and refuses a kernel with scratch.
# (gemm256p.hip's fp8 forms reload two registers from scratch in their epilogue; its K loop waits for vmcnt(0) every
# step, so that only over-waits there.  gemm256e.hip's loop never waits for zero.)
NO_SCRATCH = {"gemm256e.hip": "gemm256e_kernel", "attention_mfma_xrow.hip": "attn_mfma_xrow_kernel"}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP kernels cannot be built")
    return exe


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def scratch_users(report, kernel_substr):
    """From hipcc's -Rpass-analysis=kernel-resource-usage remarks: {kernel: bytes/lane} for kernels using scratch."""
    import re
    out, name = {}, None
    for line in report.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and kernel_substr in name and int(m.group(1)) > 0:
            out[name] = int(m.group(1))
    return out


def build(force=False, verbose=False, extra_flags=()):
    """Compile every .hip under csrc/ and link the shared library.  Returns its path.
    Incremental: an object is rebuilt when its source, a header or the flags changed."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")]
    headers.append(os.path.join(INCLUDE, "dfdclip.h"))
    jobs, objs = [], []
    for src in sources():
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJ_DIR, src[:-4] + ".o")
        stamp = obj + ".sha"
        dig = _digest([sp] + headers) + " ".join(extra_flags)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((sp, obj, stamp, dig))

    def compile_one(job):
        sp, obj, stamp, dig = job
        guard = NO_SCRATCH.get(os.path.basename(sp))
        cmd = [hipcc, *FLAGS, *extra_flags, *(["-Rpass-analysis=kernel-resource-usage"] if guard else []), "-c", sp, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {sp}:\n{r.stderr}")
        if guard:
            bad = scratch_users(r.stderr, guard)
            if bad:
                raise RuntimeError(f"{sp}: kernels with hand-counted vmcnt waits must not use scratch, but: {bad}")
        elif verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(stamp, "w") as f:
            f.write(dig)
        return sp

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for done in ex.map(compile_one, jobs):
                if verbose:
                    print("built", os.path.basename(done))
    if jobs or not os.path.exists(LIB_PATH):
        r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH, *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
