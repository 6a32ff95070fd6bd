#!/usr/bin/env python3
"""ISA lint for the gfx950 kernels: no vector store of more than 8 bytes per lane with a REGISTER in its scalar-offset field.

Why: between such a store and a VALU write of its data registers the hardware needs wait states; the compiler inserts them
— except when the scalar-offset field holds a register (LLVM's GCNHazardRecognizer::createsVALUHazard exempts that form).
On gfx950 the exempted form was seen to go out with the following instruction's result in its first data register
(attention_mfma_xrow.hip, round 3; csrc/attention_common.hpp: attn_store_line).  Keeping the field at 0 makes the compiler's
own hazard handling apply.

usage: isa_lint.py [file.hip ...]   (default: every csrc/*.hip); exit code 1 and one line per finding.
Compiles the device code of each file to assembly with hipcc (no GPU needed)."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dfd-clip_amd", "csrc")
STORE = re.compile(r"^\s*(buffer_store_dwordx[34]|buffer_store_b(96|128))\s+(.*)$")


def device_asm(path):
    hipcc = "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", "-", path]
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def findings(asm, name):
    out, kernel = [], "?"
    for n, line in enumerate(asm.splitlines(), 1):
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L"):
            kernel = m.group(1)
        m = STORE.match(line)
        if not m:
            continue
        ops = [o.strip() for o in m.group(3).split(";")[0].split(",")]
        # vdata, vaddr (or `off`), srsrc, soffset [modifiers]
        soff = ops[3].split()[0] if len(ops) > 3 else ""
        if re.fullmatch(r"s\d+|m0|s\[\d+:\d+\]|ttmp\d+", soff):
            out.append(f"{name}: {kernel}: asm line {n}: `{line.strip()}` has register {soff} in its scalar-offset field")
    return out


def main(argv):
    files = argv or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(max_workers=min(8, len(files))) as ex:
        asms = list(ex.map(device_asm, files))
    bad = [f for path, asm in zip(files, asms) for f in findings(asm, os.path.basename(path))]
    for f in bad:
        print(f)
    print(f"isa_lint: {len(files)} files, {len(bad)} findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
