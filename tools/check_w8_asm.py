#!/usr/bin/env python3
"""Build-time check of gemm_wx_kernel<NW = 8> (gemm_w4.hip): its 128 accumulator registers are written by NAME inside the K-tile inline
assembly and read by name in the epilogue -- the compiler is never told about the values (only that the registers are clobbered).  That is
sound only while hipcc itself emits NO instruction on an accumulator register in those kernels (no spill to AGPRs, no AV-class copies) and no
scratch.  This script compiles the file to assembly and fails if a compiler-emitted line of an NW = 8 kernel mentions a[..] / aN, or if such
a kernel uses scratch.  Run by `make -C vit.cpp_amd check_w8` (part of `all`).
"""
import re
import subprocess
import sys

src, inc = sys.argv[1], sys.argv[2:]
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", *inc, "--cuda-device-only", "-S", src, "-o", "-"],
                     capture_output=True, text=True)
if asm.returncode != 0:
    sys.exit("check_w8_asm: compile failed:\n" + asm.stderr[-2000:])
bad, kernels, cur, in_asm = [], 0, None, False
for line in asm.stdout.splitlines():
    m = re.match(r"^(_ZN4vitx14gemm_wx_kernel\S*Li8EEEv\S*):", line)
    if m:
        cur, in_asm = m.group(1), False; kernels += 1; continue
    if cur is None:
        continue
    if "s_endpgm" in line:
        cur = None; continue
    if "#ASMSTART" in line: in_asm = True; continue
    if "#ASMEND" in line: in_asm = False; continue
    code = line.split(";")[0]
    if not in_asm and (re.search(r"\ba\[\d+:\d+\]|\ba\d+\b|v_accvgpr", code) or "scratch_" in code):
        bad.append(f"{cur}: {line.strip()}")
if kernels == 0:
    sys.exit("check_w8_asm: no NW = 8 kernel found in the assembly")
if bad:
    sys.exit("check_w8_asm: compiler-emitted accumulator-register or scratch instructions in an NW = 8 kernel:\n  " + "\n  ".join(bad[:20]))
print(f"check_w8_asm: {kernels} kernels, no compiler-emitted accumulator-register instruction, no scratch")
