"""The laboratory GEMM structures of r06 (vit.cpp_amd/csrc/gemm_w4.hip: four waves / one per SIMD and eight free-running waves, K-tiles as generated
inline assembly) are not part of libvitx.so, but their record (profiles/r06/gemm_structures.md) claims they are CORRECT and keep the K order of the
product kernel.  This runs the laboratory binary on the small shapes (full tiles, an edge row block, K = 128 .. 512, both operand types, both epilogues)
and on one sub-batch-sized qkv launch: every line must pass the f64 check and equal the ping-pong kernel bit for bit."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "tools", "gemm_lab.bin")


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["w4", "w8"])
def test_free_running_gemms_match_the_product_kernel(torch_gpu, family):
    if not os.path.exists(LAB):
        pytest.skip("tools/gemm_lab.bin not built (make -C tools)")
    flt = ",".join(f"{s}:{family}:" for s in ("tiny", "edge", "ragged", "k256", "k128", "hqkv"))
    r = subprocess.run([LAB, "2", flt], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if f" {family} " in l]
    assert len(lines) >= 15, r.stdout[-3000:]
    for l in lines:
        assert " ok (" in l and "== pp bit for bit" in l, l
