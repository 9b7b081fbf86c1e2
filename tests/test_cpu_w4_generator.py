"""tools/gen_w4_ktile.py writes the K-tile bodies of the laboratory GEMMs (vit.cpp_amd/csrc/gemm_w4.hip) as inline assembly.  The GPU test
(tests/test_gpu_lab_gemm.py) checks the kernels' bits; this checks the generator's invariants on the CPU, so that an edit that drops a fragment
read, a DMA piece or a wait is caught where it is made:
  * per K-tile and wave: 2 x (8 TU) MFMAs (+ the C = 0 copy of block 0 in the B0 body), 2 x (8 + TU) ds_read_b128, 64 / NW LDS-DMA pieces, 1 barrier;
  * every accumulator tile is updated exactly once per block; every fragment of the set a block multiplies with was read in the block before;
  * the first-K-tile wait leaves exactly the bias pieces (+ the previous tile's stores) in flight; LDS-DMA targets tile the staged buffer exactly."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_w4", os.path.join(ROOT, "tools", "gen_w4_ktile.py"))
gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)


@pytest.mark.parametrize("NW", [4, 8])
@pytest.mark.parametrize("B", [0, 1])
def test_ktile_body_invariants(NW, B):
    TU, NM, NR, NP = 32 // NW, 8 * (32 // NW), 8 + 32 // NW, 64 // NW
    stores = 32 if NW == 4 else 16
    body = gen.ktile("bf16", NW, B, stores, 0)
    mf = [l for l in body if l.startswith("v_mfma")]
    rd = [l for l in body if l.startswith("ds_read_b128")]
    dm = [l for l in body if l.startswith("buffer_load_dwordx4")]
    assert len(mf) == (3 if B == 0 else 2) * NM and len(rd) == (3 if B == 0 else 2) * NR and len(dm) == NP
    assert body.count("s_barrier") == 1 and body[-1] == "s_waitcnt lgkmcnt(0)"
    # block structure: split at the barrier; B0 carries two copies of block 0 (accumulate / from zero)
    bar = body.index("s_barrier")
    blk1 = body[bar + 1:]
    acc = lambda l: l.split()[1].rstrip(",")
    assert sorted(acc(l) for l in blk1 if l.startswith("v_mfma")) == sorted(set(acc(l) for l in mf)) and len(set(acc(l) for l in mf)) == NM
    zero = [l for l in body[:bar] if l.startswith("v_mfma") and l.endswith(", 0")]
    assert len(zero) == (NM if B == 0 else 0)                      # the first K-tile of a tile starts every accumulator from C = 0
    # block 1 multiplies with set 1 (read before the barrier, from THIS buffer) and reads set 0 of the other buffer
    other = (B ^ 1) * 2 * gen.IMG
    for l in blk1:
        if l.startswith("v_mfma"):
            assert "%[w1_" in l and "%[a1_" in l
        if l.startswith("ds_read"):
            off = int(l.split("offset:")[1])
            assert re.search(r"%\[[aw]0_\d\]", l) and other <= off < other + 2 * gen.IMG
    reads0 = [l for l in body[:bar] if l.startswith("ds_read")]
    assert all(re.search(r"%\[[aw]1_\d\]", l) for l in reads0)
    names = {re.search(r"%\[([aw]1_\d)\]", l).group(1) for l in reads0}
    assert names == {f"a1_{t}" for t in range(8)} | {f"w1_{u}" for u in range(TU)}
    # LDS-DMA: the wave's pieces of both operands land at distinct KiB of THIS buffer's two 32 KiB blocks
    tgt = [int(l.split(",")[-1]) for l in body if l.startswith("s_add_u32 m0")]
    assert len(tgt) == NP and len(set(tgt)) == NP
    a_blk, w_blk = B * 2 * gen.IMG, 4 * gen.IMG + B * 2 * gen.IMG
    assert sorted(tgt) == sorted([a_blk + i * NW * 1024 for i in range(NP // 2)] + [w_blk + i * NW * 1024 for i in range(NP // 2)])
    # waits: the ordinary K-tile drains everything; the first K-tile of a tile lets the bias pieces (and an epilogue's stores) stay in flight
    waits = [l for l in body if l.startswith("s_waitcnt vmcnt")]
    nb = TU // 4
    if B == 0:
        assert waits == ["s_waitcnt vmcnt(0) lgkmcnt(0)", f"s_waitcnt vmcnt({min(nb + stores, 63)}) lgkmcnt(0)", f"s_waitcnt vmcnt({nb}) lgkmcnt(0)"]
    else:
        assert waits == ["s_waitcnt vmcnt(0) lgkmcnt(0)"]


def test_ablation_builds_drop_exactly_what_they_name():
    full = gen.ktile("bf16", 8, 1, 16, 0)
    for abl, prefix in ((4, "buffer_load"), (8, "ds_read"), (16, "v_mfma")):
        body = gen.ktile("bf16", 8, 1, 16, abl)
        assert not any(l.startswith(prefix) for l in body)
        assert [l for l in body if not l.startswith(("s_add_u32", "s_mov_b32"))] == [l for l in full if not l.startswith((prefix, "s_add_u32", "s_mov_b32"))]
    assert "s_barrier" not in gen.ktile("bf16", 8, 1, 16, 64)
