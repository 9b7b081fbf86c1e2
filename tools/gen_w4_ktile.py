#!/usr/bin/env python3
"""Generates vit.cpp_amd/csrc/gemm_w4_ktile.inc: the K-tile bodies of gemm_w4_kernel (gemm_w4.hip) as inline-assembly strings.

Why assembly: hipcc 7.2 cannot hold 64 accumulator tiles (256 AGPRs) + two fragment sets in place through a software-pipelined loop -- the
C++ build of the same schedule moved half the accumulators between VGPRs and AGPRs around every MFMA and spilled ~100 registers
(profiles/r06/gemm_structures.md).  The register ALLOCATION stays with the compiler (every register is an asm operand); only the
instruction order and the wait counts inside a K-tile are fixed here.

One K-tile (BK = 64) of one wave = 128 v_mfma_f32_16x16x32 on an 8 x 8 grid of accumulator tiles:
  block 0: 64 MFMAs on fragment set 0 (k-step 0); between them the 16 ds_read_b128 that fill set 1 (k-step 1, same LDS buffer)
  s_waitcnt vmcnt(V) lgkmcnt(0) ; s_barrier      -- every wave's pieces of the NEXT K-tile have landed, nobody reads this buffer any more
  block 1: 64 MFMAs on set 1; between them the 16 ds_read_b128 that fill set 0 from the next K-tile's buffer and the 16 LDS-DMA pieces
           (buffer_load_dwordx4 ... lds) of the K-tile two ahead into THIS buffer
  s_waitcnt lgkmcnt(0)

Operand names (the C++ side binds them, gemm_w4.hip W4_OPERANDS):
  NW = 4:  c{t}_{u}   accumulator tile, A row tile t, W column tile u: "+v" for t < ACC_V_ROWS, "+a" beyond (the register file is unified: MFMA
           takes C / D from arch VGPRs and A / B from accumulator registers as well as the other way round); fragment sets "+a" / "=&a"
  NW = 8:  accumulator tile (t, u) IS a[4 (4 t + u) .. + 3]: all 128 accumulator registers of the wave, named in the text and clobbered (hipcc
           splits a 256-register wave 128 / 128 once accumulator registers are used at all); fragment sets in arch VGPRs ("+v" / "=&v")
  a0_{t} w0_{u} / a1_{t} w1_{u}   fragment sets 0 ("+a", live across statements) and 1 ("=&a" scratch): ds_read_b128 straight into accumulator registers
  ra{pz}{k2} rw{pz}{k2}   per-lane LDS read addresses (tile parity pz, k-step k2), ("v")
  va vw   per-lane global byte offsets of the wave's LDS-DMA pieces ("v");  ra_ rw_ = buffer resources ("s", 128 bit)
  soa sow = scalar byte offset of the staged K-tile in A / W;  ast wst = 8 NW rows in bytes (between a wave's pieces);  ldsw = LDS base + wave * 1024;  st = scratch SGPR
"""
import sys

IMG = 16384
ACC_V_ROWS = {4: 6, 8: 6}     # accumulator row tiles 0..5 live in arch VGPRs, 6..7 in accumulator registers (see gemm_w4.hip), per waves-per-workgroup


def ktile(dt, NW, B, stores, abl, dma_every=None, nt_a=False):
    """NW = waves per workgroup: 4 (2 x 2, 128 x 128 of C per wave, one wave per SIMD) or 8 (2 x 4, 128 x 64 per wave, two per SIMD).
    abl bits: 4 no LDS-DMA, 8 no fragment reads, 16 no MFMAs, 64 no barrier.
    B = 0: the body also serves a tile's FIRST K-tile, selected at run time by the scalar operand `mode` (0 = not first, 1 = first, 2 = first
    and `stores` epilogue stores of the previous tile may still be in flight): block 0 exists twice (accumulate / start from C = 0), and the
    wait in front of the barrier lets the bias pieces (and those stores) stay outstanding."""
    TU = 32 // NW                   # 16-column tiles per wave
    NM = 8 * TU                     # MFMAs per block
    NR = 8 + TU                     # fragment reads per set
    NP = 64 // NW                   # LDS-DMA pieces per wave and K-tile
    NBIAS = TU // 4                 # bias pieces (64 columns each) staged in front of a tile's first K-tile
    if dma_every is None:
        dma_every = NM // NP
    mf = "v_mfma_f32_16x16x32_" + dt
    L = []

    def mma(k2, j, zero):
        if abl & 16:
            return
        t, u = j // TU, j % TU
        c = f"%[c{t}_{u}]" if NW == 4 else f"a[{4 * (t * TU + u)}:{4 * (t * TU + u) + 3}]"
        L.append(f"{mf} {c}, %[w{k2}_{u}], %[a{k2}_{t}], {'0' if zero else c}")

    def read(buf, k2, i):
        # order of use by the next block: A0, W0..W(TU-1), A1..A7
        if abl & 8:
            return
        if i == 0:
            kind, n = "a", 0
        elif i <= TU:
            kind, n = "w", i - 1
        else:
            kind, n = "a", i - TU
        off = buf * 2 * IMG + (n >> 1) * 4096
        L.append(f"ds_read_b128 %[{kind}{k2}_{n}], %[r{kind}{n & 1}{k2}] offset:{off}")

    def dma(p):
        # the wave's piece i of an operand = physical KiB (wave + NW i) of the operand's 32 KiB block of this buffer: rows 8 NW i further down
        if abl & 4:
            return
        half = NP // 2
        op, i = p // half, p % half
        lds = (4 * IMG if op else 0) + B * 2 * IMG + i * NW * 1024
        so, step, vo, rs = (("sow", "wst", "vw", "rw_") if op else ("soa", "ast", "va", "ra_"))
        L.append(f"s_add_u32 m0, %[ldsw], {lds}")
        if i == 0:
            L.append(f"s_mov_b32 %[st], %[{so}]")
        else:
            L.append(f"s_add_u32 %[st], %[st], %[{step}]")
        L.append(f"buffer_load_dwordx4 %[{vo}], %[{rs}], %[st] offen lds" + (" nt" if (nt_a and not op) else ""))

    def block0(zero):
        for j in range(NM):
            mma(0, j, zero)
            if (j & 1) and (j >> 1) < NR:
                read(B, 1, j >> 1)

    if B == 0:
        L.append("s_cmp_eq_u32 %[mode], 0")
        L.append("s_cbranch_scc0 .Lwx_first_%=")
        block0(False)
        L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
        L.append("s_branch .Lwx_join_%=")
        L.append(".Lwx_first_%=:")
        block0(True)
        L.append("s_cmp_eq_u32 %[mode], 1")
        L.append("s_cbranch_scc1 .Lwx_f1_%=")
        L.append(f"s_waitcnt vmcnt({min(NBIAS + stores, 63)}) lgkmcnt(0)")
        L.append("s_branch .Lwx_join_%=")
        L.append(".Lwx_f1_%=:")
        L.append(f"s_waitcnt vmcnt({NBIAS}) lgkmcnt(0)")
        L.append(".Lwx_join_%=:")
    else:
        block0(False)
        L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if not abl & 64:
        L.append("s_barrier")
    npieces = 0
    for j in range(NM):
        mma(1, j, False)
        if (j & 1) and (j >> 1) < NR:
            read(B ^ 1, 0, j >> 1)
        if dma_every and (j % dma_every) == dma_every - 2 and npieces < NP:
            dma(npieces); npieces += 1
    while npieces < NP:
        dma(npieces); npieces += 1
    L.append("s_waitcnt lgkmcnt(0)")
    return L


def operands(NW):
    TU = 32 // NW
    vrows = ACC_V_ROWS[NW]
    out = []
    if NW == 4:
        for t in range(8):
            for u in range(TU):
                out.append(f'[c{t}_{u}] "+{"v" if t < vrows else "a"}"(acc[{u >> 2}][{t}][{u & 3}])')
    fc = "a" if NW == 4 else "v"        # NW = 8: the 128 accumulator registers ARE the accumulators (named in the text, clobbered), fragments in arch VGPRs
    for t in range(8):
        out.append(f'[a0_{t}] "+{fc}"(fa0[{t}])')
    for u in range(TU):
        out.append(f'[w0_{u}] "+{fc}"(fw0[{u}])')
    for t in range(8):
        out.append(f'[a1_{t}] "=&{fc}"(fa1[{t}])')
    for u in range(TU):
        out.append(f'[w1_{u}] "=&{fc}"(fw1[{u}])')
    out.append('[st] "=&s"(st_)')
    ins = []
    for pz in range(2):
        for k2 in range(2):
            ins.append(f'[ra{pz}{k2}] "v"(rdA[{pz}][{k2}])')
            ins.append(f'[rw{pz}{k2}] "v"(rdW[{pz}][{k2}])')
    ins += ['[va] "v"(a_voff)', '[vw] "v"(w_voff)', '[ra_] "s"(rsA)', '[rw_] "s"(rsW)', '[soa] "s"(so_a)', '[sow] "s"(so_w)',
            '[ast] "s"(a_step)', '[wst] "s"(w_step)', '[ldsw] "s"(ldsw)', '[mode] "s"(mode)']
    return out, ins


def main(path):
    o = ["// GENERATED by tools/gen_w4_ktile.py -- do not edit.  K-tile bodies of gemm_wx_kernel (gemm_w4.hip) as inline-assembly strings.", "#pragma once"]
    for NW in (4, 8):
        outs, ins = operands(NW)
        clob = ['"memory"'] + ([f'"a{i}"' for i in range(128)] if NW == 8 else [])
        o.append(f"#define WX{NW}_OPERANDS \\\n    : " + ", \\\n      ".join(outs) + " \\\n    : " + ", \\\n      ".join(ins) + " \\\n    : " + ", ".join(clob))
        if NW == 8:      # copy of accumulator tile idx (a[4 idx .. 4 idx + 3]) to VGPRs v0..v3, registers named in the text
            cases = [f'case {i}: asm volatile("v_accvgpr_read_b32 %0, a{4 * i}\\n\\tv_accvgpr_read_b32 %1, a{4 * i + 1}\\n\\tv_accvgpr_read_b32 %2, a{4 * i + 2}\\n\\tv_accvgpr_read_b32 %3, a{4 * i + 3}" '
                     f': "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3)); break;' for i in range(32)]
            o.append("#define WX8_ACC_READ_CASES \\\n    " + " \\\n    ".join(cases))
        s1, s2 = (32, 64) if NW == 4 else (16, 32)
        for dt, DT in (("bf16", "BF16"), ("f16", "F16")):
            for abl in (0, 4, 8, 12, 16, 64, 76):
                if abl and dt != "bf16":
                    continue
                for name, B, stores in (("B0S1", 0, s1), ("B0S2", 0, s2), ("B1", 1, 0)):
                    for nt in (0, 1):
                        if nt and abl:
                            continue
                        body = ktile(dt, NW, B, stores, abl, nt_a=bool(nt))
                        o.append(f"#define WX{NW}_KT_{DT}_{name}_A{abl}_NT{nt} \\\n    " + " \\\n    ".join('"' + l + '\\n"' for l in body))
    with open(path, "w") as f:
        f.write("\n".join(o) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "vit.cpp_amd/csrc/gemm_w4_ktile.inc")
