#!/usr/bin/env python3
"""Generates vit.cpp_amd/csrc/gemm_w4_ktile.inc: the K-tile bodies of gemm_w4_kernel (gemm_w4.hip) as inline-assembly strings.

Why assembly: hipcc 7.2 cannot hold 64 accumulator tiles (256 AGPRs) + two fragment sets in place through a software-pipelined loop -- the
C++ build of the same schedule moved half the accumulators between VGPRs and AGPRs around every MFMA and spilled ~100 registers
(profiles/r06/w4_hipcc_attempt.txt).  The register ALLOCATION stays with the compiler (every register is an asm operand); only the
instruction order and the wait counts inside a K-tile are fixed here.

One K-tile (BK = 64) of one wave = 128 v_mfma_f32_16x16x32 on an 8 x 8 grid of accumulator tiles:
  block 0: 64 MFMAs on fragment set 0 (k-step 0); between them the 16 ds_read_b128 that fill set 1 (k-step 1, same LDS buffer)
  s_waitcnt vmcnt(V) lgkmcnt(0) ; s_barrier      -- every wave's pieces of the NEXT K-tile have landed, nobody reads this buffer any more
  block 1: 64 MFMAs on set 1; between them the 16 ds_read_b128 that fill set 0 from the next K-tile's buffer and the 16 LDS-DMA pieces
           (buffer_load_dwordx4 ... lds) of the K-tile two ahead into THIS buffer
  s_waitcnt lgkmcnt(0)

Operand names (the C++ side binds them, gemm_w4.hip W4_OPERANDS):
  c{t}_{u}   accumulator tile, A row tile t, W column tile u: "+v" for t < ACC_V_ROWS, "+a" beyond (the register file is unified: MFMA takes C / D
             from arch VGPRs and A / B from accumulator registers as well as the other way round)
  a0_{t} w0_{u} / a1_{t} w1_{u}   fragment sets 0 ("+a", live across statements) and 1 ("=&a" scratch): ds_read_b128 straight into accumulator registers
  ra{pz}{k2} rw{pz}{k2}   per-lane LDS read addresses (tile parity pz, k-step k2), ("v")
  va vw   per-lane global byte offsets of the wave's LDS-DMA pieces ("v");  ra_ rw_ = buffer resources ("s", 128 bit)
  soa sow = scalar byte offset of the staged K-tile in A / W;  a32 w32 = 32 rows in bytes;  ldsw = LDS base + wave * 1024;  st = scratch SGPR
"""
import sys

IMG = 16384
ACC_V_ROWS = 6      # accumulator row tiles 0..5 (48 tiles, 192 registers) live in arch VGPRs, 6..7 in accumulator registers (see gemm_w4.hip)


def ktile(dt, B, stores, abl, dma_every=4, nt_a=False):
    """abl bits: 4 no LDS-DMA, 8 no fragment reads, 16 no MFMAs, 64 no barrier.
    B = 0: the body also serves a tile's FIRST K-tile, selected at run time by the scalar operand `mode` (0 = not first, 1 = first, 2 = first
    and `stores` epilogue stores of the previous tile may still be in flight): block 0 exists twice (accumulate / start from C = 0), and the
    wait in front of the barrier lets the 2 bias pieces (and those stores) stay outstanding."""
    mf = "v_mfma_f32_16x16x32_" + dt
    L = []

    def mma(k2, j, zero):
        if abl & 16:
            return
        t, u = j >> 3, j & 7
        c = f"%[c{t}_{u}]"
        L.append(f"{mf} {c}, %[w{k2}_{u}], %[a{k2}_{t}], {'0' if zero else c}")

    def read(buf, k2, i):
        # order of use by the next block: A0, W0..W7, A1..A7
        if abl & 8:
            return
        if i == 0:
            kind, n = "a", 0
        elif i <= 8:
            kind, n = "w", i - 1
        else:
            kind, n = "a", i - 8
        off = buf * 2 * IMG + (n >> 1) * 4096
        L.append(f"ds_read_b128 %[{kind}{k2}_{n}], %[r{kind}{n & 1}{k2}] offset:{off}")

    def dma(p):
        if abl & 4:
            return
        img, i = p >> 2, p & 3
        if img < 2:
            lds = (B * 2 + img) * IMG + i * 4096
            so, step, vo, rs = "soa", "a32", "va", "ra_"
        else:
            lds = 4 * IMG + (B * 2 + img - 2) * IMG + i * 4096
            so, step, vo, rs = "sow", "w32", "vw", "rw_"
        L.append(f"s_add_u32 m0, %[ldsw], {lds}")
        if p in (0, 8):
            L.append(f"s_mov_b32 %[st], %[{so}]")
        else:
            L.append(f"s_add_u32 %[st], %[st], %[{step}]")
        L.append(f"buffer_load_dwordx4 %[{vo}], %[{rs}], %[st] offen lds" + (" nt" if (nt_a and img < 2) else ""))

    def block0(zero):
        for j in range(64):
            mma(0, j, zero)
            if (j & 1) and j < 32:
                read(B, 1, j >> 1)

    if B == 0:
        L.append("s_cmp_eq_u32 %[mode], 0")
        L.append("s_cbranch_scc0 .Lw4_first_%=")
        block0(False)
        L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
        L.append("s_branch .Lw4_join_%=")
        L.append(".Lw4_first_%=:")
        block0(True)
        L.append("s_cmp_eq_u32 %[mode], 1")
        L.append("s_cbranch_scc1 .Lw4_f1_%=")
        L.append(f"s_waitcnt vmcnt({min(2 + stores, 63)}) lgkmcnt(0)")
        L.append("s_branch .Lw4_join_%=")
        L.append(".Lw4_f1_%=:")
        L.append("s_waitcnt vmcnt(2) lgkmcnt(0)")
        L.append(".Lw4_join_%=:")
    else:
        block0(False)
        L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if not abl & 64:
        L.append("s_barrier")
    npieces = 0
    for j in range(64):
        mma(1, j, False)
        if (j & 1) and j < 32:
            read(B ^ 1, 0, j >> 1)
        if dma_every and (j % dma_every) == dma_every - 2 and npieces < 16:
            dma(npieces); npieces += 1
    while npieces < 16:
        dma(npieces); npieces += 1
    L.append("s_waitcnt lgkmcnt(0)")
    return L


def operands():
    out = []
    for t in range(8):
        for u in range(8):
            out.append(f'[c{t}_{u}] "+{"v" if t < ACC_V_ROWS else "a"}"(acc[{u >> 2}][{t}][{u & 3}])')
    for t in range(8):
        out.append(f'[a0_{t}] "+a"(fa0[{t}])')
    for u in range(8):
        out.append(f'[w0_{u}] "+a"(fw0[{u}])')
    for t in range(8):
        out.append(f'[a1_{t}] "=&a"(fa1[{t}])')
    for u in range(8):
        out.append(f'[w1_{u}] "=&a"(fw1[{u}])')
    out.append('[st] "=&s"(st_)')
    ins = []
    for pz in range(2):
        for k2 in range(2):
            ins.append(f'[ra{pz}{k2}] "v"(rdA[{pz}][{k2}])')
            ins.append(f'[rw{pz}{k2}] "v"(rdW[{pz}][{k2}])')
    ins += ['[va] "v"(a_voff)', '[vw] "v"(w_voff)', '[ra_] "s"(rsA)', '[rw_] "s"(rsW)', '[soa] "s"(so_a)', '[sow] "s"(so_w)',
            '[a32] "s"(a32)', '[w32] "s"(w32)', '[ldsw] "s"(ldsw)', '[mode] "s"(mode)']
    return out, ins


def main(path):
    o = ["// GENERATED by tools/gen_w4_ktile.py -- do not edit.  K-tile bodies of gemm_w4_kernel as inline-assembly strings.", "#pragma once"]
    outs, ins = operands()
    o.append("#define W4_OPERANDS \\\n    : " + ", \\\n      ".join(outs) + " \\\n    : " + ", \\\n      ".join(ins) + " \\\n    : \"memory\"")
    for dt, DT in (("bf16", "BF16"), ("f16", "F16")):
        for abl in (0, 4, 8, 12, 16, 64, 76):
            if abl and dt != "bf16":
                continue
            for name, B, stores in (("B0S32", 0, 32), ("B0S64", 0, 64), ("B1", 1, 0)):
                for nt in (0, 1):
                    if nt and abl:
                        continue
                    body = ktile(dt, B, stores, abl, nt_a=bool(nt))
                    o.append(f"#define W4_KT_{DT}_{name}_A{abl}_NT{nt} \\\n    " + " \\\n    ".join('"' + l + '\\n"' for l in body))
    with open(path, "w") as f:
        f.write("\n".join(o) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "vit.cpp_amd/csrc/gemm_w4_ktile.inc")
