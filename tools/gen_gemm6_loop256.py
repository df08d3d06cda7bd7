#!/usr/bin/env python3
"""Generates clover_amd/csrc/gemm6_loop256.inc: main loop + epilogue of k_m4_gemm_fp6_t256 (gemm6.hip) -- the 256 x 256 workgroup
tile of the FP6 GEMM -- as ONE inline-asm string per mode (scaled fp32 result / exact int32 sums).

Why a second tile shape: with the 128 x 128 tile the three costs of a stage add up on the SIMD issue path (24 MFMAs x 32 cycles,
24 folds x ~50, 18 LDS-DMA requests x ~45; profiles/r02_gemm_loop_experiments.txt).  A 256 x 256 tile stages (256 + 256) rows for
16 wave tiles instead of (128 + 128) for 4: HALF the LDS-DMA requests and L2 bytes per MFMA.  16 waves = one workgroup per CU = 4
waves per SIMD, so a wave has 128 VGPRs:

    v0..63     four 32x32 accumulators (tile T = 2 a + b at v[16T : 16T+15])
    v64..95    TWO result sets (the fold lags ONE MFMA behind: set m & 1)
    v96..119   fragments FA0 FA1 FB0 FB1 (6 registers each)
    v120 E8M0 scale word, v121 scratch, v122 c of the next second K-block (parked until s42 is free)
    + 5 operand registers (fragment base addresses a16 a8 b16 b8, DMA lane offset voff)      = 128

The block scales come through SMEM (no VGPRs): the s_load for the NEXT stage is issued behind the last counted LDS wait of a stage,
so the only wait that follows it is the stage's vmcnt(0) lgkmcnt(0) before the barrier (SMEM returns out of order: a counted
lgkmcnt wait behind it would be wrong).  The kernel is persistent (one workgroup per CU walks tiles): the store of C is issued
asynchronously and drains while the next tile starts.

Per stage (2 K-blocks = 8 units of one MFMA + one fold per wave), E/O tile visiting order as in gen_gemm6_loop.py:

    u0: MFMA  DMA#0  fold(prev u7)  s42 <- c1      u4: MFMA  fold(u3)
    u1: MFMA  FA0<-j1  DMA#1  fold(u0)             u5: MFMA  s_load next scales  fold(u4)
    u2: MFMA  FB1<-j1  DMA#2  fold(u1)             -- vmcnt(0) lgkmcnt(0), barrier, pointers, buffer toggle, FA0<-next, c0/c1 of next --
    u3: MFMA  FB0,FA1<-j1  fold(u2)                u6: MFMA  FB0<-next  fold(u5)       u7: MFMA  FB1,FA1<-next  fold(u6)

SGPRs named literally: s40/s42 c of the first / second K-block (s41, s43 pad the packed-fma pairs), s44:45 / s46:47 scales of A / B
(two K-blocks), s52:53 s54:55 s56:57 global address of the wave's three DMA pieces of the next stage, s58:59 / s72:73 address of the
next stage's scales of A / B, s60 stages left, s61 LDS address of the buffer the DMA writes to, s62 +-BUF, s63 row stride of C,
s64:65 scratch, s66:67 running row address of C, s68..70 the three pieces' offsets inside a stage buffer, s74 store flag.

THREE stage buffers (144 KiB): with one workgroup per CU nobody else fills the matrix pipe while 16 waves wait at a barrier for a
DMA, so a stage's requests go out TWO stages ahead and the wait before the barrier is vmcnt(3): the older three (next stage) have
landed, this stage's three stay in flight.  s62 = index of the buffer the fragment bases name, s75 = index of the DMA target.
"""
import os
import sys

SUB = 256 * 48            # one operand, one K-block: [row][48 B]
BUF = 4 * SUB             # stage image [A j0][A j1][B j0][B j1] = 48 KiB; three of them
STAGE_BYTES = 2 * SUB     # global bytes of one operand per stage
NS = int(os.environ.get("G6T_NS", "8"))     # scalar fmas per fold, the rest packed (even)
MODE = "scaled"
# timing-only experiment switches (results are wrong by construction): parts of the loop LEFT OUT
OFF = set()               # subset of {"dma", "lds", "barrier", "store", "fold"}
FOLD = os.environ.get("G6T_FOLD", "mixed")   # mixed: NS scalar v_fma + packed; fmac: 16 VOP2 v_fmac_f32; pk: 8 v_pk_fma_f32

FRAG = {"A0": 96, "A1": 102, "B0": 108, "B1": 114}
VS, VT, VC1 = 120, 121, 122


class Emit:
    def __init__(self):
        self.lines = []
        self.lds_q = []

    def __call__(self, s):
        self.lines.append(s)

    def ds_frag(self, frag, j, idx):
        r = FRAG[frag]
        op = "a" if frag[0] == "A" else "b"
        off = j * SUB + idx * 32 * 48
        if "lds" in OFF or ("lds25" in OFF and frag == "A1"):      # lds25: a QUARTER of the fragment reads left out (what a 128 x 64
            return                                                  # wave tile would save per MFMA: 6 fragments for 8 MFMAs instead of 8)
        self(f"ds_read_b128 v[{r}:{r+3}], %[{op}16] offset:{off}")
        self(f"ds_read_b64 v[{r+4}:{r+5}], %[{op}8] offset:{off}")
        self.lds_q += [frag, frag]

    def wait_frags(self, *frags):
        last = -1
        for i, f in enumerate(self.lds_q):
            if f in frags:
                last = i
        if last < 0:
            return
        n = len(self.lds_q) - 1 - last
        assert n <= 15
        self(f"s_waitcnt lgkmcnt({n})")
        self.lds_q = self.lds_q[last + 1:]


def mfma(e, m, fa, fb, tile):
    d, c = 64 + 16 * (m & 1), "0"
    if MODE == "i32":
        d = 16 * tile
        c = f"v[{d}:{d+15}]"
    e(f"v_mfma_scale_f32_32x32x64_f8f6f4 v[{d}:{d+15}], v[{FRAG[fa]}:{FRAG[fa]+5}], v[{FRAG[fb]}:{FRAG[fb]+5}], {c}, v{VS}, v{VS} op_sel_hi:[0,0,0] cbsz:2 blgp:2")


def fold(e, m, tile, creg):
    if MODE == "i32" or "fold" in OFF:
        return
    a, r = 16 * tile, 64 + 16 * (m & 1)
    if FOLD == "fmac":                         # VOP2: 4-byte encodings, acc = s * r + acc
        for i in range(16):
            e(f"v_fmac_f32 v{a+i}, s{creg}, v{r+i}")
        return
    ns = 0 if FOLD == "pk" else NS
    for i in range(ns):
        e(f"v_fma_f32 v{a+i}, s{creg}, v{r+i}, v{a+i}")
    for i in range(ns, 16, 2):
        e(f"v_pk_fma_f32 v[{a+i}:{a+i+1}], s[{creg}:{creg+1}], v[{r+i}:{r+i+1}], v[{a+i}:{a+i+1}] op_sel_hi:[0,1,1]")


def dma(e, k):
    if "dma" in OFF:
        return
    e(f"s_add_u32 m0, s61, s{68 + k}")
    e("s_nop 0")
    e(f"global_load_lds_dwordx4 %[voff], s[{52 + 2 * k}:{53 + 2 * k}]")


def load_scales(e):
    if MODE == "i32":
        return
    e("s_load_dwordx2 s[44:45], s[58:59], 0x0")
    e("s_load_dwordx2 s[46:47], s[72:73], 0x0")


def make_c(e):
    """c = f32(f32(sA * 1/49) * sB) of both K-blocks in s44:47: first -> s40, second -> v122 (parked)"""
    if MODE == "i32":
        return
    e(f"v_mov_b32 v{VT}, s44")
    e(f"v_mov_b32 v{VC1}, s45")
    e(f"v_mul_f32 v{VT}, 0x3ca72f05, v{VT}")
    e(f"v_mul_f32 v{VC1}, 0x3ca72f05, v{VC1}")
    e(f"v_mul_f32 v{VT}, s46, v{VT}")
    e(f"v_mul_f32 v{VC1}, s47, v{VC1}")
    e("s_nop 0")
    e(f"v_readfirstlane_b32 s40, v{VT}")


UNITS = [("A0", "B0", 0), ("A0", "B1", 1), ("A1", "B1", 3), ("A1", "B0", 2),
         ("A0", "B1", 1), ("A0", "B0", 0), ("A1", "B0", 2), ("A1", "B1", 3)]
CREG = [40, 40, 40, 40, 42, 42, 42, 42]
Q0 = ["A0", "A0", "B0", "B0", "B1", "B1", "A1", "A1"]


def advance_pointers(e, limit, images_only=False, scales_only=False):
    """the image pointers name the stage the NEXT DMA fetches, the scale pointers the stage whose scales are loaded next; both
    move on only while that following stage exists (s60 = stages left, the current one included)"""
    if "salu" in OFF:
        return
    ptrs = [] if scales_only else [(52, STAGE_BYTES), (54, STAGE_BYTES), (56, STAGE_BYTES)]
    if MODE != "i32" and not images_only:
        ptrs += [(58, 8), (72, 8)]
    for lo, step in ptrs:
        e(f"s_cmp_gt_u32 s60, {limit}")
        e(f"s_cselect_b32 s64, {step}, 0")
        e(f"s_add_u32 s{lo}, s{lo}, s64")
        e(f"s_addc_u32 s{lo+1}, s{lo+1}, 0")


def rotate(e):
    """three stage buffers: the fragment bases move on to the next buffer (s62 = index of the one they name), and so does the DMA
    target (s75 = its index, always two ahead)"""
    if "salu" in OFF:
        return
    e("s_add_u32 s62, s62, 1")
    e("s_cmp_eq_u32 s62, 3")
    e("s_cselect_b32 s64, s76, s77")          # back by two buffers from the third, else one on
    e("s_cselect_b32 s62, 0, s62")
    for op in ("a16", "a8", "b16", "b8"):
        e(f"v_add_u32 %[{op}], s64, %[{op}]")
    e("s_add_u32 s75, s75, 1")
    e("s_cmp_eq_u32 s75, 3")
    e("s_cselect_b32 s64, s76, s77")
    e("s_cselect_b32 s75, 0, s75")
    e("s_add_u32 s61, s61, s64")


def generate():
    e = Emit()
    nzero = 64 if MODE == "i32" else 96
    # ---------------- prologue (per tile; the kernel is persistent) ----------------
    e("s_barrier")                             # every wave is done with the previous tile's LDS reads before its buffers are refilled
    e("s_mov_b64 s[52:53], %[g0]")
    e("s_mov_b64 s[54:55], %[g1]")
    e("s_mov_b64 s[56:57], %[g2]")
    e("s_mov_b64 s[58:59], %[sa]")
    e("s_mov_b64 s[72:73], %[sb]")
    e("s_mov_b32 s60, %[np]")
    e("s_mov_b32 s61, %[lds]")
    e("s_mov_b32 s62, 0")                      # buffer the fragment bases name
    e("s_mov_b32 s75, 2")                      # buffer the DMA will write to once the loop runs
    e(f"s_mov_b32 s76, {-2 * BUF & 0xFFFFFFFF}")
    e(f"s_mov_b32 s77, {BUF}")
    e("s_mov_b32 s63, %[cstride]")
    e("s_mov_b64 s[66:67], %[cb]")
    e("s_mov_b32 s68, %[l0]")
    e("s_mov_b32 s69, %[l1]")
    e("s_mov_b32 s70, %[l2]")
    e("s_mov_b32 s74, %[flag]")
    for r in (40, 41, 42, 43):
        e(f"s_mov_b32 s{r}, 0")
    e(f"v_mov_b32 v{VS}, 0x82828282")
    for i in range(nzero):
        e(f"v_mov_b32 v{i}, 0")
    load_scales(e)
    for k in range(3):
        dma(e, k)                              # stage 0 -> buffer 0
    advance_pointers(e, 1, images_only=True)
    e(f"s_add_u32 s61, s61, {BUF}")
    for k in range(3):
        dma(e, k)                              # stage 1 (or stage 0 again when there is only one) -> buffer 1
    advance_pointers(e, 2, images_only=True)
    advance_pointers(e, 1, scales_only=True)
    e(f"s_add_u32 s61, s61, {BUF}")            # the DMA target is buffer 2 from here on
    e("s_waitcnt vmcnt(3) lgkmcnt(0)")         # stage 0 has landed (LDS-DMA returns in order); stage 1 may still be in flight
    e("s_barrier")
    make_c(e)
    e.ds_frag("A0", 0, 0)
    e.ds_frag("B0", 0, 0)
    e.ds_frag("B1", 0, 1)
    e.ds_frag("A1", 0, 1)
    # ---------------- one stage per iteration ----------------
    e("1:")
    for m in range(6):
        fa, fb, tile = UNITS[m]
        e.wait_frags(fa, fb)
        mfma(e, m, fa, fb, tile)
        if m == 0:
            dma(e, 0)
            fold(e, 7, UNITS[7][2], 42)        # previous stage's last unit (first stage: set 1 and s42 are zero)
            if MODE != "i32":
                e(f"v_readfirstlane_b32 s42, v{VC1}")
        elif m == 1:
            e.ds_frag("A0", 1, 0)
            dma(e, 1)
            fold(e, 0, UNITS[0][2], CREG[0])
        elif m == 2:
            e.ds_frag("B1", 1, 1)
            dma(e, 2)
            fold(e, 1, UNITS[1][2], CREG[1])
        elif m == 3:
            e.ds_frag("B0", 1, 0)
            e.ds_frag("A1", 1, 1)
            fold(e, 2, UNITS[2][2], CREG[2])
        elif m == 4:
            fold(e, 3, UNITS[3][2], CREG[3])
        else:
            e.wait_frags("A1")                 # the last counted LDS wait of the stage: SMEM may be outstanding from here on
            if "lds25" in OFF:
                e("s_waitcnt lgkmcnt(0)")
                e.lds_q = []
            assert not e.lds_q
            load_scales(e)                     # next stage's scales, covered by the wait before the barrier
            fold(e, 4, UNITS[4][2], CREG[4])
    e("s_waitcnt vmcnt(3) lgkmcnt(0)")         # the NEXT stage's image (requested a whole stage ago) has landed; this stage's requests fly on
    if "barrier" not in OFF:
        e("s_barrier")
    advance_pointers(e, 3, images_only=True)   # the DMA pointers run two stages ahead
    advance_pointers(e, 2, scales_only=True)
    rotate(e)
    e.ds_frag("A0", 0, 0)
    make_c(e)
    for m in (6, 7):
        fa, fb, tile = UNITS[m]
        mfma(e, m, fa, fb, tile)
        if m == 6:
            e.ds_frag("B0", 0, 0)
        else:
            e.ds_frag("B1", 0, 1)
            e.ds_frag("A1", 0, 1)
        fold(e, m - 1, UNITS[m - 1][2], CREG[m - 1])
    assert e.lds_q == Q0 or "lds" in OFF or "lds25" in OFF, e.lds_q
    e("s_sub_u32 s60, s60, 1")
    e("s_cmp_lg_u32 s60, 0")
    e("s_cbranch_scc1 1b")
    # ---------------- drain ----------------
    e("s_waitcnt lgkmcnt(0)")
    if MODE == "i32":
        e("s_nop 15")
        for i in range(64):
            e(f"v_cvt_i32_f32 v{i}, v{i}")
    else:
        e("s_nop 7")                           # the last MFMA's result: 12 states before the fold reads it (the wait above + these)
        e("s_nop 3")
        fold(e, 7, UNITS[7][2], 42)
    # the base registers go back to buffer 0 for the next tile, and no DMA may be in flight when its prologue refills the buffers
    e(f"s_mul_i32 s64, s62, {BUF}")
    for op in ("a16", "a8", "b16", "b8"):
        e(f"v_sub_u32 %[{op}], %[{op}], s64")
    e("s_waitcnt vmcnt(0)")
    # ---------------- store C (skipped by waves whose 64 x 64 tile lies outside the matrix) ----------------
    e("s_cmp_eq_u32 s74, 0")
    e("s_cbranch_scc1 3f")
    if "store" in OFF:
        e("s_branch 3f")
    # lane offset: (4 (lane >> 5)) rows + (lane & 31) columns; the result sets are free now
    e("v_mbcnt_lo_u32_b32 v64, -1, 0")
    e("v_mbcnt_hi_u32_b32 v64, -1, v64")
    e("v_lshrrev_b32 v65, 5, v64")
    e("s_lshl_b32 s64, s63, 2")
    e("v_mul_lo_u32 v65, v65, s64")
    e("v_and_b32 v64, 31, v64")
    e("v_lshl_add_u32 v64, v64, 2, v65")
    e("s_mul_i32 s65, s63, 5")
    first = True
    for a in range(2):
        for t in range(16):
            if not first:
                step = "s65" if (t & 3) == 0 else "s63"
                e(f"s_add_u32 s66, s66, {step}")
                e("s_addc_u32 s67, s67, 0")
            first = False
            e(f"global_store_dword v64, v{16 * (2 * a + 0) + t}, s[66:67] nt")
            e(f"global_store_dword v64, v{16 * (2 * a + 1) + t}, s[66:67] offset:128 nt")
    e("s_nop 1")                               # the stores have read their data before the next tile zeroes the accumulators
    e("3:")
    return e.lines


EXPERIMENTS = {1: {"dma"}, 2: {"lds"}, 3: {"dma", "lds"}, 4: {"barrier"}, 5: {"fold"}, 6: {"store"}, 7: {"dma", "lds", "barrier", "store"},
               8: {"dma", "store"}, 9: {"dma", "lds", "barrier", "store", "salu"}, 10: {"dma", "lds", "store", "salu"},
               # round 4: the bound of the one lever left (128 x 64 wave tiles at two waves per SIMD): what a quarter fewer fragment reads
               # could give at UNCHANGED occupancy (11), and together with a store that costs nothing (12)
               11: {"lds25"}, 12: {"lds25", "store"},
               # round 6: the bound of wave specialisation (loader waves issue every LDS-DMA request, the MFMA waves none): 1 = the requests
               # gone from the MFMA waves' stream, 13 = the requests AND the stage barriers gone (what loader waves could at the very most buy)
               13: {"dma", "barrier"}}


CANDIDATES = {1: {"FOLD": "fmac"}, 2: {"FOLD": "pk"}, 3: {"FOLD": "mixed"}}      # `candidates`: CORRECT alternative schedules in slots v1..


def write(out, variants, define_experiments):
    global MODE, OFF, FOLD
    base_fold = FOLD
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_gemm6_loop256.py -- do not edit; see that file for the schedule and the register map.\n")
        todo = [("G6T_LOOP_ASM", "scaled", set()), ("G6T_LOOP_ASM_I32", "i32", set())]
        if define_experiments:
            f.write("#define G6T_LOOP_EXPERIMENTS 1\n")
        for v, spec in variants.items():
            todo += [(f"G6T_LOOP_ASM_V{v}", "scaled", spec), (f"G6T_LOOP_ASM_I32_V{v}", "i32", spec)]
        for name, mode, spec in todo:
            MODE, OFF, FOLD = mode, spec, base_fold
            if isinstance(spec, dict):                      # a candidate: a CORRECT alternative schedule
                OFF, FOLD = set(), spec.get("FOLD", base_fold)
            lines = generate()
            f.write(f"#define {name} \\\n")
            for ln in lines:
                f.write('    "%s\\n" \\\n' % ln)
            f.write('    ""\n')
            print(f"{out}: {name}: {len(lines)} instructions")
        vregs = list(range(0, 123))
        sregs = list(range(40, 48)) + list(range(52, 78))
        f.write("#define G6T_LOOP_CLOBBERS " + ", ".join(f'"v{i}"' for i in vregs) + ", " + ", ".join(f'"s{i}"' for i in sregs) + ', "scc", "memory"\n')


def main():
    """no argument: clover_amd/csrc/gemm6_loop256.inc, the product's loops (libclover_hip.so; committed).
       <out> experiments: the same + the timing-only variants v1..v12 with parts LEFT OUT (wrong results by construction).  The build
            (clover_amd/build.py) generates this into clover_amd/lib/obj/gemm6_loop256_exp.inc and compiles it ONLY into the bench-only probe
            library tools/_build/libclover_hip_probe.so (-DCLV_GEMM_EXPERIMENTS), where CLV_GEMM_LOOP=vN selects a variant --
            bench.py's `gemm.ceiling` and tools/gemm_bench.py load that library explicitly; the product library has no such switch.
       <out> candidates: v1.. = CORRECT alternative schedules (CANDIDATES) for A/B runs through the same switch."""
    if len(sys.argv) > 2 and sys.argv[2] == "candidates":
        write(sys.argv[1], {v: CANDIDATES.get(v, {}) for v in range(1, 11)}, True)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "experiments":
        write(sys.argv[1], EXPERIMENTS, True)
        return
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "clover_amd", "csrc")
    write(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "gemm6_loop256.inc"), {}, False)


if __name__ == "__main__":
    main()
