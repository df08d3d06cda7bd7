#!/usr/bin/env python3
"""Generates clover_amd/csrc/gemm6_loop.inc: the hand-scheduled main loop + epilogue of k_m4_gemm_fp6_asm (gemm6.hip) as ONE
inline-asm string, so that hipcc cannot re-schedule it (measurements behind the schedule: profiles/r02_mfma_fold_probe*.txt).

What a wave does per stage (= two K-blocks of its 64x64 tile = 8 "units" of one 32x32x64 FP6 MFMA + the fold of its result):

    unit m:   [wait for the fragments of unit m]   MFMA(m) -> result set m % 4
              [fragment reloads / one LDS-DMA request / scale bookkeeping that belong here]
              fold(m - 2): acc[tile] = fma(c, result, acc[tile]) -- NS scalar v_fma_f32 first (they issue in the shadow of the MFMA
                           that has just been issued), the rest as v_pk_fma_f32 (twice the work per issue, but serialised with the
                           matrix pipe: measured, r02_mfma_fold_probe2)

The fold lags two MFMAs behind, so the 12 wait states an 8-pass MFMA result needs before a VALU may read it are always covered by
real work.  Fragments are single-buffered: an operand register set is re-loaded right after the last MFMA that reads it, and the
visiting order of the four tiles alternates between the two K-blocks so that every re-load has at least one whole unit of lead.

Registers named literally (all listed as clobbers):
    v0..63     four 32x32 accumulators, tile T = 2 a + b at v[16T : 16T+15]
    v64..127   four result sets
    v128..151  fragments FA0 FA1 FB0 FB1 (6 registers each: ds_read_b128 + ds_read_b64)
    v152       E8M0 scale word 2^3 (both operands): magnitude/8 codes come back as integers
    v153       c of the NEXT second K-block, parked until s42's last reader has run;  v154 = 0;  v155 scratch
    v156:159   scales of the stage being fetched (A j0, A j1, B j0, B j1), loaded by VMEM one stage ahead
    s40/s42    c of the stage's first / second K-block (s41, s43: the unused halves of the packed-fma operand pairs)
    s52:53 / s54:55   global address of the next stage image of A / B;  s56:57 / s58:59 address of the next stage's scales
    s60 stages left, s61 LDS address the DMA writes to (the other buffer), s62 +-BUF (buffer toggle), s63 row stride of C in bytes,
    s64:65 scratch (DMA chunk address; 5 * s63 in the epilogue), s66:67 running row address of C

One barrier per stage, placed after unit 5: by then the next stage image has landed (its 6 DMA requests went out in units 0-2) and
every fragment load of this stage's buffer has been issued AND waited for, so after it (a) the next stage's fragments can be
requested while units 6, 7 still compute -- no LDS latency is exposed at the stage boundary -- and (b) the DMA of the stage after
next may overwrite this buffer.
"""
import sys

SUB = 128 * 48            # one operand, one K-block: [row][48 B]
BUF = 4 * SUB             # stage image [A j0][A j1][B j0][B j1]
STAGE_BYTES = 2 * SUB     # global bytes of one operand per stage
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 8      # scalar fmas per fold (even), the other 16 - NS elements go packed
# experiment switches (timing only, results are wrong): which parts of the loop are left out
SKIP = set()
MODE = "scaled"         # "i32": no scales, the MFMAs accumulate across K-blocks (exact integers), C is converted to int32

FRAG = {"A0": 128, "A1": 134, "B0": 140, "B1": 146}
VS, VC1, VZERO, VT = 152, 153, 154, 155          # scale word, c of the second K-block (kept in a VGPR until s42 is free), 0, scratch
VSCALE = 156                                     # v156:157 = A scales (j0, j1), v158:159 = B scales of the stage being fetched


class Emit:
    def __init__(self):
        self.lines = []
        self.lds_q = []           # outstanding ds_read instructions, oldest first (fragment names)
        self.in_loop = False

    def __call__(self, s):
        self.lines.append(s)

    def ds_frag(self, frag, j, idx):
        """fragment `frag` (A0/A1/B0/B1) <- K-block j of the buffer the base registers name, 32-row tile idx"""
        r = FRAG[frag]
        op = "a" if frag[0] == "A" else "b"
        off = j * SUB + idx * 32 * 48
        if "lds" in SKIP and self.in_loop:
            return
        self(f"ds_read_b128 v[{r}:{r+3}], %[{op}16] offset:{off}")
        self(f"ds_read_b64 v[{r+4}:{r+5}], %[{op}8] offset:{off}")
        self.lds_q += [frag, frag]

    def wait_frags(self, *frags):
        """wait until the named fragments have landed (LDS returns in order; no SMEM is ever outstanding here)"""
        last = -1
        for i, f in enumerate(self.lds_q):
            if f in frags:
                last = i
        if last < 0:
            return
        n = len(self.lds_q) - 1 - last
        assert n <= 15
        if not ("lds" in SKIP and self.in_loop):
            self(f"s_waitcnt lgkmcnt({n})")
        self.lds_q = self.lds_q[last + 1:]


def mfma(e, m, fa, fb, tile=None):
    d = 64 + 16 * (m % 4)
    c = "0"
    if MODE == "i32":            # accumulate in place: sums of integers below 2^24 are exact in fp32 whatever the internal order
        d = 16 * tile
        c = f"v[{d}:{d+15}]"
    e(f"v_mfma_scale_f32_32x32x64_f8f6f4 v[{d}:{d+15}], v[{FRAG[fa]}:{FRAG[fa]+5}], v[{FRAG[fb]}:{FRAG[fb]+5}], {c}, v{VS}, v{VS} op_sel_hi:[0,0,0] cbsz:2 blgp:2")


def fold(e, m, tile, creg):
    a, r = 16 * tile, 64 + 16 * (m % 4)
    if ("fold" in SKIP and e.in_loop) or MODE == "i32":
        return
    for i in range(NS):
        e(f"v_fma_f32 v{a+i}, s{creg}, v{r+i}, v{a+i}")
    for i in range(NS, 16, 2):
        e(f"v_pk_fma_f32 v[{a+i}:{a+i+1}], s[{creg}:{creg+1}], v[{r+i}:{r+i+1}], v[{a+i}:{a+i+1}] op_sel_hi:[0,1,1]")


def dma(e, k):
    """request k of the 6 that bring the NEXT stage image into the other buffer: A chunk i = k // 2 (k even) or B chunk i (k odd)"""
    i, is_b = k // 2, k & 1
    lo = 54 if is_b else 52
    if "dma" in SKIP and e.in_loop:
        return
    if "dmaA" in SKIP and e.in_loop and not is_b:
        return
    off = 4096 * i + (2 * SUB if is_b else 0)
    if i:
        e(f"s_add_u32 s64, s{lo}, {4096 * i}")
        e(f"s_addc_u32 s65, s{lo+1}, 0")
    e(f"s_add_u32 m0, s61, {off}")
    e("s_nop 0")
    e(f"global_load_lds_dwordx4 %[voff], s[{'64:65' if i else f'{lo}:{lo+1}'}]")


def load_scales(e):
    """the scales of the stage s56:57 / s58:59 name: every lane reads the same 8 + 8 bytes (VMEM, so the stage's vmcnt(0) covers it)"""
    if MODE == "i32":
        return
    e(f"global_load_dwordx2 v[{VSCALE}:{VSCALE+1}], v{VZERO}, s[56:57]")
    e(f"global_load_dwordx2 v[{VSCALE+2}:{VSCALE+3}], v{VZERO}, s[58:59]")


def make_c(e):
    """c = f32(f32(sA * 1/49) * sB) (CloverVector4.h:1124-1127) of both K-blocks of the fetched stage: first -> s40, second -> v153"""
    if MODE == "i32":
        return
    e(f"v_mul_f32 v{VT}, 0x3ca72f05, v{VSCALE}")
    e(f"v_mul_f32 v{VT}, v{VSCALE+2}, v{VT}")
    e(f"v_mul_f32 v{VC1}, 0x3ca72f05, v{VSCALE+1}")
    e(f"v_mul_f32 v{VC1}, v{VSCALE+3}, v{VC1}")
    e(f"v_readfirstlane_b32 s40, v{VT}")


# tile visiting order: (fragment of A, fragment of B, accumulator tile 2a+b); K-block j0 then j1
UNITS = [("A0", "B0", 0), ("A0", "B1", 1), ("A1", "B1", 3), ("A1", "B0", 2),
         ("A0", "B1", 1), ("A0", "B0", 0), ("A1", "B0", 2), ("A1", "B1", 3)]
CREG = [40, 40, 40, 40, 42, 42, 42, 42]
Q0 = ["A0", "A0", "B0", "B0", "B1", "B1", "A1", "A1"]      # fragment loads in flight when a stage begins


def advance_pointers(e, limit):
    """image / scale pointers move on to the following stage only while it exists (s60 = stages left, this one included)"""
    for lo, step in ((52, STAGE_BYTES), (54, STAGE_BYTES), (56, 8), (58, 8)):
        if "dmafixed" in SKIP and lo in (52, 54):
            step = 0
        e(f"s_cmp_gt_u32 s60, {limit}")
        e(f"s_cselect_b32 s64, {step}, 0")
        e(f"s_add_u32 s{lo}, s{lo}, s64")
        e(f"s_addc_u32 s{lo+1}, s{lo+1}, 0")


def toggle(e):
    for op in ("a16", "a8", "b16", "b8"):
        e(f"v_add_u32 %[{op}], s62, %[{op}]")
    e("s_sub_u32 s61, s61, s62")
    e("s_sub_u32 s62, 0, s62")


def generate():
    e = Emit()
    # ---------------- prologue ----------------
    e("s_mov_b64 s[52:53], %[ga]")
    e("s_mov_b64 s[54:55], %[gb]")
    e("s_mov_b64 s[56:57], %[sa]")
    e("s_mov_b64 s[58:59], %[sb]")
    e("s_mov_b32 s60, %[np]")
    e("s_mov_b32 s61, %[dma]")
    e(f"s_mov_b32 s62, {BUF}")
    e("s_mov_b32 s63, %[cstride]")
    e("s_mov_b64 s[66:67], %[cb]")
    for r in (40, 41, 42, 43):
        e(f"s_mov_b32 s{r}, 0")
    e(f"v_mov_b32 v{VS}, 0x82828282")
    e(f"v_mov_b32 v{VZERO}, 0")
    for i in range(128):
        e(f"v_mov_b32 v{i}, 0")
    # stage 0 -> buffer 0 (s61 names it), its scales -> v156:159
    load_scales(e)
    for k in range(6):
        dma(e, k)
    advance_pointers(e, 1)                     # -> stage 1 if there is one
    e(f"s_add_u32 s61, s61, {BUF}")            # the DMA target is now buffer 1
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    make_c(e)
    e.ds_frag("A0", 0, 0)
    e.ds_frag("B0", 0, 0)
    e.ds_frag("B1", 0, 1)
    e.ds_frag("A1", 0, 1)
    assert e.lds_q == Q0
    # ---------------- one stage per iteration ----------------
    e("1:")
    e.in_loop = True
    for m in range(6):
        fa, fb, tile = UNITS[m]
        e.wait_frags(fa, fb)
        mfma(e, m, fa, fb, tile)
        if m == 0:
            load_scales(e)                     # next stage's scales
            dma(e, 0); dma(e, 1)
            fold(e, 6, UNITS[6][2], 42)        # previous stage's unit 6 (first stage: result sets and s42 are zero)
        elif m == 1:
            e.ds_frag("A0", 1, 0)
            dma(e, 2); dma(e, 3)
            fold(e, 7, UNITS[7][2], 42)        # previous stage's unit 7: the last reader of the old s42
            if MODE != "i32":
                e(f"v_readfirstlane_b32 s42, v{VC1}")
        elif m == 2:
            e.ds_frag("B1", 1, 1)
            dma(e, 4); dma(e, 5)
            fold(e, 0, UNITS[0][2], CREG[0])
        elif m == 3:
            e.ds_frag("B0", 1, 0)
            e.ds_frag("A1", 1, 1)
            fold(e, 1, UNITS[1][2], CREG[1])
        else:
            fold(e, m - 2, UNITS[m - 2][2], CREG[m - 2])
    # the next stage image has landed and every wave is done reading this one's buffer (all its fragment loads were issued by unit 3)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e.lds_q = []
    if "barrier" not in SKIP:
        e("s_barrier")
    advance_pointers(e, 2)
    toggle(e)                                  # base registers -> next buffer, DMA target -> this one
    e.ds_frag("A0", 0, 0)                      # next stage, first K-block (after the last stage: harmless reads of valid LDS)
    make_c(e)                                  # s40 <- c of its first K-block (s40's last reader was the fold behind unit 5)
    for m in (6, 7):
        fa, fb, tile = UNITS[m]
        mfma(e, m, fa, fb, tile)               # FA1, FB0, FB1 landed before the barrier
        if m == 6:
            e.ds_frag("B0", 0, 0)
        else:
            e.ds_frag("B1", 0, 1)
            e.ds_frag("A1", 0, 1)
        fold(e, m - 2, UNITS[m - 2][2], CREG[m - 2])
    assert e.lds_q == Q0 or "lds" in SKIP, e.lds_q
    e.in_loop = False
    e("s_sub_u32 s60, s60, 1")
    e("s_cmp_lg_u32 s60, 0")
    e("s_cbranch_scc1 1b")
    # ---------------- drain: the last two folds ----------------
    e("s_waitcnt lgkmcnt(0)")
    fold(e, 6, UNITS[6][2], 42)
    fold(e, 7, UNITS[7][2], 42)
    # ---------------- store C: 32x32 tile layout col = lane & 31, row = (t & 3) + 8 (t >> 2) + 4 (lane >> 5) ----------------
    if MODE == "i32":
        e("s_nop 15")                                             # the last MFMAs' results (12 states) before a VALU reads them
        for i in range(64):
            e(f"v_cvt_i32_f32 v{i}, v{i}")
    e("s_mul_i32 s65, s63, 5")
    first = True
    for a in range(2):
        for t in range(16):
            if not first:
                step = "s65" if (t & 3) == 0 else "s63"          # rows 3 -> 8, 11 -> 16, ..., 27 -> 32: five rows on
                e(f"s_add_u32 s66, s66, {step}")
                e("s_addc_u32 s67, s67, 0")
            first = False
            e(f"global_store_dword %[coff], v{16 * (2 * a + 0) + t}, s[66:67] nt")
            e(f"global_store_dword %[coff], v{16 * (2 * a + 1) + t}, s[66:67] offset:128 nt")
    return e.lines


VARIANTS = [("", set()), ("_NODMA", {"dma"}), ("_NOLDS", {"lds"}), ("_NODMA_NOLDS", {"dma", "lds"}), ("_NOBARRIER", {"barrier"}),
            ("_NOFOLD", {"fold"}), ("_ARITH", {"dma", "lds", "barrier"}), ("_DMAFIXED", {"dmafixed"}), ("_DMABONLY", {"dmaA"})]


def main():
    global SKIP, MODE
    out = sys.argv[1] if len(sys.argv) > 1 else "clover_amd/csrc/gemm6_loop.inc"
    vregs = list(range(0, 160))
    sregs = list(range(40, 68))
    experiments = len(sys.argv) > 3 and sys.argv[3] == "experiments"
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_gemm6_loop.py (NS = %d) -- do not edit; see that file for the schedule and the register map.\n" % NS)
        for suffix, skip in (VARIANTS if experiments else VARIANTS[:1]):
            SKIP = skip
            lines = generate()
            f.write("#define G6_LOOP_ASM%s \\\n" % suffix)
            for ln in lines:
                f.write('    "%s\\n" \\\n' % ln)
            f.write('    ""\n')
            print(f"{out}: G6_LOOP_ASM{suffix}: {len(lines)} instructions")
        SKIP, MODE = set(), "i32"
        lines = generate()
        f.write("#define G6_LOOP_ASM_I32 \\\n")
        for ln in lines:
            f.write('    "%s\\n" \\\n' % ln)
        f.write('    ""\n')
        print(f"{out}: G6_LOOP_ASM_I32: {len(lines)} instructions")
        MODE = "scaled"
        if experiments:
            f.write("#define G6_LOOP_EXPERIMENTS 1\n")
        f.write("#define G6_LOOP_CLOBBERS " + ", ".join(f'"v{i}"' for i in vregs) + ", " + ", ".join(f'"s{i}"' for i in sregs) + ', "scc", "memory"\n')


if __name__ == "__main__":
    main()
