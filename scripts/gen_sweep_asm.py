#!/usr/bin/env python3
"""Generator of the hand-written block-run loop of the K = 16, R = 2 float32 sweeps (phlash_amd/csrc/sweep_run_k16r2.inc).

Why (profiles/r06_ab_experiments.txt items 0, 6): a wave of the sweeps pays an issue slot for every instruction, scalar tests and
not-taken branches included.  The C++ block body needs sixteen test-and-branch pairs per 8-site block (is site i hom in every
lane?) although 72-82 % of the blocks at 1 % hets are hom throughout, and every attempt to give such blocks a copy of the body
without the tests -- as a second C++ instance (rounds 4, 5), as grouped tests, as an inline-asm block beside the C++ block
(round 6) -- lost to the register allocator's copies at the join of the two bodies.  This file therefore owns the WHOLE run of
hot blocks: the loop, the prefetch of checkpoints / block exponents / observation words, the block's masks, an all-hom body
without tests and a mixed body with them, under ONE register plan, as one asm statement whose operands are the run's state.

The arithmetic is the C++ body's, operation for operation (psmc_kernels.hip, bwd_kernel, PHK_HET_REGS body; Lane::scans,
scans_adj, suffix_vw, beta_prev, carry with the SPLIT layout at R = 2): every fma / mul / add of the C++ source is one
instruction here with the same operands in the same order, so the results are bit-identical
(tests/test_hip_parity.py::test_asm_block_run_equals_the_cxx_body).

Register plan (physical VGPRs are clobbers of the statement; the state's registers are its operands, chosen by the compiler):
    v[TBASE ...]   R[j][h], j = 0..7: the block's w vectors / the beta chain (w_j lives in R[j+1], the incoming beta -- an
                   operand -- is R[8]);  forward-pass temporaries: A / T (state, ping-pong), the three scan chains X (prefix of
                   u.*a), Y (suffix of a), Z (suffix of v.*w), lane totals, carries;  the prefetched checkpoint (two quads),
                   codes, scale, addresses.
Hazards are kept by construction: `emit()` records the registers every instruction defines and uses, and `finish()` inserts an
s_nop wherever (a) a packed-float32 instruction's result would be read by the very next instruction, or a packed instruction
would read a result of the very next-to-last one (one wait state, r05 item 19), (b) a DPP or readlane operand was written less
than three instructions earlier (two wait states).
"""
import sys

TBASE = 118          # first clobbered VGPR
NP = 4               # packed pairs per lane (8 states)
T = 8                # sites per block


class Gen:
    def __init__(self):
        self.ins = []       # (text, defs, uses, kind)
        self.next_free = TBASE
        self.labels = 0

    # ---- register helpers ------------------------------------------------------------------------------------------------
    def pair(self):
        if self.next_free % 2:
            self.next_free += 1
        r = self.next_free
        self.next_free += 2
        assert self.next_free <= 256, "out of temporaries"
        return ("p", r)

    def single(self):
        r = self.next_free
        self.next_free += 1
        assert self.next_free <= 256
        return ("s", r)

    def quad(self):
        while self.next_free % 4:
            self.next_free += 1
        r = self.next_free
        self.next_free += 4
        assert self.next_free <= 256
        return ("q", r)

    @staticmethod
    def txt(r):
        """assembly text of a register object: physical pair / single / quad, or an operand placeholder"""
        if isinstance(r, str):
            return r
        k, n = r[0], r[1]
        if k == "p":
            return f"v[{n}:{n + 1}]"
        if k == "q":
            return f"v[{n}:{n + 3}]"
        if k == "s":
            return f"v{n}"
        raise ValueError(r)

    @staticmethod
    def regs(r):
        """set of register ids (ints for physical VGPRs, strings for operands) an object covers"""
        if isinstance(r, str):
            return {r}
        k, n = r[0], r[1]
        return set(range(n, n + {"p": 2, "q": 4, "s": 1}[k]))

    @staticmethod
    def lo(p):
        assert p[0] == "p"
        return ("s", p[1])

    @staticmethod
    def hi(p):
        assert p[0] == "p"
        return ("s", p[1] + 1)

    def emit(self, text, defs=(), uses=(), kind="valu"):
        d, u = set(), set()
        for r in defs:
            d |= self.regs(r)
        for r in uses:
            u |= self.regs(r)
        self.ins.append([text, d, u, kind])

    def label(self, name):
        self.ins.append([name + ":", set(), set(), "label"])

    def newlabel(self, stem):
        self.labels += 1
        return f".Lphk_{stem}_{self.labels}_%="

    # ---- instruction helpers (packed float32 unless said otherwise) ------------------------------------------------------
    def pk_fma(self, d, a, b, c):
        ct = "0" if c is None else self.txt(c)
        mod = " op_sel_hi:[1,1,0]" if c is None else ""
        self.emit(f"v_pk_fma_f32 {self.txt(d)}, {self.txt(a)}, {self.txt(b)}, {ct}{mod}", [d], [a, b] + ([] if c is None else [c]), "pk")

    def pk_mul(self, d, a, b):
        self.emit(f"v_pk_mul_f32 {self.txt(d)}, {self.txt(a)}, {self.txt(b)}", [d], [a, b], "pk")

    def pk_add(self, d, a, b):
        if b is None:  # x + 0 (the C++ source adds the carry to a zero-initialised scan entry)
            self.emit(f"v_pk_add_f32 {self.txt(d)}, {self.txt(a)}, 0 op_sel_hi:[1,0]", [d], [a], "pk")
        else:
            self.emit(f"v_pk_add_f32 {self.txt(d)}, {self.txt(a)}, {self.txt(b)}", [d], [a, b], "pk")

    def lane_total(self, d, x):
        """(x.lo + x.hi, x.hi + x.lo)"""
        self.emit(f"v_pk_add_f32 {self.txt(d)}, {self.txt(x)}, {self.txt(x)} op_sel:[0,1] op_sel_hi:[1,0]", [d], [x], "pk")

    def carry(self, dst, lt, tot, mask, prefix):
        """Lane::carry at R = 2.  prefix: c = row_shr:1(lane total) * up1, r = (c, c + tot.lo);  suffix: c = row_shl:1(...) * dn1,
        r = (c + tot.hi, c).  lt: pair holding the lane total in its low half."""
        if prefix:
            self.emit(f"v_mul_f32_dpp {self.txt(self.lo(dst))}, {self.txt(self.lo(lt))}, {self.txt(mask)} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1",
                      [self.lo(dst)], [self.lo(lt), mask], "dpp")
            self.emit(f"v_add_f32_e32 {self.txt(self.hi(dst))}, {self.txt(self.lo(tot))}, {self.txt(self.lo(dst))}", [self.hi(dst)], [self.lo(tot), self.lo(dst)])
        else:
            self.emit(f"v_mul_f32_dpp {self.txt(self.hi(dst))}, {self.txt(self.lo(lt))}, {self.txt(mask)} row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1",
                      [self.hi(dst)], [self.lo(lt), mask], "dpp")
            self.emit(f"v_add_f32_e32 {self.txt(self.lo(dst))}, {self.txt(self.hi(tot))}, {self.txt(self.hi(dst))}", [self.lo(dst)], [self.hi(tot), self.hi(dst)])

    def mov64(self, d, s):
        self.emit(f"v_mov_b64_e32 {self.txt(d)}, {self.txt(s)}", [d], [s])

    # ---- finishing: hazards ----------------------------------------------------------------------------------------------
    def finish(self):
        out = []   # final instruction list
        hist = []  # (defs, kind) of the last real instructions, most recent last ("nop" entries have empty defs)
        nops = 0
        for text, d, u, kind in self.ins:
            if kind == "label":
                out.append(text)
                hist = []  # a join: be conservative, treat as fresh (the predecessor's last instructions are unknown)
                # conservative: after a label require the generic gaps again by assuming a packed write of everything -> handled by
                # emitting an s_nop 1 at labels that are branch targets (cheap: labels are block / run boundaries or cold paths)
                continue
            need = 0
            if kind in ("valu", "pk", "dpp", "lane", "vmem_addr", "lds"):
                for back, (pd, pk) in enumerate(reversed(hist[-3:])):  # back = 0: previous instruction
                    if not (pd & u):
                        continue
                    gap_have = back
                    if kind in ("dpp", "lane"):
                        need = max(need, 2 - gap_have)
                    elif pk == "pk" or kind == "pk":
                        need = max(need, 1 - gap_have)
            if need > 0:
                out.append(f"s_nop {need - 1}")
                nops += 1
                for _ in range(need):
                    hist.append((set(), "nop"))
            out.append(text)
            if kind in ("branch",):
                hist = []
            else:
                hist.append((d if kind in ("valu", "pk", "dpp", "lane") else set(), kind))
        return out, nops


def build():
    g = Gen()
    # ---- operands of the asm statement (placeholders) -------------------------------------------------------------------
    B = [f"%[b{h}]" for h in range(NP)]          # beta (in / out)
    GB = [f"%[gb{h}]" for h in range(NP)]
    GD = [f"%[gd{h}]" for h in range(NP)]
    GU = [f"%[gu{h}]" for h in range(NP)]
    GV = [f"%[gv{h}]" for h in range(NP)]
    G1 = [f"%[g1{h}]" for h in range(NP)]
    Pb = [f"%[pb{h}]" for h in range(NP)]
    Pd = [f"%[pd{h}]" for h in range(NP)]
    Pu = [f"%[pu{h}]" for h in range(NP)]
    Pv = [f"%[pv{h}]" for h in range(NP)]
    RH = [f"%[rh{h}]" for h in range(NP)]
    UP1, DN1 = "%[up1]", "%[dn1]"
    AN = [f"%[an{i}]" for i in range(8)]         # prefetched checkpoint (floats, in / out)
    WCUR, WPREV, ENEXT = "%[wcur]", "%[wprev]", "%[enext]"
    CKQ, EBQ, WORDS, LDS = "%[ckq]", "%[ebq]", "%[words]", "%[lds]"
    BLK, WIDX, STOP, BLKLO = "%[blk]", "%[widx]", "%[stop]", "%[blklo]"
    CKSTEP, EBSTEP, CKPIECE = "%[ckstep]", "%[ebstep]", "%[ckpiece]"  # 64-bit SGPR pairs: -ck_step bytes, -eb step bytes, +piece distance bytes

    # ---- temporaries ----------------------------------------------------------------------------------------------------
    R = [[g.pair() for h in range(NP)] for j in range(T)]   # R[j]: beta after site j's step = w of site j - 1 ... (see below)
    A = [g.pair() for h in range(NP)]
    Tt = [g.pair() for h in range(NP)]
    # the scan chains as quads (so that the cold missing-site bodies can land their 16-byte LDS rows in them: the chains are dead there)
    XQ, YQ, ZQ = [g.quad(), g.quad()], [g.quad(), g.quad()], [g.quad(), g.quad()]
    XC = [("p", XQ[h // 2][1] + 2 * (h % 2)) for h in range(NP)]
    YC = [("p", YQ[h // 2][1] + 2 * (h % 2)) for h in range(NP)]
    ZC = [("p", ZQ[h // 2][1] + 2 * (h % 2)) for h in range(NP)]
    LTX, LTY, LTZ = g.pair(), g.pair(), g.pair()
    CX, CY, CZ = g.pair(), g.pair(), g.pair()
    PQ0, PQ1 = g.quad(), g.quad()                            # checkpoint pieces: states 0..3 / 4..7 of the lane
    CKQ2 = g.pair()
    ADDR = g.pair()
    FXP = g.pair()                                            # the block's scale 2^-e in the low half (read as a pair by v_pk_mul)
    FX = g.lo(FXP)
    CODES, TMP, C16, CC, LADDR = g.single(), g.single(), g.single(), g.single(), g.single()
    ER0, ER1, GR0, GR1 = XQ[0], XQ[1], YQ[0], YQ[1]            # missing-site path: emission row and mass row halves (dead chains)
    S_T0, S_W, S_SH, S_A, S_B, S_U, S_NH, S_NM, S_TMP = "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96"
    S_SAVE = "s[98:99]"
    S_OFF = "s[86:87]"

    W = lambda i: (B if i == T - 1 else R[i + 1])  # w of site i lives where beta stood when the site was entered
    # beta after site i's step goes to R[i]; block output R[0] -> copied to B at the block's end

    def quad_lane(q, i):
        return ("s", q[1] + i)

    # ---- prologue of the run: checkpoint prefetch registers, second piece pointer -----------------------------------------
    for i in range(4):
        g.emit(f"v_mov_b32_e32 {g.txt(quad_lane(PQ0, i))}, {AN[i]}", [quad_lane(PQ0, i)], [AN[i]])
        g.emit(f"v_mov_b32_e32 {g.txt(quad_lane(PQ1, i))}, {AN[4 + i]}", [quad_lane(PQ1, i)], [AN[4 + i]])
    g.emit(f"v_lshl_add_u64 {g.txt(CKQ2)}, {CKQ}, 0, {CKPIECE}", [CKQ2], [CKQ], "valu")

    top = g.newlabel("top")
    g.label(top)
    # ---- enter(): words, checkpoint, exponent ---------------------------------------------------------------------------
    g.emit("s_waitcnt vmcnt(0)", kind="wait")
    g.emit(f"s_lshl_b32 {S_T0}, {BLK}, 3", kind="salu")
    g.emit(f"s_lshr_b32 {S_W}, {S_T0}, 4", kind="salu")
    g.emit(f"s_cmp_lg_u32 {S_W}, {WIDX}", kind="salu")
    same = g.newlabel("sameword")
    g.emit(f"s_cbranch_scc0 {same}", kind="branch")
    g.emit(f"s_mov_b32 {WIDX}, {S_W}", kind="salu")
    g.emit(f"v_mov_b32_e32 {WCUR}, {WPREV}", [WCUR], [WPREV])
    g.emit(f"s_sub_i32 {S_TMP}, {S_W}, 1", kind="salu")
    g.emit(f"s_max_i32 {S_TMP}, {S_TMP}, 0", kind="salu")
    g.emit(f"s_lshl_b32 s86, {S_TMP}, 2", kind="salu")
    g.emit("s_mov_b32 s87, 0", kind="salu")
    g.emit(f"v_lshl_add_u64 {g.txt(ADDR)}, {WORDS}, 0, {S_OFF}", [ADDR], [WORDS])
    g.emit("s_nop 0", kind="salu")
    g.emit(f"global_load_dword {WPREV}, {g.txt(ADDR)}, off", [WPREV], [ADDR], "vmem")
    g.label(same)
    g.emit("s_nop 1", kind="salu")
    # codes = wcur >> (2 * (t0 & 15)); t0 & 15 is 0 or 8
    g.emit(f"s_and_b32 {S_SH}, {S_T0}, 8", kind="salu")
    g.emit(f"s_lshl_b32 {S_SH}, {S_SH}, 1", kind="salu")
    g.emit(f"v_lshrrev_b32_e32 {g.txt(CODES)}, {S_SH}, {WCUR}", [CODES], [WCUR])
    # al0: state i of the lane -> pair i % 4, half i / 4
    for h in range(NP):
        g.emit(f"v_mov_b32_e32 {g.txt(g.lo(A[h]))}, {g.txt(quad_lane(PQ0, h))}", [g.lo(A[h])], [quad_lane(PQ0, h)])
        g.emit(f"v_mov_b32_e32 {g.txt(g.hi(A[h]))}, {g.txt(quad_lane(PQ1, h))}", [g.hi(A[h])], [quad_lane(PQ1, h)])
    # beta *= 2^-e_fwd (e_fwd = e_next as it stands)
    g.emit(f"v_sub_u32_e32 {g.txt(TMP)}, 0, {ENEXT}", [TMP], [ENEXT])
    g.emit(f"v_ldexp_f32 {g.txt(FX)}, 1.0, {g.txt(TMP)}", [FX], [TMP])
    # prefetch the previous block's checkpoint and exponent (blk > blk_lo), step the pointers
    nopf = g.newlabel("nopf")
    g.emit(f"s_cmp_gt_i32 {BLK}, {BLKLO}", kind="salu")
    g.emit(f"s_cbranch_scc0 {nopf}", kind="branch")
    g.emit(f"global_load_dwordx4 {g.txt(PQ0)}, {CKQ}, off", [PQ0], [CKQ], "vmem")
    g.emit(f"global_load_dwordx4 {g.txt(PQ1)}, {g.txt(CKQ2)}, off", [PQ1], [CKQ2], "vmem")
    g.emit(f"global_load_sshort {ENEXT}, {EBQ}, off", [ENEXT], [EBQ], "vmem")
    g.emit(f"v_lshl_add_u64 {CKQ}, {CKQ}, 0, {CKSTEP}", [CKQ], [CKQ])
    g.emit(f"v_lshl_add_u64 {g.txt(CKQ2)}, {g.txt(CKQ2)}, 0, {CKSTEP}", [CKQ2], [CKQ2])
    g.emit(f"v_lshl_add_u64 {EBQ}, {EBQ}, 0, {EBSTEP}", [EBQ], [EBQ])
    g.label(nopf)
    g.emit("s_nop 1", kind="salu")
    # scale: beta[h] *= fx (both halves from the one register: op_sel_hi[0] = 0)
    for h in range(NP):
        g.emit(f"v_pk_mul_f32 {B[h]}, {g.txt(FXP)}, {B[h]} op_sel_hi:[0,1]", [B[h]], [FX, B[h]], "pk")
    # masks: OR over the wave of the block's code bits = lane 0's | lane 63's (the wave holds at most two rows)
    g.emit(f"v_and_b32_e32 {g.txt(C16)}, 0xffff, {g.txt(CODES)}", [C16], [CODES])
    g.emit("s_nop 1", kind="salu")
    g.emit(f"v_readfirstlane_b32 {S_A}, {g.txt(C16)}", [], [C16], "lane")
    g.emit(f"v_readlane_b32 {S_B}, {g.txt(C16)}, 63", [], [C16], "lane")
    g.emit("s_nop 0", kind="salu")
    g.emit(f"s_or_b32 {S_U}, {S_A}, {S_B}", kind="salu")
    g.emit(f"s_lshr_b32 {S_TMP}, {S_U}, 1", kind="salu")
    g.emit(f"s_and_b32 {S_NM}, {S_TMP}, 0x5555", kind="salu")
    g.emit(f"s_or_b32 {S_NH}, {S_U}, {S_TMP}", kind="salu")
    g.emit(f"s_and_b32 {S_NH}, {S_NH}, 0x5555", kind="salu")
    g.emit(f"s_cmp_eq_u32 {S_NH}, 0", kind="salu")
    mixed = g.newlabel("mixed")
    nxt = g.newlabel("next")
    g.emit(f"s_cbranch_scc0 {mixed}", kind="branch")

    cold = []  # out-of-line bodies of the mixed path: (label, emitter function)

    def beta_site(i, tests):
        Wi = W(i)
        Bn = R[i]
        if tests:
            lab = g.newlabel(f"bh{i}")
            ret = g.newlabel(f"br{i}")
            g.emit(f"s_bitcmp1_b32 {S_NH}, {2 * i}", kind="salu")
            g.emit(f"s_cbranch_scc1 {lab}", kind="branch")
            g.label(ret)
            g.emit("s_nop 0", kind="salu")
            cold.append(("beta", i, lab, ret))
        # suffix of v.*w (Z chain) and prefix of b.*w (X chain), interleaved
        g.pk_fma(ZC[0], Pv[3], Wi[3], None)
        g.pk_fma(XC[0], Pb[0], Wi[0], None)
        g.pk_fma(ZC[1], Pv[2], Wi[2], ZC[0])
        g.pk_fma(XC[1], Pb[1], Wi[1], XC[0])
        g.pk_fma(ZC[2], Pv[1], Wi[1], ZC[1])
        g.pk_fma(XC[2], Pb[2], Wi[2], XC[1])
        g.pk_fma(ZC[3], Pv[0], Wi[0], ZC[2])   # tv
        g.pk_fma(XC[3], Pb[3], Wi[3], XC[2])   # tb
        g.lane_total(LTZ, ZC[3])
        g.lane_total(LTX, XC[3])
        # nb_h = fma(d_h, w_h, pbw_h): pbw = (0, X0, X1, X2)  (independent work between the totals and the DPP reads)
        g.pk_fma(YC[0], Pd[0], Wi[0], None)
        g.pk_fma(YC[1], Pd[1], Wi[1], XC[0])
        g.pk_fma(YC[2], Pd[2], Wi[2], XC[1])
        g.pk_fma(YC[3], Pd[3], Wi[3], XC[2])
        g.carry(CZ, LTZ, ZC[3], DN1, prefix=False)   # cv
        g.carry(CX, LTX, XC[3], UP1, prefix=True)    # cb
        # svw_h += cv: svw = (Z2, Z1, Z0, 0)
        g.pk_add(ZC[3], CZ, None)        # svw3 = 0 + cv   (the C++ source: splat(0) + cv)
        g.pk_add(ZC[0], ZC[0], CZ)       # svw2
        g.pk_add(ZC[1], ZC[1], CZ)       # svw1
        g.pk_add(ZC[2], ZC[2], CZ)       # svw0
        # nb += cb ; beta = fma(u, svw, nb)
        for h in range(NP):
            g.pk_add(YC[h], YC[h], CX)
        svw = [ZC[2], ZC[1], ZC[0], ZC[3]]
        for h in range(NP):
            g.pk_fma(Bn[h], Pu[h], svw[h], YC[h])

    def fwd_site(i, tests, a, t):
        """a: registers of the state entering the site; t: where p = A' a goes (the next site's a)"""
        Wi = W(i)
        last = i == T - 1
        g.pk_fma(XC[0], Pu[0], a[0], None)
        g.pk_add(YC[0], a[3], None)            # 0 + a3
        g.pk_fma(ZC[0], Pv[3], Wi[3], None)
        g.pk_fma(XC[1], Pu[1], a[1], XC[0])
        g.pk_add(YC[1], YC[0], a[2])
        g.pk_fma(ZC[1], Pv[2], Wi[2], ZC[0])
        g.pk_fma(XC[2], Pu[2], a[2], XC[1])
        g.pk_add(YC[2], YC[1], a[1])
        g.pk_fma(ZC[2], Pv[1], Wi[1], ZC[1])
        g.pk_fma(XC[3], Pu[3], a[3], XC[2])    # tu
        g.pk_add(YC[3], YC[2], a[0])           # ta
        g.pk_fma(ZC[3], Pv[0], Wi[0], ZC[2])   # tv
        g.lane_total(LTX, XC[3])
        g.lane_total(LTY, YC[3])
        g.lane_total(LTZ, ZC[3])
        # gd += w .* a (needs nothing of the scans: fills the gap before the DPP reads)
        for h in range(NP):
            g.pk_fma(GD[h], Wi[h], a[h], GD[h])
        g.carry(CX, LTX, XC[3], UP1, prefix=True)     # cu
        g.carry(CY, LTY, YC[3], DN1, prefix=False)    # ca
        g.carry(CZ, LTZ, ZC[3], DN1, prefix=False)    # cv
        # pre = (0, X0, X1, X2) + cu ; suf = (Y2, Y1, Y0, 0) + ca ; svw = (Z2, Z1, Z0, 0) + cv
        g.pk_add(XC[3], CX, None)
        g.pk_add(YC[3], CY, None)
        g.pk_add(ZC[3], CZ, None)
        for k in range(3):
            g.pk_add(XC[k], XC[k], CX)
            g.pk_add(YC[k], YC[k], CY)
            g.pk_add(ZC[k], ZC[k], CZ)
        pre = [XC[3], XC[0], XC[1], XC[2]]
        suf = [YC[2], YC[1], YC[0], YC[3]]
        svw = [ZC[2], ZC[1], ZC[0], ZC[3]]
        for h in range(NP):
            g.pk_fma(GB[h], Wi[h], suf[h], GB[h])
        for h in range(NP):
            g.pk_fma(GV[h], Wi[h], pre[h], GV[h])
        for h in range(NP):
            g.pk_fma(GU[h], a[h], svw[h], GU[h])
        if last and not tests:
            return
        for h in range(NP):
            g.pk_mul(t[h], Pd[h], a[h])
        for h in range(NP):
            g.pk_fma(t[h], Pv[h], pre[h], t[h])
        for h in range(NP):
            g.pk_fma(t[h], Pb[h], suf[h], t[h])
        if tests:
            lab = g.newlabel(f"fh{i}")
            ret = g.newlabel(f"fr{i}")
            g.emit(f"s_bitcmp1_b32 {S_NH}, {2 * i}", kind="salu")
            g.emit(f"s_cbranch_scc1 {lab}", kind="branch")
            g.label(ret)
            g.emit("s_nop 0", kind="salu")
            cold.append(("fwd", i, lab, ret, t))

    def block(tests):
        for i in range(T - 1, -1, -1):
            beta_site(i, tests)
        a, t = A, Tt
        for i in range(T):
            fwd_site(i, tests, a, t)
            a, t = t, a
        # the block's result: beta at its left edge
        for h in range(NP):
            g.mov64(B[h], R[0][h])

    # ---- all-hom block ---------------------------------------------------------------------------------------------------
    block(False)
    g.emit(f"s_branch {nxt}", kind="branch")
    # ---- mixed block -----------------------------------------------------------------------------------------------------
    g.label(mixed)
    g.emit("s_nop 1", kind="salu")
    block(True)
    g.label(nxt)
    g.emit(f"s_sub_i32 {BLK}, {BLK}, 1", kind="salu")
    g.emit(f"s_cmp_ge_i32 {BLK}, {STOP}", kind="salu")
    g.emit(f"s_cbranch_scc1 {top}", kind="branch")
    done = g.newlabel("done")
    g.emit(f"s_branch {done}", kind="branch")

    # ---- cold bodies of the mixed block ----------------------------------------------------------------------------------
    for c in cold:
        if c[0] == "beta":
            _, i, lab, ret = c
            Wi = W(i)
            g.label(lab)
            g.emit("s_nop 1", kind="salu")
            g.emit(f"v_bfe_u32 {g.txt(CC)}, {g.txt(CODES)}, {2 * i}, 2", [CC], [CODES])
            g.emit("s_nop 0", kind="salu")
            g.emit(f"v_cmp_eq_u32_e32 vcc, 1, {g.txt(CC)}", [], [CC])
            g.emit(f"s_and_saveexec_b64 {S_SAVE}, vcc", kind="salu")
            for h in range(NP):
                g.pk_mul(Wi[h], Wi[h], RH[h])
            g.emit(f"s_mov_b64 exec, {S_SAVE}", kind="salu")
            g.emit(f"s_bitcmp1_b32 {S_NM}, {2 * i}", kind="salu")
            g.emit(f"s_cbranch_scc0 {ret}", kind="branch")
            # a missing site: w .*= row (1 / emis0 where this lane's site is missing, row 0 = ones elsewhere)
            g.emit(f"v_cmp_eq_u32_e32 vcc, 2, {g.txt(CC)}", [], [CC])
            g.emit(f"v_cndmask_b32_e64 {g.txt(LADDR)}, 0, 64, vcc", [LADDR], [])
            g.emit(f"v_add_u32_e32 {g.txt(LADDR)}, {LDS}, {g.txt(LADDR)}", [LADDR], [LADDR, LDS])
            g.emit("s_nop 0", kind="salu")
            g.emit(f"ds_read_b128 {g.txt(ER0)}, {g.txt(LADDR)}", [ER0], [LADDR], "lds")
            g.emit(f"ds_read_b128 {g.txt(ER1)}, {g.txt(LADDR)} offset:16", [ER1], [LADDR], "lds")
            g.emit("s_waitcnt lgkmcnt(0)", kind="wait")
            for h in range(NP):
                q = ER0 if h < 2 else ER1
                e = ("p", q[1] + 2 * (h % 2))
                g.pk_mul(Wi[h], Wi[h], e)
            g.emit(f"s_branch {ret}", kind="branch")
        else:
            _, i, lab, ret, t = c
            Wi = W(i)
            g.label(lab)
            g.emit("s_nop 1", kind="salu")
            g.emit(f"v_bfe_u32 {g.txt(CC)}, {g.txt(CODES)}, {2 * i}, 2", [CC], [CODES])
            g.emit("s_nop 0", kind="salu")
            g.emit(f"v_cmp_eq_u32_e32 vcc, 1, {g.txt(CC)}", [], [CC])
            g.emit(f"s_and_saveexec_b64 {S_SAVE}, vcc", kind="salu")
            for h in range(NP):
                g.pk_fma(G1[h], t[h], Wi[h], G1[h])
            for h in range(NP):
                g.pk_mul(t[h], t[h], RH[h])
            g.emit(f"s_mov_b64 exec, {S_SAVE}", kind="salu")
            g.emit(f"s_bitcmp1_b32 {S_NM}, {2 * i}", kind="salu")
            g.emit(f"s_cbranch_scc0 {ret}", kind="branch")
            # a missing site: mass p .* w into the LDS row of the lane's code (2, or 0 where this lane is not missing: nobody
            # reads row 0), then p .*= the lane's emission row
            g.emit(f"v_cmp_eq_u32_e32 vcc, 2, {g.txt(CC)}", [], [CC])
            g.emit(f"v_cndmask_b32_e64 {g.txt(LADDR)}, 0, 64, vcc", [LADDR], [])
            g.emit(f"v_add_u32_e32 {g.txt(LADDR)}, {LDS}, {g.txt(LADDR)}", [LADDR], [LADDR, LDS])
            g.emit("s_nop 0", kind="salu")
            g.emit(f"ds_read_b128 {g.txt(GR0)}, {g.txt(LADDR)} offset:112", [GR0], [LADDR], "lds")
            g.emit(f"ds_read_b128 {g.txt(GR1)}, {g.txt(LADDR)} offset:128", [GR1], [LADDR], "lds")
            g.emit(f"ds_read_b128 {g.txt(ER0)}, {g.txt(LADDR)}", [ER0], [LADDR], "lds")
            g.emit(f"ds_read_b128 {g.txt(ER1)}, {g.txt(LADDR)} offset:16", [ER1], [LADDR], "lds")
            g.emit("s_waitcnt lgkmcnt(0)", kind="wait")
            for h in range(NP):
                q = GR0 if h < 2 else GR1
                gr = ("p", q[1] + 2 * (h % 2))
                g.pk_fma(gr, t[h], Wi[h], gr)
            for h in range(NP):
                q = ER0 if h < 2 else ER1
                e = ("p", q[1] + 2 * (h % 2))
                g.pk_mul(t[h], t[h], e)
            g.emit("s_nop 0", kind="salu")
            g.emit(f"ds_write_b128 {g.txt(LADDR)}, {g.txt(GR0)} offset:112", [], [LADDR, GR0], "lds")
            g.emit(f"ds_write_b128 {g.txt(LADDR)}, {g.txt(GR1)} offset:128", [], [LADDR, GR1], "lds")
            g.emit(f"s_branch {ret}", kind="branch")

    g.label(done)
    g.emit("s_waitcnt vmcnt(0) lgkmcnt(0)", kind="wait")
    for i in range(4):
        g.emit(f"v_mov_b32_e32 {AN[i]}, {g.txt(quad_lane(PQ0, i))}", [AN[i]], [quad_lane(PQ0, i)])
        g.emit(f"v_mov_b32_e32 {AN[4 + i]}, {g.txt(quad_lane(PQ1, i))}", [AN[4 + i]], [quad_lane(PQ1, i)])
    lines, nops = g.finish()
    return lines, nops, g.next_free


def main():
    lines, nops, top = build()
    out = sys.argv[1] if len(sys.argv) > 1 else "phlash_amd/csrc/sweep_run_k16r2.inc"
    ops_io = []
    for nm, var in (("b", "beta"), ("gb", "gb"), ("gd", "gd"), ("gu", "gu"), ("gv", "gv"), ("g1", "g1")):
        ops_io += [f'[{nm}{h}] "+v"({var}[{h}])' for h in range(NP)]
    ops_io += [f'[an{i}] "+v"(anext[{i}])' for i in range(8)]
    ops_io += ['[wcur] "+v"(wcur)', '[wprev] "+v"(wprev)', '[enext] "+v"(e_next)', '[ckq] "+v"(asm_ckq)', '[ebq] "+v"(asm_ebq)',
               '[blk] "+s"(asm_blk)', '[widx] "+s"(asm_widx)']
    ops_in = []
    for nm, var in (("pb", "lane.b"), ("pd", "lane.d"), ("pu", "lane.u"), ("pv", "lane.v"), ("rh", "rhet")):
        ops_in += [f'[{nm}{h}] "v"({var}[{h}])' for h in range(NP)]
    ops_in += ['[up1] "v"(lane.g.up1)', '[dn1] "v"(lane.g.dn1)', '[words] "v"(asm_words)', '[lds] "v"(asm_lds)', '[stop] "s"(asm_stop)',
               '[blklo] "s"(asm_blklo)', '[ckstep] "s"(asm_ckstep)', '[ebstep] "s"(asm_ebstep)', '[ckpiece] "s"(asm_ckpiece)']
    clob = [f'"v{i}"' for i in range(TBASE, 256)] + [f'"s{i}"' for i in range(86, 100)] + ['"vcc"', '"scc"', '"memory"']
    with open(out, "w") as f:
        f.write("// GENERATED by scripts/gen_sweep_asm.py -- do not edit.  The run of hot blocks of bwd_kernel<float, 16, 2, 8, 4, *> as one asm\n"
                f"// statement ({len(lines)} lines, {nops} s_nop inserted by the generator's hazard pass, temporaries v{TBASE}..v{top - 1}).\n")
        f.write("asm volatile(\n")
        for ln in lines:
            f.write(f'    "{ln}\\n\\t"\n')
        f.write("    : " + ", ".join(ops_io) + "\n    : " + ", ".join(ops_in) + "\n    : " + ", ".join(clob) + ");\n")
    print(f"wrote {out}: {len(lines)} lines, {nops} s_nop, temporaries up to v{top - 1}")


if __name__ == "__main__":
    main()
