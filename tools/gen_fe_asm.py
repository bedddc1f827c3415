#!/usr/bin/env python3
"""Generates rust-kzg-bn254_amd/csrc/fe_asm.h: the Montgomery products of field29.h (fe_mul2, fe_sqr2, fe_mulsub) as ONE inline-asm
statement each for gfx950.

Why: hipcc pads every asm statement with an s_nop, and the C++ forms need an empty asm per multiply-add to keep
LLVM from reassociating the column sums (fe_mac_b) and from proving limbs non-negative (fe_opaque): 245 s_nop + ~50 moves in the
2 450 instructions of one mixed addition.  One statement per product pair = one pad, the column carry is the addend of the next
v_mad_i64_i32, and nothing is left for the compiler to rearrange.  The arithmetic is the same exact integer arithmetic, column by
column, as the C++ forms (which stay as the host / bound-check build and as the definition): results are bit-identical.

Register plan of one statement: inputs "v" (9 limbs each), outputs "=&v" 9 limbs per product which ALSO hold the Montgomery
quotient digits m_0..m_8 during the first nine columns (m_j dies after column j + 8, out_(k-9) is born in column k), the two 64-bit
column accumulators in FIXED registers v[0:1], v[2:3] (listed as clobbers: the low word is needed on its own, and inline asm has
no sub-register modifier), the modulus limbs and -m^-1 mod 2^29 in SGPRs, vcc as the unused carry-out.
"""
import os

NL = 9
MASK = "0x1fffffff"


class Emit:
    def __init__(self):
        self.lines = []
        self.ops = []           # (constraint, c-expression)
        self.idx = {}

    def op(self, key, constraint, expr):
        self.idx[key] = len(self.ops)
        self.ops.append((constraint, expr))

    def r(self, key):
        return "%%%d" % self.idx[key]

    def ins(self, s):
        self.lines.append(s)


def gen_product_columns(e, prods, acc_regs):
    """prods: list of dicts {terms: fn(k) -> list of (x_key, y_key) products of column k, out: key prefix}.  Interleaves the products
    statement by statement like field29.h fe_mul2."""
    n = len(prods)
    started = [False] * n           # accumulator holds a value (else the first mad of the product takes the constant 0)

    def mad(p, x, y, sgpr_y=False):
        acc = acc_regs[p]
        src2 = acc if started[p] else "0"
        e.ins("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (acc, x, y, src2))
        started[p] = True

    for k in range(2 * NL - 1):
        term_lists = [pr["terms"](k) for pr in prods]
        # data products, interleaved across the products of the statement
        for t in range(max(len(tl) for tl in term_lists)):
            for p in range(n):
                if t < len(term_lists[p]):
                    x, y = term_lists[p][t]
                    mad(p, e.r(x), e.r(y))
        # reduction products m_j * P[k - j]
        lo_j = 0 if k < NL else k - NL + 1
        hi_j = k if k < NL else NL          # exclusive
        for j in range(lo_j, hi_j):
            for p in range(n):
                mad(p, e.r("%s%d" % (prods[p]["out"], j)), e.r("P%d" % (k - j)))
        if k < NL:
            for p in range(n):
                m = e.r("%s%d" % (prods[p]["out"], k))
                e.ins("v_mul_lo_u32 %s, %s, %s" % (m, acc_regs[p].replace("[", "").split(":")[0], e.r("INV")))
            for p in range(n):
                m = e.r("%s%d" % (prods[p]["out"], k))
                e.ins("v_and_b32 %s, %s, %s" % (m, MASK, m))
            for p in range(n):
                mad(p, e.r("%s%d" % (prods[p]["out"], k)), e.r("P0"))
            for p in range(n):
                e.ins("v_ashrrev_i64 %s, 29, %s" % (acc_regs[p], acc_regs[p]))
        else:
            for p in range(n):
                o = e.r("%s%d" % (prods[p]["out"], k - NL))
                e.ins("v_and_b32 %s, %s, %s" % (o, MASK, acc_regs[p].replace("[", "").split(":")[0]))
            for p in range(n):
                e.ins("v_ashrrev_i64 %s, 29, %s" % (acc_regs[p], acc_regs[p]))
    for p in range(n):
        e.ins("v_mov_b32 %s, %s" % (e.r("%s%d" % (prods[p]["out"], NL - 1)), acc_regs[p].replace("[", "").split(":")[0]))


def mul_terms(a, b):
    def f(k):
        return [("%s%d" % (a, j), "%s%d" % (b, k - j)) for j in range(max(0, k - NL + 1), min(k, NL - 1) + 1)]
    return f


def sqr_terms(d, a):
    """symmetric products once against the doubled limb d = 2a (field29.h fe_sqr)"""
    def f(k):
        t = [("%s%d" % (d, j), "%s%d" % (a, k - j)) for j in range(max(0, k - NL + 1), NL) if 2 * j < k]
        if k % 2 == 0:
            t.append(("%s%d" % (a, k // 2), "%s%d" % (a, k // 2)))
        return t
    return f


def mulsub_terms(a, b, nc, d):
    def f(k):
        t = []
        for j in range(max(0, k - NL + 1), min(k, NL - 1) + 1):
            t.append(("%s%d" % (a, j), "%s%d" % (b, k - j)))
            t.append(("%s%d" % (nc, j), "%s%d" % (d, k - j)))
        return t
    return f


def statement(name, sig, outs, ins, prods_fn, n_acc):
    e = Emit()
    for key_prefix, cexpr in outs:
        for j in range(NL):
            e.op("%s%d" % (key_prefix, j), "=&v", "%s[%d]" % (cexpr, j))
    for key_prefix, cexpr in ins:
        for j in range(NL):
            e.op("%s%d" % (key_prefix, j), "v", "%s[%d]" % (cexpr, j))
    for j in range(NL):
        e.op("P%d" % j, "s", "(int32_t)F::P[%d]" % j)
    e.op("INV", "s", "(int32_t)F::INV")
    acc_regs = ["v[0:1]", "v[2:3]"][:n_acc]
    gen_product_columns(e, prods_fn(), acc_regs)
    n_out = len(outs) * NL
    out_ops = ", ".join('"%s"(%s)' % (c, x) for c, x in e.ops[:n_out])
    in_ops = ", ".join('"%s"(%s)' % (c, x) for c, x in e.ops[n_out:])
    clob = ", ".join('"v%d"' % i for i in range(2 * n_acc)) + ', "vcc"'
    body = "\n".join('        "%s\\n\\t"' % ln for ln in e.lines)
    n_instr = len(e.lines)
    txt = "// %d instructions\n%s {\n    asm(\n%s\n        : %s\n        : %s\n        : %s);\n}\n" % (n_instr, sig, body, out_ops, in_ops, clob)
    return txt, n_instr


def main():
    parts = []
    hdr = '''// fe_asm.h -- GENERATED by tools/gen_fe_asm.py, do not edit.  The Montgomery products of field29.h as one gfx950 inline-asm
// statement each (see the generator's docstring for the why and the register plan).  Device pass only; limb arrays in and out.
#pragma once
#include <cstdint>
#if defined(__HIP_DEVICE_COMPILE__)
namespace kzg {
'''
    parts.append(hdr)
    t, n1 = statement("fe_mul2_asm",
                      "template <class F>\n__device__ __forceinline__ void fe_mul2_asm(int32_t (&r1)[9], int32_t (&r2)[9], const int32_t (&a1)[9], const int32_t (&b1)[9], "
                      "const int32_t (&a2)[9], const int32_t (&b2)[9])",
                      [("r", "r1"), ("s", "r2")], [("a", "a1"), ("b", "b1"), ("c", "a2"), ("d", "b2")],
                      lambda: [{"terms": mul_terms("a", "b"), "out": "r"}, {"terms": mul_terms("c", "d"), "out": "s"}], 2)
    parts.append(t)
    t, n2 = statement("fe_sqr2_asm",
                      "// d1 = 2 a1, d2 = 2 a2 limb by limb (the caller doubles)\ntemplate <class F>\n__device__ __forceinline__ void fe_sqr2_asm(int32_t (&r1)[9], int32_t (&r2)[9], "
                      "const int32_t (&a1)[9], const int32_t (&d1)[9], const int32_t (&a2)[9], const int32_t (&d2)[9])",
                      [("r", "r1"), ("s", "r2")], [("a", "a1"), ("b", "d1"), ("c", "a2"), ("d", "d2")],
                      lambda: [{"terms": sqr_terms("b", "a"), "out": "r"}, {"terms": sqr_terms("d", "c"), "out": "s"}], 2)
    parts.append(t)
    t, n3 = statement("fe_mulsub_asm",
                      "// r = a b + nc d with nc = -c limb by limb (the caller negates): a b - c d, one reduction\ntemplate <class F>\n__device__ __forceinline__ void fe_mulsub_asm(int32_t (&r1)[9], "
                      "const int32_t (&a1)[9], const int32_t (&b1)[9], const int32_t (&nc)[9], const int32_t (&d1)[9])",
                      [("r", "r1")], [("a", "a1"), ("b", "b1"), ("c", "nc"), ("d", "d1")],
                      lambda: [{"terms": mulsub_terms("a", "b", "c", "d"), "out": "r"}], 1)
    parts.append(t)
    t, n4 = statement("fe_mul_asm",
                      "template <class F>\n__device__ __forceinline__ void fe_mul_asm(int32_t (&r1)[9], const int32_t (&a1)[9], const int32_t (&b1)[9])",
                      [("r", "r1")], [("a", "a1"), ("b", "b1")],
                      lambda: [{"terms": mul_terms("a", "b"), "out": "r"}], 1)
    parts.append(t)
    t, n5 = statement("fe_sqr_asm",
                      "// d1 = 2 a1 limb by limb (the caller doubles)\ntemplate <class F>\n__device__ __forceinline__ void fe_sqr_asm(int32_t (&r1)[9], const int32_t (&a1)[9], const int32_t (&d1)[9])",
                      [("r", "r1")], [("a", "a1"), ("b", "d1")],
                      lambda: [{"terms": sqr_terms("b", "a"), "out": "r"}], 1)
    parts.append(t)
    parts.append("constexpr int FE_ASM_INSTRUCTIONS_SQR = %d;\n" % n5)
    parts.append("constexpr int FE_ASM_INSTRUCTIONS_MUL2 = %d, FE_ASM_INSTRUCTIONS_SQR2 = %d, FE_ASM_INSTRUCTIONS_MULSUB = %d, FE_ASM_INSTRUCTIONS_MUL = %d;\n" % (n1, n2, n3, n4))
    parts.append("}  // namespace kzg\n#endif\n")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-kzg-bn254_amd", "csrc", "fe_asm.h")
    open(out, "w").write("\n".join(parts))
    print("wrote", out, "instructions:", n1, n2, n3, n4)


if __name__ == "__main__":
    main()
