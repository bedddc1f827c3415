#!/usr/bin/env python3
"""count_isa.py -- regenerate bench.py's VALU_MIX_COUNTS / MADS_PER_MIXED_ADD from the compiler's own listing.

`roofline.valu` in bench.py prices ONE mixed addition of k_msm_accumulate's hot loop by instruction class.  Those counts used to be
typed in by hand from an ISA listing and drifted (179 masks typed, 153 in the binary).  This script compiles
csrc/msm.hip to gfx950 assembly (`hipcc -S`, device pass only, the library's own flags), finds the fast path through the
accumulate loop of k_msm_accumulate -- the run of fall-through blocks from the header of the loop that holds the 64-byte gather to the
join block behind the inlined xyzz_madd -- and counts its instructions by the classes bench.py prices:

    (v_mad_i64_i32, v_mul_lo_u32, 64-bit shifts [v_ashrrev_i64 / v_lshrrev_b64 / v_lshlrev_b64], v_and_b32, every other instruction, s_nop)

Usage:  python tools/count_isa.py            print the counts as JSON
        python tools/count_isa.py --check    exit 1 when bench.py's constants differ (tests/test_isa_counts.py runs this)
        python tools/count_isa.py --asm F    count an existing listing instead of compiling
No GPU needed: hipcc cross-compiles."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rust-kzg-bn254_amd", "csrc")
KERNEL = "k_msm_accumulate"
SHIFT64 = ("v_ashrrev_i64", "v_lshrrev_b64", "v_lshlrev_b64")


def compile_listing(path):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", os.path.join(CSRC, "msm.hip"), "-o", path]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=CSRC)


def kernel_blocks(lines):
    """Basic blocks of the kernel in layout order: (label or None, [mnemonics], branch target or None).  Labels, directives and
    comments are dropped; a block ends at a label or behind a branch."""
    start = next(i for i, ln in enumerate(lines) if re.match(r"^_ZN3kzg\d+%s\w*:" % KERNEL, ln))
    blocks, cur, label = [], [], "entry"
    for ln in lines[start + 1:]:
        s = ln.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            if cur or label:
                blocks.append((label, cur, None))
            cur, label = [], m.group(1)
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        cur.append(op)
        if op.startswith("s_cbranch") or op == "s_branch":
            blocks.append((label, cur, s.split()[1]))
            cur, label = [], None
    if cur:
        blocks.append((label, cur, None))
    return blocks


def hot_path(blocks):
    """The fast path through the accumulate loop: LLVM lays the likely successor out as the fall-through, so from the header of the loop
    that holds the 64-byte gather (4 x global_load_dwordx4) the path is the run of consecutive blocks up to the join block behind the
    last multiply-add (the rarely taken branches -- bucket boundary, identity base, P = +-Q -- leave this run and are not counted)."""
    index = {lab: i for i, (lab, _, _) in enumerate(blocks) if lab}
    best = None
    for i, (_, _, target) in enumerate(blocks):
        if target is None or target not in index or index[target] > i:
            continue                                          # not a backward branch
        h = index[target]
        body = [op for _, ops, _ in blocks[h:i + 1] for op in ops]
        if sum(1 for op in body if op.startswith("global_load_dwordx4")) < 4:
            continue
        path, seen_mad = [], False
        for _, ops, _ in blocks[h:i + 1]:
            mads = sum(1 for op in ops if op.startswith("v_mad_i64_i32"))
            if seen_mad and mads == 0:
                break
            seen_mad = seen_mad or mads > 0
            path += ops
        n_mads = sum(1 for op in path if op.startswith("v_mad_i64_i32"))
        if best is None or n_mads > best[0]:
            best = (n_mads, path)
    if best is None:
        raise SystemExit("count_isa: no loop with a 64-byte gather found in %s" % KERNEL)
    return best[1]


def classify(block):
    c = [0] * 6
    for op in block:
        base = op.split("_e32")[0].split("_e64")[0]
        if base == "v_mad_i64_i32":
            c[0] += 1
        elif base == "v_mul_lo_u32":
            c[1] += 1
        elif base in SHIFT64:
            c[2] += 1
        elif base == "v_and_b32":
            c[3] += 1
        elif base == "s_nop":
            c[5] += 1
        else:
            c[4] += 1
    return c


def count(asm_path=None):
    if asm_path is None:
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "msm.s")
            compile_listing(p)
            lines = open(p).read().splitlines()
    else:
        lines = open(asm_path).read().splitlines()
    blocks = kernel_blocks(lines)
    hot = hot_path(blocks)
    counts = classify(hot)
    vgpr = None
    start = next(i for i, ln in enumerate(lines) if re.match(r"^_ZN3kzg\d+%s\w*:" % KERNEL, ln))
    for ln in lines[start:]:
        m = re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", ln)
        if m:
            vgpr = int(m.group(1))
            break
    return {"kernel": KERNEL, "hot_block_instructions": len(hot), "mix": counts,
            "mix_classes": ["v_mad_i64_i32", "v_mul_lo_u32", "64-bit shifts", "v_and_b32", "other", "s_nop"],
            "mads_per_mixed_add": counts[0], "vgprs": vgpr, "mfma_in_kernel": sum(1 for _, ops, _ in blocks for op in ops if op.startswith("v_mfma")),
            "scratch_ops_in_hot_block": sum(1 for op in hot if op.startswith("scratch_") or op.startswith("buffer_") and "scratch" in op)}


def bench_constants():
    src = open(os.path.join(ROOT, "bench.py")).read()
    mix = tuple(int(v) for v in re.search(r"^VALU_MIX_COUNTS\s*=\s*\(([^)]*)\)", src, re.M).group(1).split(","))
    mads = int(re.search(r"^MADS_PER_MIXED_ADD\s*=\s*(\d+)", src, re.M).group(1))
    return mix, mads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--strict", action="store_true", help="with --check: any drifting class fails, not only the multiply-adds")
    ap.add_argument("--asm")
    args = ap.parse_args()
    got = count(args.asm)
    print(json.dumps(got))
    if args.check:
        mix, mads = bench_constants()
        # structural: the multiply-adds of one mixed addition (8M + 2S with one fused reduction) -- a different count is a different formula.
        # The other classes (shifts, masks, moves, nops) are scheduling detail that a ROCm point release may move by a few: reported, not fatal
        # (ADVICE r4), unless --strict.
        if got["mads_per_mixed_add"] != mads or got["mix"][0] != mix[0]:
            print("bench.py drifted: MADS_PER_MIXED_ADD = %d, the compiler's listing says %d" % (mads, got["mads_per_mixed_add"]), file=sys.stderr)
            sys.exit(1)
        if tuple(got["mix"]) != mix:
            print("warning: bench.py VALU_MIX_COUNTS = %s, the compiler's listing says %s (regenerate with tools/count_isa.py)" % (mix, tuple(got["mix"])), file=sys.stderr)
            if args.strict:
                sys.exit(1)


if __name__ == "__main__":
    main()
