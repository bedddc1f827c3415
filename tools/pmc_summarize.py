"""Summarises rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass as MI355X_MICROARCH.md prescribes)
into profiles/<name>.json: per kernel, average KB per dispatch.  bench.py reads the k_msm_accumulate entry for
`roofline.traffic`.

Units / gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in KB.  FETCH_SIZE halves wide
coalesced 16 B/lane STREAMING reads; k_msm_accumulate's reads are 64-byte random gathers (one 64 B point per lane,
4 x dwordx4), calibrated here against the known byte count of the gather (entries x 64 B + entries x 4 B):
the counter reads 1.2x that minimum, i.e. it is NOT halved for this pattern, so no doubling is applied.
Usage: python tools/pmc_summarize.py <fetch_csv> <write_csv> <out_json> [log_n]"""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sources k_msm_accumulate is compiled from: the summary is stamped with their digest, and bench.py reports `roofline.traffic` as
# stale (null + the reason) when the tree no longer matches -- the one figure of the line the run does not produce itself
KERNEL_SOURCES = ("msm_kernels.h", "msm.hip", "curve.h", "field29.h", "fe_asm.h")


def kernel_sources_sha256(root=ROOT):
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(root, "rust-kzg-bn254_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0].replace("kzg::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    out = {"units": "KB per dispatch (average)", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        out["kernels"][k] = {"FETCH_SIZE_KB": fetch.get(k), "WRITE_SIZE_KB": write.get(k)}
    out["log_n"] = int(sys.argv[4]) if len(sys.argv) > 4 else 20          # workload the passes were collected on (bench.py checks it)
    out["kernel_sources"] = list(KERNEL_SOURCES)
    out["kernel_sources_sha256"] = kernel_sources_sha256()                 # of the tree the passes ran on (the snapshot gpurun sent)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    acc = out["kernels"].get("k_msm_accumulate")
    if acc:
        print("k_msm_accumulate HBM bytes per launch:", (acc["FETCH_SIZE_KB"] + acc["WRITE_SIZE_KB"]) * 1024)


if __name__ == "__main__":
    main()
