"""The table of profiles/r06_batch_verify_tail.md (last section) and DESIGN.md 6.3 against the stored traces it was made from: tools/tail_summary.py over
profiles/r06_tail_traces/box_{w,x,y}*.txt must give 1 634 calls, 66 of them above 1.3 x their run's median, 46 of those with one thread's runqueue wait of
>= 0.6 x the excess, and no call in a throttled period of the process's cgroup.  CPU only; reads committed text files."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tail_summary_reproduces_the_documented_counts():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_tail_traces", "box_[wxy]*.txt")))
    assert len(files) == 8
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tail_summary.py")] + files, capture_output=True, text=True, check=True).stdout
    rows = re.findall(r"median ([\d.]+) ms, (\d+) slow of (\d+) calls, (\d+) explained by one thread's runqueue wait, (\d+) calls in a throttled period", out)
    assert len(rows) == 8
    calls = sum(int(r[2]) for r in rows); slow = sum(int(r[1]) for r in rows); explained = sum(int(r[3]) for r in rows); throttled = sum(int(r[4]) for r in rows)
    assert (calls, slow, explained, throttled) == (1634, 66, 46, 0)
    doc = open(os.path.join(ROOT, "profiles", "r06_batch_verify_tail.md")).read()
    assert "1 634 traced calls" in doc and "46 of the 66 slow calls" in doc
