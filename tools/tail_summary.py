"""Slow calls of traced batch verifications against what the kernel says about the process's threads.
Input: outputs of tools/trace_batch_verify.py (profiles/r06_tail_traces/*.txt).  A call is SLOW above 1.3 x the run's median; it is EXPLAINED when, during it, one
thread of the process sat runnable without a CPU (/proc/self/task/<tid>/schedstat, field 2) for more than 0.6 x the call's excess over the median.
Usage: python tools/tail_summary.py profiles/r06_tail_traces/*.txt"""
import statistics as st
import sys

for path in sys.argv[1:]:
    out = open(path).read().splitlines()
    ts = [float(x) for x in [ln for ln in out if ln.startswith("per call ms:")][0].split("|")[0].split()[3:]][2:]     # the first two calls size buffers and start the pool
    ws = [tuple(map(float, x.split("/"))) for x in [ln for ln in out if ln.startswith("runqueue wait")][0].split("):")[1].split()][2:]
    th = [ln for ln in out if ln.startswith("cgroup throttling")][0].split("):")[1].split()
    cpu = [ln for ln in out if ln.startswith("process CPU")][0].split("(")[0].strip()
    med = st.median(ts)
    slow = [(i + 2, t, a) for i, (t, (a, _b)) in enumerate(zip(ts, ws)) if t > 1.3 * med]
    explained = [x for x in slow if x[2] > 0.6 * (x[1] - med)]
    print("%s: median %.2f ms, %d slow of %d calls, %d explained by one thread's runqueue wait, %d calls in a throttled period of the process's cgroup, %s"
          % (path.split("/")[-1], med, len(slow), len(ts), len(explained), sum(1 for x in th if not x.startswith("0/")), cpu))
    if slow:
        print("    call:ms/longest runqueue wait  " + " ".join("%d:%.1f/%.1f" % x for x in slow))
