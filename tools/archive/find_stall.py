"""Diagnostic: the longest kernels of a rocprofv3 kernel trace (csv) and the timeline of the long ones at its end."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
t0 = ev[0][0]
print("longest kernels:")
for s, e, n, q in sorted(ev, key=lambda x: x[1] - x[0], reverse=True)[:12]:
    print("  %10.3f ms  dur %8.3f ms  q %s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, n))
print("kernels longer than 0.25 ms among the last 700 events:")
for s, e, n, q in ev[-700:]:
    if e - s > 250000:
        print("  %10.3f ms  dur %8.3f ms  q %s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, n))
print("gaps longer than 1.5 ms between consecutive kernels (any queue), with the kernels on either side:")
last_end, last_name = ev[0][1], ev[0][2]
for s, e, n, q in ev[1:]:
    if s - last_end > 1500000 and (s - t0) / 1e6 < 1290:
        print("  at %10.3f ms: %7.3f ms idle after %-40s before %s (q %s)" % ((last_end - t0) / 1e6, (s - last_end) / 1e6, last_name[:40], n[:40], q))
    if e > last_end:
        last_end, last_name = e, n
