"""Developer probe: share of the wall time between the first and the last kernel of a rocprofv3 --kernel-trace CSV during which
at least one kernel was running, and the same with at least two (overlap of the batches in flight).
usage: dev_gpu_busy.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t0 = iv[0][0] + (iv[-1][1] - iv[0][0]) * skip   # the later part of the run: steady-state frames
ev = []
for a, b in iv:
    if b <= t0: continue
    ev.append((max(a, t0), 1)); ev.append((b, -1))
ev.sort()
busy1 = busy2 = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
wall = ev[-1][0] - ev[0][0]
print("window %.1f ms: some kernel running %.3f, two or more %.3f" % (wall / 1e6, busy1 / wall, busy2 / wall))
