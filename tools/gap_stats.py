#!/usr/bin/env python3
"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (one stream, steady-state window):
usage: gap_stats.py <kernel_trace.csv> [skip_fraction]   -> kernels, busy time, gap time, gap histogram"""
import csv, sys
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]                      # steady state: drop warm-up / capture
busy = sum(e - s for s, e, _ in rows)
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
span = rows[-1][1] - rows[0][0]
print("kernels %d  span %.2f ms  busy %.2f ms (%.1f%%)  gaps %.2f ms  mean gap %.2f us  median %.2f us" % (
    len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, sum(gaps) / 1e6, sum(gaps) / len(gaps) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9)):
    sel = [g for g in gaps if lo * 1e3 <= g < hi * 1e3]
    print("  gaps %3g-%-4g us: %6d  (%.2f ms)" % (lo, hi if hi < 1e9 else float('inf'), len(sel), sum(sel) / 1e6))
