#!/usr/bin/env python3
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
if steps <= 0:      # infer the number of steps in the window from a kernel that runs exactly 5x per step
    for r in rows:
        if 'attn_fwd_kernel<2, false' in r['Name'] or 'attn_fwd_bf16_kernel<2, false' in r['Name'] or 'attn_fwd_bf16_kernel<6, false' in r['Name']:
            steps = float(r['Calls']) / 5.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.2f ms  (%.2f ms/step over %g steps)" % (tot / 1e6, tot / 1e6 / steps, steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    n = r['Name'].replace('(anonymous namespace)::', '')
    print("%6.2f%% calls=%6s avg=%9.1fus /step=%7.3fms  %s" % (float(r['Percentage']), r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6 / steps, n[:100]))
