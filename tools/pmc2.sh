#!/bin/bash
# usage: tools/pmc2.sh <tag> <python args...>
# Counter passes (rocprofv3 --pmc, one group per pass, with --kernel-trace only -- never combined with sys/hip/hsa traces) over
# one python command; per kernel symbol: average of every counter + the average dispatch duration of the SAME pass.
# -> gpurun_out/pmc_<tag>.json and gpurun_out/pmc_<tag>_summary.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc2 && mkdir -p /tmp/pmc2
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc2/p$i -o p -- python "$@" > /dev/null 2>&1
  i=$((i+1))
done
python3 - "$tag" <<'PY'
import csv, glob, json, os, sys, collections
tag = sys.argv[1]
KEEP = ('dw192', 'dx_lnbwd', 'gemm_', 'attn_', 'emm_', 'colsum', 'ln_', 'splitk', 'rowdot', 'tokens_', 'mlp_', 'linear_', 'ds_matmul', 'conv', 'bn_')
def short(k):
    return k.replace('(anonymous namespace)::', '').replace('rpgemm::', '').replace('void ', '').split('(')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))     # kernel -> pass -> duration
for d in sorted(glob.glob('/tmp/pmc2/p*')):
    pid = os.path.basename(d)
    disp = {}
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if any(s in k for s in KEEP):
                dur[k][pid][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3; dur[k][pid][1] += 1
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if any(s in k for s in KEEP):
                a = acc[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
out = {}
for k in acc:
    c = {n: v[0] / max(v[1], 1) for n, v in acc[k].items()}
    ds = {p: v[0] / max(v[1], 1) for p, v in dur[k].items()}
    c['launches_per_pass'] = max(v[1] for v in acc[k].values())
    c['duration_us_pass0'] = ds.get('p0', 0.0)        # pass with the SQ counters
    c['duration_us_pass2'] = ds.get('p2', 0.0)        # pass with GRBM_GUI_ACTIVE
    out[k] = c
root = os.environ.get('GRAFT_REPO_ROOT', '.')
json.dump(out, open(os.path.join(root, 'gpurun_out', 'pmc_%s.json' % tag), 'w'), indent=1, sort_keys=True)
with open(os.path.join(root, 'gpurun_out', 'pmc_%s_summary.txt' % tag), 'w') as f:
    f.write("# per kernel symbol, averages per launch.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs);\n"
            "# clock = GRBM_GUI_ACTIVE/8 / duration of the same pass; HBM = 2*FETCH_SIZE KiB (gfx950 tallies 128-B requests at 64 B) + WRITE_SIZE KiB\n")
    for k in sorted(out, key=lambda k: -out[k].get('GRBM_GUI_ACTIVE', 0) * out[k]['launches_per_pass']):
        c = out[k]; g = lambda n: c.get(n, 0.0)
        gui = g('GRBM_GUI_ACTIVE') / 8
        if gui <= 0: continue
        wc = max(g('SQ_WAVE_CYCLES'), 1)
        hbm = 2 * g('FETCH_SIZE') * 1024 + g('WRITE_SIZE') * 1024
        d2 = max(c['duration_us_pass2'], 1e-9)
        f.write("%-42s n=%3d  %8.1f us  clock %.2f GHz | MFMA busy %.3f | wave-cycles: wait %.2f issue-stall %.2f active %.2f | "
                "VALU/MFMA insts %.2f LDS/MFMA %.2f | LDS conflict %.3f | HBM %.1f MB (%.2f TB/s) fetch %.1f write %.1f | L2 hit %.2f\n" % (
                    k[:42], c['launches_per_pass'], d2, gui / d2 * 1e-3, g('SQ_VALU_MFMA_BUSY_CYCLES') / (gui * 1024),
                    g('SQ_WAIT_ANY') / wc, g('SQ_WAIT_INST_ANY') / wc, g('SQ_ACTIVE_INST_ANY') / wc,
                    g('SQ_INSTS_VALU') / max(g('SQ_INSTS_MFMA'), 1), g('SQ_INSTS_LDS') / max(g('SQ_INSTS_MFMA'), 1),
                    g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1), hbm / 1e6, hbm / d2 * 1e-6,
                    2 * g('FETCH_SIZE') * 1024 / 1e6, g('WRITE_SIZE') * 1024 / 1e6,
                    g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum'), 1)))
print(open(os.path.join(root, 'gpurun_out', 'pmc_%s_summary.txt' % tag)).read())
PY
