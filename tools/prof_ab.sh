#!/bin/bash
# usage: prof_ab.sh "<ENV>" <pattern> bench-args...
cd /tmp && export TMPDIR=/tmp
E="$1"; P="$2"; shift 2
rm -rf /tmp/pab; env $E rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pab -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-supplementary "$@" > /tmp/pab.log 2>&1
python3 - "$P" "$E" <<'PY'
import csv,glob,sys
f=glob.glob("/tmp/pab/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ref=[r for r in rows if "emm_grad" in r["Name"]][0]; steps=int(ref["Calls"])
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("[%s] kernel ms/step %.3f"%(sys.argv[2], tot/steps/1e6))
for r in rows:
    if any(p in r["Name"] for p in sys.argv[1].split(",")):
        print("   %7.3f ms/step n=%4.1f avg %7.1f us %s"%(float(r["TotalDurationNs"])/steps/1e6,int(r["Calls"])/steps,float(r["AverageNs"])/1e3,r["Name"].replace("(anonymous namespace)::","")[:70]))
PY
