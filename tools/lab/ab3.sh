run() { env $1 python bench.py --steps 30 --warmup 5 --no-supplementary --no-cpu-baseline $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$1 $2 ->', d['ms_per_step'], r['kernel'], r['avg_launch_us'])"; }
for i in 1 2; do
run RP_DW_SPLIT3=0 "--timer-instance dw192_f32"
run RP_DW_SPLIT3=1 "--timer-instance dw192_split3"
run RP_DW_SPLIT3=1 "--timer-instance attn_bwd_dkdv_p"
run RP_DW_SPLIT3=0 "--timer-instance attn_bwd_dkdv_p"
run RP_DW_SPLIT3=1 ""
run RP_DW_SPLIT3=0 ""
done
