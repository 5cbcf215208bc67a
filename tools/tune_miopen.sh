#!/bin/bash
# Regenerate rel_pose_amd/miopen_db/ on an MI355X box: one MIOpen search (find + perf-db tuning) over the CNN front-end's
# convolutions at the bench batch, then copy the user dbs it wrote.  ~1-2 minutes.
set -e
export MIOPEN_USER_DB_PATH=$(mktemp -d)
cp $GRAFT_REPO_ROOT/rel_pose_amd/miopen_db/*.txt $MIOPEN_USER_DB_PATH/ 2>/dev/null     # extend the shipped db (MIOpen appends)
export RP_CUDNN_BENCHMARK=1 MIOPEN_FIND_ENFORCE=4
python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" | tail -1 | cut -c1-200
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/miopen_db
cp $MIOPEN_USER_DB_PATH/*.txt $GRAFT_REPO_ROOT/gpurun_out/miopen_db/
ls -la $GRAFT_REPO_ROOT/gpurun_out/miopen_db
