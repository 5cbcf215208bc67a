#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/peak_i.txt; rm -f $O
for k in 0 1; do
  for wgs in 256 512 768 1024 1280 1536 2048; do
    tools/lab/peak_lab $k $wgs 4000 3 0 >> $O
  done
done
cat $O
