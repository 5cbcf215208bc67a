#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -k "config3 or interiornet or ddp or demo" > gpurun_out/m_tests.txt 2>&1
echo "new gpu tests rc=$?"; tail -15 gpurun_out/m_tests.txt
