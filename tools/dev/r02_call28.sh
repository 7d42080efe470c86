#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/dev/pk_probe.py > gpurun_out/r02_pk_probe.txt 2>&1; grep -c . gpurun_out/r02_pk_probe.txt; grep "alone\|v_pk_mov\|,1\] op_sel_hi:\[1,1,0\]" gpurun_out/r02_pk_probe.txt | cut -c1-200 | head -60
