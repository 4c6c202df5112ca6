#!/bin/bash
# round 6, call F: the prefilter kernel's host-side knobs re-measured after the final-ranking changes:
# two sample tiles per wave from KK >= 311 (PSG_KNN_SAMPLE2_KK) and the second histogram level from KK >= 100 (PSG_KNN_FINE_CUT_KK)
mkdir -p gpurun_out/r6k
run() { name=$1; shift; env "$@" BLOCKS=$B PSG_GCN_KNN_BF_MAXD=27 python tools/knn_real_feats.py > gpurun_out/r6k/$name.log 2>&1; echo "== $name"; grep -h "^block" gpurun_out/r6k/$name.log | cut -c1-20,116-400; }
B=21,23,25,27
run s2_default X=1
run s2_off PSG_KNN_SAMPLE2_KK=999
run s2_350 PSG_KNN_SAMPLE2_KK=350
B=3,6,9,12,16
run fine_default X=1
run fine_off PSG_KNN_FINE_CUT_KK=999
run fine_40 PSG_KNN_FINE_CUT_KK=40
