#!/bin/bash
# GPU box: tests + A/B (stage times, timed step) of the working copy of one .hip file against the .head copy next to it
# usage: tools/dev/r6_ab.sh <file.hip> <outdir-name> [pytest files...]
cd "$(dirname "$0")/../.."
FILE=$1; O=gpurun_out/$2; shift 2; mkdir -p $O
timeout 900 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -4 > $O/test.txt
AB_STAGES=1 bash tools/ab_bench.sh $FILE 3 > $O/stages.txt 2>&1
bash tools/ab_bench.sh $FILE 3 > $O/step.txt 2>&1
