#!/bin/bash
cd "$(dirname "$0")/../.."
for i in 1 2 3 4; do
  python tools/dev/pipe_ab.py 65536 10 | cut -c1-110
  DMZ_HIP_NO_OVERLAP=1 python tools/dev/pipe_ab.py 65536 10 | cut -c1-110
done
