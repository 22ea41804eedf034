#!/bin/bash
cd "$(dirname "$0")/../.."
python -m pytest tests -m gpu -q 2>&1 | grep "passed\|failed\|Error" | head -3
python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/dev/sweep.sh 131072 33600 50505 2>&1 | tail -4 | cut -c1-700
bash tools/dev/sweep.sh 131072 33600 60606 1 2>&1 | tail -4 | cut -c1-700
