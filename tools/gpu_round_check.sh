cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 600 python bench.py --batch 16384 --steps 2 2>&1 | tail -3
