set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
python bench.py --steps 30 --warmup 5 2>&1 | tail -3
python bench.py --steps 30 --warmup 5 --fwd-bwd-only --cpu-steps 0 2>&1 | tail -3
