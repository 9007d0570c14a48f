set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 100 --warmup 10 --cpu-steps 0 2>&1 | tail -1 | cut -c1-2400
python bench.py --steps 50 --warmup 10 --cpu-steps 0 --mode eager 2>&1 | tail -1 | cut -c1-600
