set -e
timeout -k 10 400 python -m pytest tests/test_gpu_spconv.py tests/test_seams.py -m gpu -x -q 2>&1 | tail -4
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 200 python tools/prof_3d.py 10 batch 2>&1 | tail -2
