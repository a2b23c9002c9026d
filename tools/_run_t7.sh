set -e
timeout -k 10 400 python -m pytest tests/test_gpu_model.py tests/test_scannet_reader.py -m gpu -x -q --durations=8 2>&1 | tail -22
timeout -k 10 200 python tools/timeline_events.py 4 4 nogc > gpurun_out/r02_timeline_events_g4.txt 2>&1
tail -40 gpurun_out/r02_timeline_events_g4.txt
