set -o pipefail
timeout -k 10 900 python bench.py > gpurun_out/r2_bench1.log 2> gpurun_out/r2_bench1.err; echo "bench rc=$?" >> gpurun_out/r2_bench1.err
tail -25 gpurun_out/r2_bench1.err; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r2_bench1.log').read().strip().splitlines()[-1])
    for k in ("value","ms_per_step","latency_ms_single_scene","fp32","train","cpu_baseline"):
        print(k, d.get(k))
    print("roofline", {k:d["roofline"][k] for k in ("kernel","achieved","frac","avg_launch_us","exact_f32_kernel_us")})
    print("dense", {k:d["roofline_dense_stage"][k] for k in ("achieved","frac","ms_per_view","sparse3d_ms_per_view")})
    for r in d["roofline_kernels"]: print(r["kernel"], round(r["achieved"],1), r["unit"], round(r["frac"],3), round(r["avg_launch_us"],1))
except Exception as e:
    print("parse failed", e)
PY
