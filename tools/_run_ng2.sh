set -e
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ng -o ng -- python3 $GRAFT_REPO_ROOT/tools/ng_prof.py 5 sorted > $GRAFT_REPO_ROOT/gpurun_out/ng_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_ng/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_n" in r["Name"] or "rocprim" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
