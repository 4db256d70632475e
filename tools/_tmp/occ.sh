cd /tmp; export TMPDIR=/tmp
for B in 256 1024 1792; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/occ/b$B -o r -- python $GRAFT_REPO_ROOT/bench.py --phonemes 32 --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-auto-launch > /dev/null 2>&1
echo "B=$B"; python - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/occ/b$B/**/r_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "esmi" in r["Name"]: print("  ", r["Name"][11:60], r["Calls"], r["AverageNs"])
PY
done
