#!/bin/bash
# Development, on the GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE, one pass each) and trace duration of the mel decoder for several
# builds of the library.   usage: tools/traffic_ab.sh <tag> <config> <lib.so> [<lib.so> ...]   (paths relative to the repo root)
TAG=$1; CFG=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  for grp in FETCH_SIZE WRITE_SIZE; do
    ESMI_LIB=$REPO/$lib timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$name/$grp -o r -- \
        python $REPO/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-auto-launch > $OUT/$name.$grp.log 2>&1 < /dev/null
  done
done
python - $OUT "$@" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for lib in sys.argv[2:]:
    name = os.path.basename(lib)[:-3]
    res = {}
    for grp in ("FETCH_SIZE", "WRITE_SIZE"):
        per = {}
        for f in glob.glob(f"{out}/{name}/{grp}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "mel_decoder_kernel" in r["Kernel_Name"] and r["Counter_Name"] == grp:
                    per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        res[grp] = sum(per.values()) / max(len(per), 1) * 1024
        ts = []
        for f in glob.glob(f"{out}/{name}/{grp}/**/*kernel_trace.csv", recursive=True):
            ts += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "mel_decoder_kernel" in r["Kernel_Name"]]
        res["us_" + grp] = sum(ts) / max(len(ts), 1)
    print(f"{name:28s} fetch x2 {2 * res['FETCH_SIZE'] / 1e6:8.1f} MB  write {res['WRITE_SIZE'] / 1e6:8.1f} MB  total {(2 * res['FETCH_SIZE'] + res['WRITE_SIZE']) / 1e6:8.1f} MB"
          f"   kernel {res['us_FETCH_SIZE']:.1f} / {res['us_WRITE_SIZE']:.1f} us")
PY
