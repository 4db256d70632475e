#!/bin/bash
# Development, on the GPU box: SQ counters of every kernel a command launches, averaged per kernel name.
# usage: tools/pmc_kernel.sh <outdir under gpurun_out> -- <command...>
set -u
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/$1; shift; shift
mkdir -p $OUT
export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC"; do
  name=pmc_$(echo $grp | cut -d' ' -f1)
  (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$name -o r -- "$@" > $OUT/$name.log 2>&1 < /dev/null)
done
python - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} n={len(v):4d} mean={sum(v) / len(v):14.1f}")
PY
