#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02_g_rows_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  for form in tile rows; do
    flag=""; [ $form == rows ] && flag="--rows"
    timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/${form}_$i -o r -- python $REPO/tools/bench_decoder.py --h0 $flag --iters 2 --burst 3 > $OUT/${form}_$i.log 2>&1 < /dev/null
  done
done
cd $REPO
python - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/*/r_counter_collection.csv"):
    form=f.split('/')[-2].split('_')[0]
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'mel_decoder' not in k: continue
        acc[(form, r['Counter_Name'])].append(float(r['Counter_Value']))
for (form,c),v in sorted(acc.items(), key=lambda x:(x[0][1],x[0][0])):
    print(f"{c:34s} {form:5s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
