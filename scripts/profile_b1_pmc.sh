#!/bin/bash
# Run ON the GPU box: PMC passes (own runs, no trace domains besides the kernel trace) of the batch-1 step -> gpurun_out/pmc_b1/
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_b1; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD="python $ROOT/bench.py --steps 10 --warmup 3 --batch 1 --no-cpu-baseline --no-parity --track-frames 0 --exact-steps --streams 1"
SE3TN_NOCHECK=1 SE3TN_NO_ALT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > /dev/null 2> $OUT/trace.err
SE3TN_NOCHECK=1 SE3TN_NO_ALT=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -o pmc -- $CMD > /dev/null 2> $OUT/sq.err
SE3TN_NOCHECK=1 SE3TN_NO_ALT=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/lds -o pmc -- $CMD > /dev/null 2> $OUT/lds.err
find $OUT -name "*.csv" | head; python - <<'PY'
import csv, glob, collections, os
OUT=os.environ.get("OUT","/root/repo/gpurun_out/pmc_b1")
def load(pat):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in acc.items()}
sq=load(OUT+"/sq/**/*counter_collection.csv"); lds=load(OUT+"/lds/**/*counter_collection.csv")
dur={}
for f in glob.glob(OUT+"/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)): dur[r["Name"]]=float(r["AverageNs"])
print("%-64s %8s %9s %9s %8s %8s" % ("kernel","us","mfma busy","eff GHz","wait any","lds conf"))
for k,d in sorted(sq.items(), key=lambda kv:-dur.get(kv[0],0)):
    if "se3tn" not in k: continue
    gui=d.get("GRBM_GUI_ACTIVE",0); mf=d.get("SQ_VALU_MFMA_BUSY_CYCLES",0); wc=d.get("SQ_WAVE_CYCLES",0)
    l=lds.get(k,{})
    print("%-64s %8.1f %8.1f%% %9.2f %7.1f%% %7.2f%%" % (k[:64], dur.get(k,0)/1e3, 100*mf/(gui/8*1024) if gui else 0, gui/8/dur[k] if k in dur else 0,
          100*d.get("SQ_WAIT_ANY",0)/wc if wc else 0, 100*l.get("SQ_LDS_BANK_CONFLICT",0)/l["SQ_LDS_IDX_ACTIVE"] if l.get("SQ_LDS_IDX_ACTIVE") else 0))
PY
