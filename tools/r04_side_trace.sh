#!/bin/bash
# Do fixup_kernel (side stream, FEP_FIX_SIDE=1, ablation build) and element_kernel actually run at the same time?  Kernel trace.
out=gpurun_out/r4sidetrace; mkdir -p $out; rm -rf $out/*
cd /tmp; export TMPDIR=/tmp
export FEP_LIB_PATH=$GRAFT_REPO_ROOT/fem-elastoplasticity_amd/csrc/libfep_hip_abl.so FEP_FIX_SIDE=1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/tools/elem_bench.py P2 708 5 > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*kernel_trace.csv" | head -1); echo $f
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'element_kernel' in r['Kernel_Name'] or 'fixup_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
out=open(sys.argv[1].rsplit('/',1)[0]+'/../../overlap.txt','w')
for r in rows[-24:]:
    line=f"{r['Kernel_Name'][:28]:28s} queue {r.get('Queue_Id','?'):>3s} start {(int(r['Start_Timestamp'])-t0)/1e3:10.1f} us  end {(int(r['End_Timestamp'])-t0)/1e3:10.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}"
    print(line); out.write(line+'\n')
PY
