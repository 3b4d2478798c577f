#!/bin/bash
# Memory-path counters (vector L1 = TCP, L2 = TCC, its memory interface = EA, address / data units TA / TD) of the element route's
# kernels and, for comparison, of the P1 node route's: where do the requests wait?  One rocprofv3 --pmc pass per counter group.
# usage: gpurun -- 'bash tools/r04_mempath.sh'
out=$PWD/gpurun_out/r4mem; mkdir -p $out; rm -rf $out/*
export TMPDIR=/tmp
R=$PWD
cd /tmp
pass() { # tag type cells counters...   (every pass under its own time limit: a refused counter set leaves rocprofv3 hanging)
  tag=$1; t=$2; n=$3; shift 3
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/raw_${tag}_$t -- python3 $R/tools/elem_bench.py $t $n 10 > $out/${tag}_$t.log 2>&1 || echo "pass $tag $t failed" | tee -a $out/progress.txt
  echo "pass $tag $t done" >> $out/progress.txt
}
for tn in "P2 708" "P1 708"; do set -- $tn
pass a1 $1 $2 TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_GATE_EN1
pass a2 $1 $2 TCP_TCC_WRITE_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_TCP_TA_DATA_STALL_CYCLES TCP_TOTAL_ACCESSES
pass b1 $1 $2 TCC_REQ TCC_HIT TCC_MISS TCC_EA0_WRREQ
pass b2 $1 $2 TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL
pass c1 $1 $2 TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL
pass c2 $1 $2 TCC_BUSY TCC_CYCLE TCC_SRC_FIFO_FULL TCC_LATENCY_FIFO_FULL
pass d1 $1 $2 TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE
pass d2 $1 $2 TA_DATA_STALLED_BY_TC_CYCLES TD_TC_STALL TD_TD_BUSY
pass e $1 $2 TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_LFIFO_STALL_CYCLES
done
cd $R
python3 - $out <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/raw_*/**/*counter_collection.csv', recursive=True):
    t = f.split('/raw_')[1].split('/')[0].split('_')[1]
    per = collections.defaultdict(lambda: collections.defaultdict(float))      # (dispatch) -> counter -> value
    names = {}
    for r in csv.DictReader(open(f)):
        k = r['Dispatch_Id']; names[k] = r['Kernel_Name']
        per[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, c in per.items():
        n = names[k]
        if not any(s in n for s in ('element_kernel', 'fixup_kernel', 'p1_point_kernel', 'p1_node_lds_kernel')):
            continue
        short = n.split('(')[0].replace('void fep::', '').replace('fep::', '')[:44]
        for cn, v in c.items():
            res[(t, short)][cn].append(v)
with open(out + '/mempath.txt', 'w') as fo:
    for (t, k), c in sorted(res.items()):
        line = f'{t} {k}: ' + ', '.join(f'{cn} {sorted(v)[len(v)//2]:.4g}' for cn, v in sorted(c.items()))
        print(line); fo.write(line + '\n')
PY
rm -rf $out/raw_*
