#!/bin/bash
# Memory-path counters of one command (separate passes, --kernel-trace only): `tools/prof_mem.sh TAG python3 SCRIPT ARGS...`
# Writes gpurun_out/prof_TAG/mem*/ and a per-kernel table gpurun_out/prof_TAG/mem_counters.csv.
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum TCP_TOTAL_ACCESSES_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $out/mem$i -- "$@" > $out/mem$i.log 2>&1
done
python3 - "$out" <<'PY'
import collections, glob, os, sqlite3, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, 'mem*', '*', '*.db')):
    c = sqlite3.connect(f)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    pmc = [t for t in tabs if t.startswith('counters_collection') or t == 'counters_collection']
    try:
        rows = c.execute('select kernel_name, counter_name, sum(value) from counters_collection '
                         'group by kernel_name, counter_name, dispatch_id').fetchall()
    except Exception as e:
        print('no counters_collection view in', f, tabs[:8]); continue
    for k, n, v in rows:
        acc[k.replace('void ', '').split('(')[0].replace('fep::', '')][n].append(float(v))
names = sorted({n for k in acc for n in acc[k]})
with open(os.path.join(out, 'mem_counters.csv'), 'w') as fo:
    fo.write('kernel,' + ','.join(names) + '\n')
    for k in acc:
        fo.write('"%s",' % k + ','.join('%.0f' % (sum(acc[k][n]) / max(len(acc[k][n]), 1)) if n in acc[k] else '' for n in names) + '\n')
print(open(os.path.join(out, 'mem_counters.csv')).read())
PY
