#!/bin/bash
# rocprofv3 passes of one command: `tools/prof.sh TAG python3 SCRIPT ARGS...`  (the program itself after the tag:
# python3 / a binary, never a shell or env wrapper).  Writes gpurun_out/prof_TAG/{stats,fetch,write,sq,sq2}/ and then
# profiles-ready summaries through tools/summarize_counters.py.  Counter passes are separate runs (PMC slot limits;
# --pmc is never combined with the runtime trace domains).
set -e
tag=$1; shift
extra=""
if [ "$1" == "--traffic-latest" ]; then extra="--traffic-latest"; shift; fi
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/stats -- "$@" > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -- "$@" > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -- "$@" > $out/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY -d $out/sq -- "$@" > $out/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $out/sq2 -- "$@" > $out/sq2.log 2>&1
python3 tools/summarize_counters.py $tag $out $extra
# the summaries land in profiles/ of THIS checkout; on a gpurun box only gpurun_out/ travels back: leave copies there
cp profiles/${tag}_kernel_stats.csv profiles/${tag}_counters.csv $out/ 2>/dev/null || true
if [ "$extra" == "--traffic-latest" ]; then cp profiles/traffic_latest.json $out/ 2>/dev/null || true; fi
# the raw traces do not travel (gpurun merges at most 64 MiB back; bench.py's pre-heat alone leaves thousands of launches in each)
rm -rf $out/stats $out/fetch $out/write $out/sq $out/sq2
