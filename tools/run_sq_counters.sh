# Where the production kernel's cycles go: SQ counters of et_stream_kernel on the tiled global
# grid, a few per rocprofv3 pass (tools/tiledbench.py launches). Results: gpurun_out/TAG/summary.txt
#   bash tools/run_sq_counters.sh TAG [dtype] [math]
set -e
TAG=${1:-sq}
DT=${2:-float64}
MATH=${3:-fast}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
rm -rf $O && mkdir -p $O
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32" \
           "VALUBusy SALUBusy LDSBankConflict MemUnitStalled"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/p$i -- python3 tools/tiledbench.py --no-plain --launches 3 --rounds 1 --dtype $DT --math $MATH > $O/p$i.out 2> $O/p$i.err || { tail -3 $O/p$i.err; }
done
O=$O python3 - <<'PY' > $O/summary.txt
import csv, glob, os, collections
O = os.environ['O']
acc = collections.defaultdict(list)
for f in glob.glob(O + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'et_stream_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    v = acc[k]
    print('%-28s mean %.6g  (launches %d)' % (k, sum(v) / len(v), len(v)))
PY
find $O -name "*.csv" -delete
cat $O/summary.txt
