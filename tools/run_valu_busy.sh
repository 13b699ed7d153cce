# VALU utilisation of the production kernel per data type / arithmetic
# (rocprofv3 derived metrics VALUBusy, MemUnitStalled; one pass each).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/valu_busy
rm -rf $O && mkdir -p $O
for cfg in "float64 fast" "float32 fast" "float32 mixed"; do
  set -- $cfg
  tag=$1_$2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc VALUBusy MemUnitStalled --output-format csv -d $O/$tag -- python3 tools/tiledbench.py --no-plain --launches 4 --rounds 1 --dtype $1 --math $2 > $O/$tag.out 2> $O/$tag.err || tail -3 $O/$tag.err
  O=$O tag=$tag python - <<'PY'
import csv, glob, os, collections
O, tag = os.environ['O'], os.environ['tag']
acc = collections.defaultdict(list)
for f in glob.glob(O + '/' + tag + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'et_stream_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(tag, {k: round(sum(v) / len(v), 2) for k, v in acc.items()}, 'launches', {k: len(v) for k, v in acc.items()})
PY
done
find $O -name "*.csv" -size +1M -delete
