# rocprofv3 evidence for forms of the forward run on the tiled layout: kernel stats, VALUBusy /
# MemUnitStalled, FETCH_SIZE and WRITE_SIZE (separate passes), per "FORM DTYPE MATH" given in CASES.
#   CASES="raw:float64:fast raw_total8_hours:float64:fast raw:float32:mixed" bash tools/run_form_profiles.sh TAG
set -e
TAG=${1:-forms}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
rm -rf $O && mkdir -p $O
: > $O/summary.jsonl
for c in ${CASES:-raw:float64:fast}; do
  IFS=: read form dtype math <<< "$c"
  tag=${form}_${dtype}_${math}
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag/stats -- python3 tools/formbench.py $form $dtype $math 10800 10 > $O/$tag.line 2> $O/$tag.err || { tail -5 $O/$tag.err; exit 1; }
  for pmc in "VALUBusy MemUnitStalled" FETCH_SIZE WRITE_SIZE; do
    d=$O/$tag/pmc_$(echo $pmc | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $d -- python3 tools/formbench.py $form $dtype $math 10800 3 > /dev/null 2>> $O/$tag.err || { tail -5 $O/$tag.err; exit 1; }
  done
  O=$O tag=$tag python3 - <<'PY' >> $O/summary.jsonl
import csv, glob, json, os, collections
O, tag = os.environ['O'], os.environ['tag']
line = json.loads([l for l in open(O + '/' + tag + '.line') if l.startswith('{')][-1])
acc = collections.defaultdict(list)
for f in glob.glob(O + '/' + tag + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'et_stream_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
stats = None
for f in glob.glob(O + '/' + tag + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'et_stream_kernel' in r['Name']:
            stats = {'kernel': r['Name'][:80], 'calls': int(r['Calls']), 'avg_ms': float(r['AverageNs']) / 1e6,
                     'min_ms': float(r['MinNs']) / 1e6, 'max_ms': float(r['MaxNs']) / 1e6}
mean = {k: sum(v) / len(v) for k, v in acc.items()}
n = line['pixels']
rec = dict(line, rocprof=stats, VALUBusy=mean.get('VALUBusy'), MemUnitStalled=mean.get('MemUnitStalled'))
if 'FETCH_SIZE' in mean and 'WRITE_SIZE' in mean:
    rd, wr = mean['FETCH_SIZE'] * 1024 * 2, mean['WRITE_SIZE'] * 1024     # gfx950: FETCH_SIZE x 2 (MI355X_MICROARCH.md)
    rec.update(hbm_read_bytes_per_pixel=rd / n, hbm_write_bytes_per_pixel=wr / n, hbm_traffic_bytes_per_pixel=(rd + wr) / n)
if stats:
    rec['frac_8TBs_rocprof_avg'] = line['bytes_per_pixel'] * n / (stats['avg_ms'] * 1e-3) / 8e12
print(json.dumps(rec))
PY
done
find $O -name "*.csv" -delete
cat $O/summary.jsonl
