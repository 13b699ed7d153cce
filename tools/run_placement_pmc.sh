set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/placement_pmc
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/run -- python3 tools/placement.py 21600 0 0 40 80 120 0 > $O/placement.out 2> $O/placement.err || { tail -5 $O/placement.err; }
cat $O/placement.out | grep -v amdgpu
python - <<'PY'
import csv, glob, collections, json
O = 'gpurun_out/placement_pmc'
rows = []
for f in glob.glob(O + '/run/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
by = collections.defaultdict(dict)
names = {}
for r in rows:
    by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
    names[int(r['Dispatch_Id'])] = r['Kernel_Name']
dur = {}
for f in glob.glob(O + '/run/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[int(r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
group, out = -1, collections.defaultdict(list)
for d in sorted(names):
    if 'synth_kernel' in names[d]:
        group += 1
    elif 'et_stream_kernel' in names[d]:
        out[group].append((dur.get(d), by[d]))
for g in sorted(out):
    v = out[g]
    n = len(v)
    agg = {k: sum(x[1].get(k, 0) for x in v) / n for k in v[0][1]}
    print(json.dumps({'placement': g, 'launches': n, 'ms': round(sum(x[0] for x in v) / n, 3), **{k: round(a, 1) for k, a in agg.items()}}))
PY
find $O -name "*.csv" -size +1M -delete
