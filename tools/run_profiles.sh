# Validation on the MI355X box: GPU tests, bench line, rocprofv3 kernel stats and
# the two PMC passes; results under gpurun_out/TAG.
#   bash tools/run_profiles.sh TAG [bench.py arguments, e.g. --dtype float32 --math mixed]
# SKIP_TESTS=1 leaves the test suite out.
set -e
TAG=${1:-r01}
shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
rm -rf $O && mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
  tail -2 $O/pytest_gpu.log
fi
timeout -k 10 500 python bench.py "$@" > $O/bench_line.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench_line.json
# profiled runs: one raster layout per kernel symbol (--no-plain), nothing but the headline configuration (--no-configs)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py "$@" --no-cpu-baseline --no-plain --no-configs --steps 40 --warmup 5 > $O/bench_line_under_rocprof.json 2> $O/rocprof_stats.err
# the dominant kernel over the TIMED dispatches only (the stats file averages every call of the symbol:
# warm-up, the extra steps of the sensor leg, ...): tools/timed_dispatches.py names what it leaves out
find $O/stats -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/timed_dispatches.py {} 5 40 > $O/kernel_timed_dispatches.json || true
cat $O/kernel_timed_dispatches.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py "$@" --no-cpu-baseline --no-parity --no-plain --no-configs --steps 3 --warmup 1 > $O/pmc_fetch.out 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py "$@" --no-cpu-baseline --no-parity --no-plain --no-configs --steps 3 --warmup 1 > $O/pmc_write.out 2> $O/pmc_write.err
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
O=$O python - <<'PY'
import csv, glob, json, os
O = os.environ['O']
res = {}
for name in ('FETCH_SIZE', 'WRITE_SIZE'):
    d = os.path.join(O, 'pmc_fetch' if name == 'FETCH_SIZE' else 'pmc_write')
    vals = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'et_stream_kernel' in r['Kernel_Name'] and r['Counter_Name'] == name:
                vals.append(float(r['Counter_Value']))
    res[name] = (sum(vals) / len(vals) if vals else None, len(vals))
print(json.dumps(res))
json.dump(res, open(os.path.join(O, 'pmc_raw.json'), 'w'))
# the record bench.py reads back as roofline.traffic (gfx950: FETCH_SIZE x 2, MI355X_MICROARCH.md HBM)
try:
    line = json.load(open(os.path.join(O, 'bench_line_under_rocprof.json')))
    n = line['roofline']['pixels_per_launch']
    rd = res['FETCH_SIZE'][0] * 1024 * 2
    wr = res['WRITE_SIZE'][0] * 1024
    import hashlib, sys
    sys.path.insert(0, os.getcwd())
    from mod16_amd import _lib
    info = {}
    try:
        info = json.load(open(os.path.join('mod16_amd', 'build_info.json')))
    except (OSError, ValueError):
        pass
    # the record belongs to the library that produced it: bench.py reports it only for this build
    rec = {'build_id': _lib.build_id(), 'lib_sha256': hashlib.sha256(open(_lib.LIB_PATH, 'rb').read()).hexdigest(),
           'git_commit': info.get('git_commit') if info.get('build_id') == _lib.build_id() else None,
           'git_dirty_at_build': info.get('git_dirty') if info.get('build_id') == _lib.build_id() else None,
           'kernel': line['roofline']['kernel'], 'layout': line['config']['raster_layout']['layout'],
           'dtype': {'f64': 'float64', 'f32': 'float32'}[line['dtype']],
           'launches_averaged': [res['FETCH_SIZE'][1], res['WRITE_SIZE'][1]],
           'FETCH_SIZE_KB_per_launch': res['FETCH_SIZE'][0], 'WRITE_SIZE_KB_per_launch': res['WRITE_SIZE'][0],
           'correction': 'gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16 B/lane streaming reads -> x2; WRITE_SIZE exact',
           'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr, 'traffic_bytes_per_launch': rd + wr,
           'pixels_per_launch': n, 'traffic_bytes_per_pixel': (rd + wr) / n,
           'algorithmic_bytes_per_pixel': line['roofline']['bytes_per_pixel']}
    json.dump(rec, open(os.path.join(O, 'pmc_hbm_traffic.json'), 'w'), indent=1)
    print(json.dumps(rec))
except Exception as exc:
    print('no traffic record:', exc)
PY
# keep the merged output small
find $O -name "*.csv" -size +2M -delete
du -sh $O
