set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/placement_pmc2
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL --output-format csv -d $O/run -- python3 tools/placement.py 21600 0 0 40 80 > $O/placement.out 2> $O/placement.err || { tail -5 $O/placement.err; }
grep -v amdgpu $O/placement.out
f=$(find $O -name "*counter_collection.csv" | head -1)
head -3 $f
python - <<'PY'
import csv, glob, collections, json
O = 'gpurun_out/placement_pmc2'
f = glob.glob(O + '/run/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
print(len(rows))
PY
