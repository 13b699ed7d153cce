set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/placement_pmc2
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL --output-format json -d $O/run -- python3 tools/placement.py 21600 0 0 40 80 > $O/placement.out 2> $O/placement.err || { tail -5 $O/placement.err; }
grep -v amdgpu $O/placement.out
ls -la $(find $O -name "*.json")
python - <<'PY'
import json, glob, collections
f = glob.glob('gpurun_out/placement_pmc2/run/**/*results.json', recursive=True)[0]
d = json.load(open(f))
top = d['rocprofiler-sdk-tool'][0]
print(top.keys())
cc = top.get('callback_records', {}).get('counter_collection', [])
print(len(cc))
if cc:
    print(json.dumps(cc[0])[:1500])
PY
