cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4c_pytest.log
python bench.py > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4c_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'))
PY
