python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "layernorm" 2>&1 | tail -3
python -m pytest tests/test_hip_model.py -x -q -m gpu -k "parity or batch or graph or full_size" 2>&1 | tail -3
for i in 1 2; do
python bench.py --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lazy ', d['value'], d['ms_per_step'])"
TR_LN_EAGER=1 python bench.py --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager', d['value'], d['ms_per_step'])"
done
