cd "$GRAFT_REPO_ROOT"
for m in synth_lstm_h18x2 synth_lstm_h10x2 synth_lstm_io lstm; do
python bench.py --model $m --streams 1024 --launch resident --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$m', 'xRT', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'err', j['max_abs_err_vs_oracle'])
"
done
