#!/bin/bash
# mixed-size batches (BASELINE.json configs[4]): stream s runs at ratio (0.0, 0.34, 0.67, 1.0)[s % 4]; one launch per
# size group, back to back on one stream (forking the groups onto their own streams was tried: the events cost more
# than the overlap buys - 85 vs 35 us per step for slimmable_wavenet.nam)
cd "$GRAFT_REPO_ROOT"
for MODE in serial; do
  for m in slimmable_wavenet slimmable_container A2; do
    python bench.py --model $m --streams 768 --slim-mix --launch block --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null > /tmp/mix.json
    python - "$MODE" "$m" <<'PY'
import json, sys
j = json.load(open("/tmp/mix.json"))
print(sys.argv[1], sys.argv[2], "mixed sizes, 768 streams: xRT", j["value"], "us/step", round(j["ms_per_step"] * 1e3, 2), "host enqueue us/step",
      j["host_enqueue_us_per_step"], "err", j["max_abs_err_vs_oracle"], "resident xRT", j["resident_launch"]["value"])
PY
  done
done
