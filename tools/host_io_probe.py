import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
r = bench.host_io(os.path.join(bench.ROOT, "tests/golden/models/wavenet_a1_standard.nam"), 256, True, budget_s=0.5)
print(json.dumps({k: ({m: v[m]["us_per_call"] for m in v if isinstance(v[m], dict)} if isinstance(v, dict) else v) for k, v in r.items() if k.endswith("frames")}))
for n in (1, 16):
    r = bench.host_io(os.path.join(bench.ROOT, "tests/golden/models/wavenet_a1_standard.nam"), n, True, budget_s=0.3)
    print(n, "streams", json.dumps({k: ({m: v[m]["us_per_call"] for m in v if isinstance(v[m], dict)} if isinstance(v, dict) else v) for k, v in r.items() if k.endswith("frames")}))
