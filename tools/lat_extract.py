import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], "latency_ms", d["latency_ms"], "python", d["latency_python_ms"]["mean"])
