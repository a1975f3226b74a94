import json, sys, torch
sys.path.insert(0, '/root/repo')
import bench
print(json.dumps(bench.bench_prefill(1024, 3072, 65536, torch.device('cuda:0')), indent=1))
