import sys, json
sys.path.insert(0, '/root/repo')
import torch, bench
r = bench.config3_fusion(torch.device('cuda:0'))
print(json.dumps({k: r[k] for k in ('ms_per_step', 'bags_per_s')}))
