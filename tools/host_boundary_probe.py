import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
g = bench.golden()
host = bench.synthesize('sp1', 1 << 20, 0x5A4B5602, g, 64)
import torch
dev = torch.device('cuda', 0)
sh = bench.Shard(host, dev, g)
os.environ['ZKV_HOST_TRACE'] = '1'
for fs in ('', '32768', '131072'):
    if fs: os.environ['ZKV_HOST_FIRST_SEGMENT'] = fs
    print('first segment', fs or 'default', bench.host_boundary_rate(sh, repeats=2))
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    t0 = time.perf_counter(); sh.enqueue(stream); torch.cuda.synchronize(); print('resident ms', (time.perf_counter() - t0) * 1e3)
