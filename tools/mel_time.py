import sys, torch
sys.path.insert(0, '/root/repo')
import bench
from mod_extraction_amd import models
dev = torch.device('cuda:0')
m = models.Spectral2DCNN(**bench.CNN_CFG).to(dev)
x = torch.rand(256, 2, 88200, device=dev) * 2 - 1
for _ in range(3): y = m.log_mel(x, (0, 0, 0, 0))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): y = m.log_mel(x, (0, 0, 0, 0))
b.record(); torch.cuda.synchronize()
print('logmel ms', a.elapsed_time(b) / 20, float(y.double().sum()))
torch.save(y.cpu(), sys.argv[1])
