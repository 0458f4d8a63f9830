import time, torch, sys
sys.path.insert(0,'.')
from alignn_amd import GraphBatch
from alignn_amd.synthetic import make_batch
raw = make_batch(64, 60)
for i in range(3):
    torch.cuda.synchronize(); t=time.perf_counter()
    b = GraphBatch.from_raw(raw, device='cuda')
    torch.cuda.synchronize(); print('from_raw (incl. H2D of COO) ms', (time.perf_counter()-t)*1e3)
import numpy as np
print('lg COO bytes', raw.lg_u.nbytes*2/1e6, 'MB')
