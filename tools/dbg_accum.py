import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, oracle
from _cases import fixtures
import malstroem_amd.algorithms as alg
fx = fixtures(); fd = fx['flowdir_noflats']
a = alg.flow.accumulated_flow(fd); o = oracle.accumulated_flow(fd)
bad = np.argwhere(a != o)
print('shape', fd.shape, 'mismatch', len(bad), 'sum', a.sum(), o.sum())
for r,c in bad[:15]: print(r,c,'got',a[r,c],'want',o[r,c],'fd',fd[r,c])
print('tile rows of mismatches', np.unique(bad[:,0]//64), 'cols', np.unique(bad[:,1]//64))
print('zeros', (a==0).sum(), 'less', (a<o).sum(), 'more', (a>o).sum())
