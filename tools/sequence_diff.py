"""Per-frame distance between the trajectory of the bundled sequence computed here and the reference's
published one, and the same for the map points in insertion order."""
import os
import tempfile
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import run_sequence
from tests.bundled import load
s,r,vo=run_sequence.run(tempfile.mkdtemp(prefix='picp_seqdiff_'), quiet=True)
b=load()
pub=b['published_trajectory']
est=np.array([p[:3,3] for p in r['trajectory']])
d=np.abs(est-pub[:len(est),1:4]).max(axis=1)
for f in range(0,len(d),4): print(f, ' '.join(f'{x:.1e}' for x in d[f:f+4]))
pp=b['published_world_points']; wp=np.array(r['world_points'])
dw=np.abs(wp-pp[:,1:4]).max(axis=1)
print('map diffs by insertion index (every 20):',' '.join(f'{x:.1e}' for x in dw[::20]))
