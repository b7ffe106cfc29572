import numpy as np, sys
sys.path.insert(0,'/root/repo')
import particlesystem_amd as ps
from particlesystem_amd.slab import step_local
n=1<<20
halves=[ps.ParticleSystem(ps.default_config(rank=r,world=2)) for r in range(2)]
xyz=halves[0].uniform_cloud(n,2026)
age=np.random.default_rng(2026).uniform(15/7,7.5,n).astype(np.float32)
for h in halves: h.fill_particles(xyz,age=age,fert_age=np.full(n,1e6,np.float32))
for st in range(3):
    step_local(halves)
    for r,h in enumerate(halves):
        print(st,r,[int(h.msg_download(6+k,64)[0]) for k in (0,1)],[int(h.msg_download(8+k,64)[0]) for k in (0,1)],h.counters['relocations'],h.counters['integrated'])
