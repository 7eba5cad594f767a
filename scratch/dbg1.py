import sys; sys.path.insert(0,'.')
import numpy as np
from bfqzip_amd import api
from oracle import orc
e = api.Engine(0, m=5, M=0)
for seed in (100, 100, 101):
    sp = api.synth_spec(3000, 50, seed=seed, coverage=25)
    b,q,r = api.synth_host(sp)
    ob,oq,st = orc.run_reads(b,q,r, orc.params(m=5, M=0))
    hb,hq,hst = e.run_reads(b,q,r)
    print("oracle", st)
    print("gpu   ", {k:(hex(v) if v>2**32 else v) for k,v in hst.items()})
