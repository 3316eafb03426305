import sys, pathlib, tempfile
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import parity_cases as P
from backends import GpuBackend
from reseq_amd import synth
d=pathlib.Path(tempfile.mkdtemp())
p=P.Pair(GpuBackend, d, "tiny_e2e", synth.TINY, [5000, 80, 3210], seed=7, num_pairs=3000)
p.align_normalization()
ofr=p.osim.sieve(1,9); o1,o2=p.osim.create_reads(ofr)
bfr,b1,b2=p.b.pairs(1,9)
print(len(o1),len(b1),len(o2),len(b2), ofr.tobytes()==bfr.tobytes())
for o,b in ((o1,b1),(o2,b2)):
    n=min(len(o),len(b))
    diff=[i for i in range(n) if o[i]!=b[i]]
    print('ndiff',len(diff), diff[:20])
    if diff:
        i=diff[0]; print(o[max(0,i-80):i+80]); print(b[max(0,i-80):i+80])
