import sys; sys.path.insert(0,'/root/repo')
from pytdscf_amd import engine as E
import time
for it in (2000, 20000, 200000, 2000000):
    print(it, E.clock_probe(it))
time.sleep(0.5)
print("after idle", E.clock_probe(2000), E.clock_probe(2000))
