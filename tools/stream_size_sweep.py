import sys,time
sys.path.insert(0,".")
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
Q=synth.unit_rows(2,0,4)
for rows in (32_768, 131_072, 262_144, 524_288, 1_000_000, 2_000_000, 4_000_000, 5_900_000):
    idx=dawn.VectorIndex(0); idx.fill_synthetic(1,0,rows,1)
    for (u,t) in ((3,512),(3,256),(8,512)):
        idx.set_option("shadow_scan_unroll",u); idx.set_option("shadow_scan_threads",t); idx.set_option("shadow_scan_blocks",256)
        for i in range(20): idx.search(Q[i%4],10)
        idx.profile_enable(True)
        for i in range(200): idx.search(Q[i%4],10)
        n,ms=idx.profile_read(); idx.profile_enable(False)
        print(rows,u,t,round(ms/n*1e3,2),"us", flush=True)
    idx.close()
