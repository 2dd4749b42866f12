import sys, numpy as np, time, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import blockgen, vp9ref, bench
hip = vp9ref.load_hip()
rng=np.random.default_rng(1440)
b=blockgen.gen_blocks(rng,2560,1440,hip.BLOCK_DTYPE,intra_frac=0.08,skip_frac=0.35)
c,e=blockgen.gen_coeffs(rng,b,2560,1440,8)
P=bench.frame_params(hip,2560,1440,8)
pk=hip.Packer()
for i in range(3): pk.pack_only(P,b,e)
import subprocess
print(subprocess.run(f"ls /proc/{os.getpid()}/task | wc -l", shell=True, capture_output=True).stdout)
t=time.perf_counter()
for i in range(50): pk.pack_only(P,b,e)
print((time.perf_counter()-t)/50*1e3, "ms")
