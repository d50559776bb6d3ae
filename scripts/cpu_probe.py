import time, numpy as np, sys, os
sys.path.insert(0,'.')
from oracle import cpu_step
rng=np.random.RandomState(0)
M,K,N=20000,512,2048
A=rng.randn(M,K).astype(np.float32); B=rng.randn(K,N).astype(np.float32)
cpu_step.gemm(A,B); t=time.time()
for _ in range(3): cpu_step.gemm(A,B)
dt=(time.time()-t)/3
print('OMP', os.environ.get('OMP_NUM_THREADS'), 'threads', cpu_step.num_threads(), '%.1f ms %.1f GFLOP/s'%(dt*1e3, 2*M*K*N/dt/1e9), flush=True)
