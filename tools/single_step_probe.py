import sys, os, time
sys.path.insert(0, os.getcwd())
from optionslab_amd import _hip
_hip.lib(); _hip.profile_enable(True)
for N in (10_000, 50_000, 100_000, 1_000_000):
    for M in (1, 2, 4, 8, 16, 64, 252):
        for _ in range(3): _hip.european(100.,100.,1.,.05,.2,0.,True,N,M,1)
        _hip.profile_reset()
        t=time.perf_counter()
        for i in range(20): _hip.european(100.,100.,1.,.05,.2,0.,True,N,M,i)
        wall=(time.perf_counter()-t)/20
        n,ms=_hip.kernel_time()
        print(f"N={N:8d} M={M:4d} kernel {ms/n*1e3:8.1f} us  wall {wall*1e6:8.1f} us")
