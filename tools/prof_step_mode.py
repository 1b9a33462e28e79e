import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from gym_cloth_amd.envs import ClothVecEnv
E=512
cfg=bench.bench_cfg(25,0.02)
env=ClothVecEnv(cfg,n_envs=E,precision="f32",consume_domrand_draws=False)
for e in range(E): env.np_randoms[e]=np.random.RandomState(1000+e)
env.reset()
acts=np.stack([np.random.RandomState(2000+e).uniform(-1,1,size=(8,4)) for e in range(E)],axis=1)
env.step(acts[0])
t0=time.perf_counter(); kms=0
pr=cProfile.Profile(); pr.enable()
for t in range(1,6):
    obs,rew,done,info=env.step(acts[t]); kms+=env.batch.last_kernel_ms
pr.disable()
dt=time.perf_counter()-t0
print("5 steps: wall %.1f ms per step, stepper kernel %.1f ms per step" % (dt/5*1e3, kms/5))
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
