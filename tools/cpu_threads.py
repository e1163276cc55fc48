import sys, time, os
sys.path.insert(0, '.')
import numpy as np
from threadpoolctl import threadpool_limits
from larynx_amd import hparams as HP, synthetic
from oracle import hifi_gan_np
vsd = synthetic.make_hifigan_state_dict(HP.HIFIGAN_HIGH, seed=1234)
mel = np.random.default_rng(0).standard_normal((80, 64)).astype(np.float32)
print('cpus', os.cpu_count())
for n in (8, 16, 32, 64, 128):
    with threadpool_limits(limits=n):
        t = time.perf_counter(); hifi_gan_np.hifigan_infer(vsd, HP.HIFIGAN_HIGH, mel); dt = time.perf_counter() - t
    print(n, 'threads', round(dt, 2), 's for 64 frames')
