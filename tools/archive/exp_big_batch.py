import sys, time
sys.path.insert(0, '.')
import torch
from weather2alert_amd import HeatAlertVecEnv, synth, tables
sd = synth.make_synth("nn_full_medicare_all", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd)
dev = torch.device("cuda:0")
for n in [int(x) for x in sys.argv[1:]] or (8388608, 8388608 + 13):
    env = HeatAlertVecEnv(n, tables=ct, device=dev)
    obs, _ = env.reset(seed=1)
    a = (torch.rand(n, device=dev) < 0.1).to(torch.uint8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(160):
        obs, r, d, _, info = env.step(a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 160
    st = env.state()
    assert (st["episode_no"] == 1).all() and (st["t"] == 7).all() and (st["used"] <= st["budget"]).all()
    assert torch.isfinite(obs).all() and torch.isfinite(r).all() and (r <= 0).all()
    # last env row is written correctly (64-bit obs offsets)
    assert obs[-1, ct.feature_names.index("dos")] == 6 and env.check_status() == 0
    print(f"n={n}: {dt*1e6:.1f} us/step = {n/dt/1e9:.2f} G env-steps/s on one GPU")
    env.close()
