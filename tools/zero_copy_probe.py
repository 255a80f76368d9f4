"""Probe: the step kernel reading actions from / writing results to PINNED HOST memory directly (one kernel node, no copy nodes)
against the staged round trip (H2D + kernel + 7 D2H copies; So100VecEnv(use_graph=False)).   python tools/zero_copy_probe.py [envs]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so100_mujoco_rl_amd.lib import So100Sim, StepIO, F_CUBE_PINNED, F_NOPADS, _check
from so100_mujoco_rl_amd.vec_env import So100VecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for flags, name in ((F_CUBE_PINNED, "free"), (F_NOPADS, "nopads")):
    sim = So100Sim(1, n, flags=flags, seed=1); sim.reset(); od = sim.obs_dim
    pin = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, pin_memory=True)
    h_act, h_obs, h_rew, h_done, h_trunc = pin(n, 6), pin(n, od), pin(n), pin(n, dt=torch.uint8), pin(n, dt=torch.uint8)
    h_tobs, h_epr, h_epl = pin(n, od), pin(n), pin(n, dt=torch.int32)
    io = StepIO(h_act.data_ptr(), h_obs.data_ptr(), h_rew.data_ptr(), h_done.data_ptr(), h_trunc.data_ptr(), h_tobs.data_ptr(),
                h_epr.data_ptr(), h_epl.data_ptr(), None, None)
    st = torch.cuda.current_stream()
    def step_zero_copy():
        _check(sim.L.so100_step(sim.h, C.byref(io), C.c_void_p(st.cuda_stream)), "so100_step")
    rs = np.random.RandomState(0)
    for it in range(5): step_zero_copy()
    st.synchronize()
    # correctness against the device-buffer path from the same state
    sim2 = So100Sim(1, n, flags=flags, seed=1); sim2.reset()
    sim3 = So100Sim(1, n, flags=flags, seed=1); sim3.reset()
    a = rs.uniform(-1, 1, (n, 6)).astype(np.float32)
    h_act.numpy()[...] = a
    io2 = StepIO(h_act.data_ptr(), h_obs.data_ptr(), h_rew.data_ptr(), h_done.data_ptr(), h_trunc.data_ptr(), h_tobs.data_ptr(), h_epr.data_ptr(), h_epl.data_ptr(), None, None)
    _check(sim2.L.so100_step(sim2.h, C.byref(io2), C.c_void_p(st.cuda_stream)), "so100_step"); st.synchronize()
    o3, r3, d3, t3 = sim3.step(torch.from_numpy(a).cuda()); st.synchronize()
    print(name, "zero-copy == device path:", bool((o3.cpu() == h_obs).all()), bool((r3.cpu() == h_rew).all()))
    for mode in ("eager", "graph"):
        if mode == "graph":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g): step_zero_copy()
            run = g.replay
        else:
            run = step_zero_copy
        for it in range(20): run(); st.synchronize()
        t0 = time.perf_counter()
        for it in range(300):
            h_act.numpy()[...] = a
            run(); st.synchronize()
            o = h_obs.numpy().copy(); r = h_rew.numpy().copy(); d = h_done.numpy().astype(bool)
        dt = (time.perf_counter() - t0)/300
        print(f"{name:7s} N={n}: zero-copy {mode}: {dt*1e6:7.1f} us per step (incl. host-side copies of actions / obs / rew / done)")
    env = So100VecEnv("Env01-v1", n, flags=flags, seed=1, use_graph=False); env.reset()
    for it in range(20): env.step_async(a); env.step_wait()
    t0 = time.perf_counter()
    for it in range(300): env.step_async(a); env.step_wait()
    print(f"{name:7s} N={n}: So100VecEnv staged:   {(time.perf_counter() - t0)/300*1e6:7.1f} us per step")
