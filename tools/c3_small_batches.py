import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch, altro_amd, bench
for batch in (64, 128, 512, 8192, 65536):
    bt, set_guess = bench.make_lane_batch("c3", batch, 0, 50, 0)
    for rep in range(2):
        bt.reset_duals(1.0); set_guess(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = bt.ilqr_solve(iterations_max=80, use_backtracking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    it = res["iterations"]
    print("batch %6d: %.2f ms, sweeps %d, mean it %.2f, converged %d, >10 iterations: %d, >20: %d  -> %.3f ms per sweep" % (batch, dt*1e3, res["sweeps"], it.mean(), (res["status"]==0).sum(), (it>10).sum(), (it>20).sum(), dt*1e3/max(res["sweeps"],1)))
    bt.close()
