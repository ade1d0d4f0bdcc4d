"""dev: what one replica exchange of the 512-chain ensemble costs with and without a process group (MTRPL_FORCE_DIST=1):
sync() = all-gather of [rows x (n_T + n_params + 1)], swap replay, table re-upload"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist
dev = torch.device("cuda", 0)
force = os.environ.get("MTRPL_FORCE_DIST") == "1"
if force:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from metrotrpl_amd import pt, pt_bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = pt_bench.setup(n, dev, 0)
ens = pt.PTEnsemble(S["ens0"]["chain_states"], S["T"], S["evaluator"], rng_mode="counter", group=None if force else pt.LOCAL_ONLY)
ens.initialise()
ens.advance(S["proposer"], 10, 10, False)
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter(); ens.sync(); torch.cuda.synchronize(); t1 = time.perf_counter()
    owner, att, accs = pt.replica_exchange(ens.table, ens.draws(1000 + rep, pt.SWAP)); t2 = time.perf_counter()
    ens.cur_tab.copy_(torch.from_numpy(np.ascontiguousarray(ens.table[ens.lo:ens.hi]))); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"force_dist={force}: sync {1e3*(t1-t0):.2f} ms, swap replay {1e3*(t2-t1):.2f} ms, table upload {1e3*(t3-t2):.2f} ms")
t0 = time.perf_counter(); ens.advance(S["proposer"], 10, 10, False); torch.cuda.synchronize(); print("10 iterations incl. one exchange: %.2f ms" % (1e3 * (time.perf_counter() - t0)))
t0 = time.perf_counter(); ens.advance(S["proposer"], 9, 1000, False); torch.cuda.synchronize(); print("9 iterations, no exchange: %.2f ms" % (1e3 * (time.perf_counter() - t0)))
if force:
    dist.destroy_process_group()
