import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"; os.environ["LOCAL_RANK"] = "0"
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import bench
from kir_graph_amd import _lib
from kir_graph_amd.engine import DeviceIndex
dev = _lib.Device(0)
sidx, gidx, sample, rec, table = bench.build_inputs(1031, 100000)
dindex = DeviceIndex(dev, gidx); mates = dev.put(rec)
out = bench.run_steps(2, dev, dindex, gidx, mates, table, sample.gene_cn, "pv")
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
# cohort all-gather path
from kir_graph_amd import cohort
comm = cohort.Comm(2)
print("rank", comm.rank, "world", comm.world, "allreduce", float(t.item()), "calls", out[0][:3])
g = comm.allgatherDepths([{"a": 1.0, "b": 2.0}, {"a": 3.0, "b": 4.0}])
print("allgather", g)
dist.destroy_process_group()
