"""torch (bundled ROCm runtime) + RCCL process group + libfba_hip.so (system ROCm runtime) in one process."""
import os, sys
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
import fba_pomdp_amd as fba
eng = fba.Engine("episodic-tiger", model=fba.MODEL_BA_TABLE, sims=256, particles=256, slots=256, runs=1 << 20, episodes=8)
eng.run_ticks(3)
x = torch.tensor([float(eng.counters().sim_steps)], dtype=torch.float64, device="cuda")
dist.all_reduce(x); dist.barrier(); torch.cuda.synchronize()
eng.run_ticks(2)
print("coexist ok", x.item(), eng.counters().sim_steps)
dist.destroy_process_group()
