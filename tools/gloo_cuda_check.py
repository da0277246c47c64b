"""Does gloo all-reduce CUDA tensors between two ranks that share one card on this box?  (rehearsal precondition)"""
import faulthandler, os, sys, torch, torch.distributed as dist
faulthandler.dump_traceback_later(40, exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
c = torch.ones(4) * (rank + 1)
dist.all_reduce(c)
print(f"[{rank}] cpu tensor ok {c.tolist()}", flush=True)
t = torch.ones(4, device="cuda") * (rank + 1)
torch.cuda.synchronize()
dist.all_reduce(t)
print(f"[{rank}] cuda tensor ok {t.tolist()}", flush=True)
dist.destroy_process_group()
