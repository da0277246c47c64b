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
# with work queued on the current stream and on a side stream that joins it, on a slice of a larger buffer
buf = torch.zeros(16, device="cuda")
a = torch.randn(4096, 4096, device="cuda")
side = torch.cuda.Stream()
for _ in range(20):
    b = a @ a
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(20):
        c2 = a @ a
torch.cuda.current_stream().wait_stream(side)
buf[2:4] += rank + 1
dist.all_reduce(buf[2:4])
print(f"[{rank}] slice after queued work ok {buf[:5].tolist()}", flush=True)
if len(sys.argv) > 1:      # the library's own kernels first (a generator forward), then the exchange
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from reid_gan_amd import engine
    from reid_gan_amd.synth import synth_state, synth_batch
    from oracle import ried_oracle as O
    st = engine.State(synth_state([(n, s) for n, s, _ in O.rigan_spec()], 101), "cuda")
    G = engine.Generator(st, 4, 64, 64)
    x, _ = synth_batch(4, 64, seed=3)
    out = G.forward(x.cuda())
    print(f"[{rank}] generator forward queued", flush=True)
    dist.all_reduce(buf[2:4])
    print(f"[{rank}] exchange after the generator forward ok", flush=True)
    torch.cuda.synchronize()
    print(f"[{rank}] synchronized {out.float().abs().mean().item():.3f}", flush=True)
dist.destroy_process_group()
