import os, torch, torch.distributed as dist
dist.init_process_group("gloo")
r = dist.get_rank()
torch.cuda.set_device(0)
x = torch.full((3, 4), float(r + 1), device="cuda")
full = torch.empty((6, 4), device="cuda")
try:
    dist.all_gather_into_tensor(full, x)
    torch.cuda.synchronize()
    print(r, "all_gather_into_tensor cuda ok", full[:, 0].tolist())
except Exception as e:
    print(r, "all_gather_into_tensor cuda FAILED:", repr(e)[:200])
t = torch.tensor([r + 5], dtype=torch.int32, device="cuda")
try:
    dist.all_reduce(t, op=dist.ReduceOp.MAX); print(r, "all_reduce max cuda ok", t.item())
except Exception as e:
    print(r, "all_reduce FAILED", repr(e)[:200])
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    try:
        dist.all_gather_into_tensor(full, x * 2); s.synchronize(); print(r, "side-stream gather ok", full[:, 0].tolist())
    except Exception as e:
        print(r, "side-stream FAILED", repr(e)[:200])
dist.destroy_process_group()
