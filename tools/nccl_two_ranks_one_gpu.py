"""Can RCCL run two ranks on ONE GPU?  (N-rank readiness on a one-GPU box)
Two processes, both on cuda:0, backend nccl, one all_reduce and one all_to_all_single; prints
what happened - the exact error text is the record when RCCL refuses.
  python tools/nccl_two_ranks_one_gpu.py"""
import os, socket, sys, traceback
import torch, torch.distributed as dist, torch.multiprocessing as mp


def work(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        x = torch.full((4,), float(rank + 1), device="cuda")
        dist.all_reduce(x)
        torch.cuda.synchronize()
        a = torch.arange(4, device="cuda", dtype=torch.int64) + 10 * rank
        b = torch.empty_like(a)
        dist.all_to_all_single(b, a)
        torch.cuda.synchronize()
        print("rank %d: nccl on a shared GPU WORKED: all_reduce -> %s, all_to_all -> %s" % (rank, x.tolist(), b.tolist()), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print("rank %d: nccl on a shared GPU FAILED: %s: %s" % (rank, type(e).__name__, str(e).strip().splitlines()[-1][:400]), flush=True)


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(work, args=(2, port), nprocs=2, join=True)
