"""Two ranks on one GPU; rank 1's first all-reduce of the peer-memory transport is never enqueued (patched out here): rank 0's
self-test cannot complete.  Both ranks must come out of comm.PeerCopy's constructor with a CommError, with a device that still
synchronises and a process group that still works (tests/test_dp_gpu.py::test_peer_copy_constructor_gives_up_together)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from pesr_amd import comm


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    os.environ["PESR_PEER_SELFTEST_TIMEOUT"] = "3"
    if rank == 1:
        real = comm.PeerCopy.all_reduce_async
        calls = [0]

        def patched(self, t):
            calls[0] += 1
            if calls[0] == 1:
                self._lookup(t)                  # (the registration is a collective of the bootstrap group: keep it)
                e = torch.cuda.Event(); e.record(torch.cuda.current_stream(dev))
                return e
            return real(self, t)
        comm.PeerCopy.all_reduce_async = patched
    try:
        comm.PeerCopy(dev, rank, world, None)
        print(f"rank {rank}: constructor returned", flush=True)
        sys.exit(3)
    except comm.CommError as e:
        print(f"rank {rank}: CommError: {e}", flush=True)
    torch.cuda.synchronize(dev)
    x = torch.ones(1024, device=dev) * (rank + 1)
    assert float(x.sum()) == 1024.0 * (rank + 1)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    assert float(t) == 1.0
    dist.destroy_process_group()
    print(f"rank {rank}: ok", flush=True)


if __name__ == "__main__":
    main()
