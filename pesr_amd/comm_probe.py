"""Child process of `comm.probe_direct`: brings the direct RCCL communicator up among the probe children of all ranks, runs one
gradient-sized all-reduce with a known answer, and exits 0.  If RCCL's C API through ctypes were to hang or fail on a node with
several GPUs - a path no single-GPU box can execute - it hangs or fails HERE, in a process its parent can kill, and the training
processes fall back to torch.distributed together instead of hanging in `ncclCommInitRank` (replaces nothing of the reference:
the guard of the replacement of nn.DataParallel, reference train.py:114-118).

    python -m pesr_amd.comm_probe <rank> <world> <cuda device index> <addr> <port> <rendezvous timeout s> [rccl|peer]
(`peer`: the same rehearsal for the peer-memory transport, comm.PeerCopy - IPC mappings between the GPUs, stream wait / write-value
operations across them, peer copies.)
"""
import datetime
import sys


def main(argv) -> int:
    rank, world, dev, addr, port, timeout = int(argv[0]), int(argv[1]), int(argv[2]), argv[3], int(argv[4]), float(argv[5])
    kind = argv[6] if len(argv) > 6 else "rccl"
    import torch
    import torch.distributed as dist
    from pesr_amd import comm

    def mark(what):                       # (the parent quotes the last mark if it has to kill this process)
        print(f"comm_probe: {what}", file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        print("comm_probe: no GPU visible", file=sys.stderr)
        return 3
    device = torch.device("cuda", dev)
    torch.cuda.set_device(device)
    # bootstrap over gloo: nothing of ProcessGroupNCCL in this process, the unique id travels as a CPU tensor
    mark("rendezvous")
    dist.init_process_group("gloo", init_method=f"tcp://{addr}:{port}", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=timeout))
    mark("communicator")
    tr = comm.PeerCopy(device, rank, world, None) if kind == "peer" else comm.DirectRccl(device, rank, world, None)   # (each includes a small self-test)
    mark("32 MB all-reduces")
    n = 8 << 20                                                   # one 32 MB bucket
    ramp = (torch.arange(n, device=device) % 7).float()
    t = torch.empty(n, dtype=torch.float32, device=device)
    for rnd in range(6):
        # fresh values every round, written by a kernel right before the exchange: a peer that read this GPU's memory before the
        # kernel's stores had left the caches - or a rank that summed a stale copy - would see the PREVIOUS round's numbers
        torch.add(ramp, float((rank + 1) * (rnd + 1)), out=t)
        tr.wait([tr.all_reduce_async(t)])
        torch.cuda.synchronize(device)
        want = ramp * world + float(world * (world + 1) // 2 * (rnd + 1))
        if not torch.equal(t, want):
            bad = int((t != want).sum())
            print(f"comm_probe: all-reduce round {rnd}: {bad} of {n} elements wrong (first: {float(t[t != want][0])}, expected {float(want[t != want][0])})", file=sys.stderr)
            return 4
    mark("host_max")
    m = tr.host_max([float(rank)])
    if m != [float(world - 1)]:
        print(f"comm_probe: MAX all-reduce gave {m}", file=sys.stderr)
        return 5
    mark("close")
    tr.close()
    dist.destroy_process_group()
    mark("done")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
