"""Which primitives of a CU-free gradient exchange work on this runtime?  Two processes time-sharing cuda:0 (the rehearsal form of two
ranks with one GPU each): IPC memory handles (dmabuf), hipStreamWriteValue32 / hipStreamWaitValue32 on plain and on signal memory -
local and through an IPC mapping -, device-to-device copies out of a peer's mapping.  Prints one line per primitive.
    python scripts/ipc_probe.py            (spawns its second process itself)
"""
import ctypes, os, sys, time
import torch
import torch.multiprocessing as mp

HIP = None


def hip():
    global HIP
    if HIP is None:
        HIP = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        HIP.hipGetErrorString.restype = ctypes.c_char_p
    return HIP


def ck(rc, what):
    ok = rc == 0
    hip().hipGetLastError()                    # (clears the sticky error: torch would otherwise raise it on its next call)
    print(f"  [{os.getpid()}] {what}: {'ok' if ok else 'FAILED rc=%d %s' % (rc, hip().hipGetErrorString(rc).decode())}", flush=True)
    return ok


class IpcMemHandle(ctypes.Structure):
    _fields_ = [("reserved", ctypes.c_ubyte * 64)]      # (c_char would hand back bytes cut at the first NUL)


def child(rank, q01, q10):
    torch.cuda.set_device(0)
    H = hip()
    qin, qout = (q10, q01) if rank == 0 else (q01, q10)
    s = torch.cuda.Stream()
    sp = ctypes.c_void_p(s.cuda_stream)
    attr = ctypes.c_int(0)
    H.hipDeviceGetAttribute(ctypes.byref(attr), 10071 if False else 0, 0)        # (placeholder; the enum value is looked up below)
    # -- plain device memory: a data buffer and a flag word, exported over IPC
    # (own hipMalloc allocations: an IPC handle names a whole allocation, and torch's caching allocator hands out pieces of its blocks)
    class Raw:
        def __init__(self, nbytes):
            self.p = ctypes.c_void_p(); assert H.hipMalloc(ctypes.byref(self.p), ctypes.c_size_t(nbytes)) == 0
            H.hipMemset(self.p, 0, ctypes.c_size_t(nbytes)); self.n = nbytes
        def data_ptr(self): return self.p.value
    data, flag = Raw(4 << 20), Raw(256)
    fill = torch.full((1 << 20,), float(rank + 1), device="cuda")
    H.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    H.hipMemcpy(ctypes.c_void_p(data.data_ptr()), ctypes.c_void_p(fill.data_ptr()), ctypes.c_size_t(4 << 20), 3)
    torch.cuda.synchronize()
    hd, hf = IpcMemHandle(), IpcMemHandle()
    ok = ck(H.hipIpcGetMemHandle(ctypes.byref(hd), ctypes.c_void_p(data.data_ptr())), "hipIpcGetMemHandle(data)")
    ok &= ck(H.hipIpcGetMemHandle(ctypes.byref(hf), ctypes.c_void_p(flag.data_ptr())), "hipIpcGetMemHandle(flag)")
    # signal memory
    sig = ctypes.c_void_p()
    sig_ok = ck(H.hipExtMallocWithFlags(ctypes.byref(sig), ctypes.c_size_t(8), ctypes.c_uint(2)), "hipExtMallocWithFlags(hipMallocSignalMemory)")
    hs = IpcMemHandle()
    sig_ipc = sig_ok and ck(H.hipIpcGetMemHandle(ctypes.byref(hs), sig), "hipIpcGetMemHandle(signal memory)")
    if sig_ok:
        H.hipMemset(sig, 0, ctypes.c_size_t(8))
    qout.put((bytes(hd.reserved), bytes(hf.reserved), bytes(hs.reserved) if sig_ipc else None, data.data_ptr(), flag.data_ptr()))
    pd, pf, ps, _, _ = qin.get(timeout=60)
    # -- open the peer's buffers
    def open_(raw, what):
        h = IpcMemHandle(); ctypes.memmove(ctypes.byref(h), raw, 64)
        p = ctypes.c_void_p()
        return p if ck(H.hipIpcOpenMemHandle(ctypes.byref(p), h, ctypes.c_uint(1)), f"hipIpcOpenMemHandle({what})") else None
    peer_data, peer_flag = open_(pd, "peer data"), open_(pf, "peer flag")
    peer_sig = open_(ps, "peer signal memory") if ps else None
    # -- local stream write / wait on plain memory
    H.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
    H.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
    w_plain = ck(H.hipStreamWriteValue32(sp, ctypes.c_void_p(flag.data_ptr() + 4), 7, 0), "hipStreamWriteValue32(own plain memory)")
    wt_plain = ck(H.hipStreamWaitValue32(sp, ctypes.c_void_p(flag.data_ptr() + 4), 7, 0, 0xffffffff), "hipStreamWaitValue32(own plain memory, GEQ)")
    s.synchronize()
    w = torch.zeros(4, dtype=torch.int32, device="cuda")
    H.hipMemcpy(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(flag.data_ptr()), ctypes.c_size_t(16), 3)
    print(f"  [{os.getpid()}] own flag word after stream write: {int(w[1])}", flush=True)
    # -- the hand-off: each rank waits (on its copy stream) until the PEER has written 1 into its flag word 0 through the IPC mapping,
    #    then copies the peer's data buffer device-to-device and checks it
    got = torch.zeros(1 << 20, device="cuda")
    t0 = time.time()
    if peer_flag is not None and peer_data is not None:
        with torch.cuda.stream(s):
            rc_wait = H.hipStreamWaitValue32(sp, ctypes.c_void_p(flag.data_ptr()), 1, 0, 0xffffffff)
            ck(rc_wait, "hipStreamWaitValue32(own flag, to be written by the peer)")
            H.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
            ck(H.hipMemcpyAsync(ctypes.c_void_p(got.data_ptr()), peer_data, ctypes.c_size_t(4 << 20), 3, sp), "hipMemcpyAsync(peer mapping -> own, D2D)")
        time.sleep(0.5 + 0.5 * rank)              # the peer's write arrives LATER than the wait was enqueued
        s2 = torch.cuda.Stream()
        ck(H.hipStreamWriteValue32(ctypes.c_void_p(s2.cuda_stream), peer_flag, 1, 0), "hipStreamWriteValue32(peer flag through the IPC mapping)")
        s2.synchronize()
        done = False
        for _ in range(100):
            if s.query():
                done = True
                break
            time.sleep(0.1)
        print(f"  [{os.getpid()}] rank {rank}: wait + copy {'completed' if done else 'DID NOT COMPLETE in 10 s'} after {time.time() - t0:.2f} s; "
              f"copied value {float(got[0]) if done else 'n/a'} (expected {float(2 - rank)})", flush=True)
        if not done:
            one = torch.ones(1, dtype=torch.int32, device="cuda")          # release the stream so that the process can exit
            H.hipMemcpy(ctypes.c_void_p(flag.data_ptr()), ctypes.c_void_p(one.data_ptr()), ctypes.c_size_t(4), 3)
            torch.cuda.synchronize()
    qout.put("done"); qin.get(timeout=60)


if __name__ == "__main__":
    mp.set_start_method("spawn")
    q01, q10 = mp.Queue(), mp.Queue()
    ps = [mp.Process(target=child, args=(r, q01, q10)) for r in range(2)]
    for p in ps: p.start()
    for p in ps:
        p.join(120)
        if p.is_alive():
            print("a process hung: killed"); p.kill()
