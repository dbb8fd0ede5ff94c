import os, time, torch, torch.multiprocessing as mp
def prod(q, q2):
    torch.cuda.set_device(0)
    t = torch.full((1 << 20,), 3.0, device="cuda")
    q.put(t)
    print("producer sent; env HSA_ENABLE_IPC_MODE_LEGACY =", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), flush=True)
    print("consumer says:", q2.get(timeout=60), flush=True)
    t.add_(1); torch.cuda.synchronize()
    q.put("changed"); print("consumer says:", q2.get(timeout=60), flush=True)
def cons(q, q2):
    torch.cuda.set_device(0)
    try:
        t = q.get(timeout=60)
        q2.put(f"got tensor on {t.device}, value {float(t[0])}, ptr {t.data_ptr():x}")
        q.get(timeout=60); torch.cuda.synchronize()
        q2.put(f"after producer add: {float(t[5])}")
    except Exception as e:
        q2.put(f"FAILED: {type(e).__name__}: {e}"); q2.put("x")
if __name__ == "__main__":
    mp.set_start_method("spawn")
    q, q2 = mp.Queue(), mp.Queue()
    a = mp.Process(target=prod, args=(q, q2)); b = mp.Process(target=cons, args=(q, q2))
    a.start(); b.start(); a.join(90); b.join(90)
