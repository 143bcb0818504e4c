#!/usr/bin/env python3
"""Two batches in flight: RCM of batch i + 1 on one handle / stream / host thread while Permute2D of batch i runs on
another (independent batches: the same matrix here).  Prints ms per step next to the sequential figure and checks the
pipelined outputs against the sequential ones bit for bit."""
import ctypes as C, os, sys, threading, time, queue
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import capi, ops, synth
lib = capi.load()
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
val = (torch.arange(nnz, device="cuda", dtype=torch.int32) % 1021).to(torch.float32)
p = lambda t: C.c_void_p(t.data_ptr())
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
# sequential reference (the bench's step)
order = torch.empty(n, dtype=torch.int32, device="cuda")
out = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
for _ in range(3):
    ops.rcm_reorder(rp, col, out=order); ops.permute_csr(n, n, rp, col, val, order, order, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    ops.rcm_reorder(rp, col, out=order); ops.permute_csr(n, n, rp, col, val, order, order, out=out)
torch.cuda.synchronize(); seq_ms = (time.perf_counter() - t0) / steps * 1e3
want = [t.clone() for t in out]
# pipelined: two handles, two streams, two host threads, two order buffers
def make():
    h = C.c_void_p(); assert lib.sbx_create(0, C.byref(h)) == 0
    s = torch.cuda.Stream(); assert lib.sbx_set_stream(h, C.c_void_p(s.cuda_stream)) == 0
    return h, s
h_r, s_r = make(); h_p, s_p = make()
orders = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(2)]
outs = [tuple(torch.empty_like(t) for t in out) for _ in range(2)]
ready, free = queue.Queue(), queue.Queue()
for i in range(2): free.put(i)
def producer(count):
    for _ in range(count):
        i = free.get()
        st = capi.RcmStats()
        rc = lib.sbx_rcm_reorder(h_r, 0, n, nnz, p(rp), p(col), p(orders[i]), C.byref(st)); assert rc == 0, lib.sbx_last_error(h_r)
        ev = torch.cuda.Event(); ev.record(s_r)
        ready.put((i, ev))
def consumer(count):
    for _ in range(count):
        i, ev = ready.get()
        s_p.wait_event(ev)
        o = outs[i]
        rc = lib.sbx_permute_csr(h_p, 0, 3, n, n, nnz, p(rp), p(col), p(val), p(orders[i]), p(orders[i]), p(o[0]), p(o[1]), p(o[2]))
        assert rc == 0, lib.sbx_last_error(h_p)
        ev2 = torch.cuda.Event(); ev2.record(s_p); ev2.synchronize()   # (the order buffer is free again)
        free.put(i)
def run(count):
    a, b = threading.Thread(target=producer, args=(count,)), threading.Thread(target=consumer, args=(count,))
    a.start(); b.start(); a.join(); b.join(); torch.cuda.synchronize()
run(4)
t0 = time.perf_counter(); run(steps); pipe_ms = (time.perf_counter() - t0) / steps * 1e3
same = all(torch.equal(a, b) for o in outs for a, b in zip(o, want))
print("sequential %.3f ms/step, two batches in flight %.3f ms/step (%.2f x), outputs identical: %s" % (seq_ms, pipe_ms, seq_ms / pipe_ms, same))
lib.sbx_sync(h_r); lib.sbx_sync(h_p); lib.sbx_destroy(h_r); lib.sbx_destroy(h_p)
