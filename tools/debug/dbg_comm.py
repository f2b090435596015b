"""Debug driver: tests/test_gpu_comm.py's worker under faulthandler (where does a silent rank sit?)."""
import faulthandler, sys, os, time, multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)

def worker(rank, world, port, q, *a):
    import test_gpu_comm as T
    faulthandler.dump_traceback_later(100, exit=False, file=sys.stderr)
    t = time.time()
    T._worker(rank, world, port, q, *a)
    print("rank", rank, "done in", time.time() - t, flush=True)

if __name__ == "__main__":
    from algoplonk_amd.parallel import free_port
    for args in [("bn254", False, True), ("bn254", False, True)]:
        ctx = mp.get_context("spawn")
        q = ctx.Queue(); port = free_port()
        t = time.time()
        procs = [ctx.Process(target=worker, args=(r, 2, port, q) + args) for r in range(2)]
        [p.start() for p in procs]
        try:
            res = [q.get(timeout=150) for _ in procs]
            print(res)
        except Exception as e:
            print("TIMEOUT", repr(e))
        [p.join(timeout=10) for p in procs]
        [p.kill() for p in procs if p.is_alive()]
        print("elapsed", time.time() - t, flush=True)
