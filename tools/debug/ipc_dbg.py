import ctypes as C, multiprocessing as mp, os, sys, time
ROOT="/root/repo"
def worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
    os.environ["APK_COMM_TIMEOUT_S"]="20"
    try:
        from algoplonk_amd import _lib, parallel, plonk as ap_plonk, setup as ap_setup, MarshalProof
        from algoplonk_amd._lib import lib, check
        from helpers import CURVES, blinding, random_chain_ccs
        from oracle.prng import tau_from_seed
        cv, ov = CURVES["bn254"]
        comm = parallel.Comm(rank, world, "127.0.0.1", port)
        ccs, w, sol = random_chain_ccs(cv, 10, 0xA190 + 10)
        srs = ap_setup.unsafe_srs(cv, ccs.domain_size(), tau_from_seed(99, cv.r), device=0)
        pk, vk = ap_plonk.Setup(ccs, srs, device=0)
        comm.bind(pk.ctx)
        print(rank, "transport", comm.transport, flush=True)
        bl = blinding(cv, 5)
        if rank == 0:
            plain = MarshalProof(ap_plonk.Prove(ccs, pk, w, bl))
            comm.split_begin()
            t=time.time()
            got = MarshalProof(ap_plonk.Prove(ccs, pk, w, bl))
            print("split proof", got == plain, time.time()-t, flush=True)
            comm.split_end()
        else:
            print("served", comm.serve(), flush=True)
        comm.close(); pk.close()
        q.put((rank,"ok"))
    except Exception as e:
        import traceback; traceback.print_exc()
        print(rank, "ERR", e, lib.apk_last_error(), flush=True)
        q.put((rank,"fail"))
if __name__=="__main__":
    sys.path.insert(0, ROOT)
    from algoplonk_amd.parallel import free_port
    ctx=mp.get_context("spawn"); q=ctx.Queue(); port=free_port()
    ps=[ctx.Process(target=worker,args=(r,2,port,q)) for r in range(2)]
    [p.start() for p in ps]
    for p in ps: p.join(timeout=100)
    for p in ps:
        if p.is_alive(): print("killing", p.pid); p.kill()
