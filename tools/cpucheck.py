import os, time, multiprocessing as mp
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
def burn(n):
    x=0
    for i in range(n): x+=i*i
    return x
if __name__=="__main__":
    for p in (1,8,32,64,128,256):
        t0=time.time()
        with mp.Pool(p) as pool: pool.map(burn,[4_000_000]*p)
        print(p, "procs:", round(time.time()-t0,2), "s")
