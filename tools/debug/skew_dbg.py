import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from algoplonk_amd import setup as ap_setup
from algoplonk_amd._lib import lib, check
from helpers import CURVES
from oracle.prng import SplitMix64, tau_from_seed
cv, ov = CURVES["bn254"]
n = 1 << 17
srs = ap_setup.unsafe_srs(cv, n, tau_from_seed(3, cv.r), device=0)
ctx = C.c_void_p()
check(lib.apk_msm_ctx_create(cv.abi, 0, srs.g1, n + 3, 0, C.byref(ctx)))
out = C.create_string_buffer(64)
g = SplitMix64(1)
case = sys.argv[1]
sc = {"ones": [1] * n, "bytes": [g.below(256) for _ in range(n)], "uniform": [g.fr(cv.r) for _ in range(n)]}[case]
d = C.c_void_p()
check(lib.apk_device_alloc(ctx, 32 * n, C.byref(d)))
buf = cv.fr_vector(sc)
check(lib.apk_device_upload(ctx, d, buf, len(buf)))
for _ in range(6):
    t0 = time.perf_counter(); check(lib.apk_msm_g1_device(ctx, 0, d, n, out)); t = time.perf_counter() - t0
print(case, round(t * 1e3, 3))
