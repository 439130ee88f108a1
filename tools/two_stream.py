"""Experiment: hide the segment scan by iterating two half-batches on two streams.
Each half is its own handle (own non-blocking stream); S is doubled so that each half still offers 256 workgroups.
   python tools/two_stream.py [halves=2] [segments=32] [workload=cw_rendezvous]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_library_amd as pkg
H = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
wl = sys.argv[3] if len(sys.argv) > 3 else "cw_rendezvous"
B = 4096
full = getattr(pkg, wl)(N=1000, batch=B)
sol = [pkg.Solver(full.slice(i * B // H, (i + 1) * B // H), pkg.Options(rho=0.05, segments=S)) for i in range(H)]
def run(n, chunk):
    for _ in range(n // chunk):
        for s in sol:
            s.run(chunk, residual_every=1, sync=False)
    for s in sol:
        s.sync()
for chunk in (1, 2, 10):
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        run(100, chunk)
    t0 = time.perf_counter(); run(400, chunk); dt = time.perf_counter() - t0
    print(f"{wl} halves={H} S={S} chunk={chunk}: {400 / dt:.0f} batch-it/s ({dt / 400 * 1e6:.1f} us per iteration of all {B} QPs)", flush=True)
