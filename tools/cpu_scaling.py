import subprocess, time, sys
for n in (1, 4, 8, 16, 32, 64):
    t0 = time.perf_counter()
    ps = [subprocess.Popen(["./kmertools_amd/bin/kmertools", "debug-fixed6", "300000"], stdout=subprocess.DEVNULL) for _ in range(n)]
    for p in ps: p.wait()
    print(n, "procs: %.2f s" % (time.perf_counter() - t0), flush=True)
