"""bisect: which level-2 kernel miscounts (genome-sampled reads, as test_ctr_k31_large_checksums; how the missing s_nop
behind the 16-byte inline-assembly stores and the spill of an in-flight register were found)"""
import os, sys, pathlib, subprocess
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from kmertools_amd import device
    n, L, k = int(sys.argv[2]), 150, 31
    G = int(sys.argv[3])
    ctx = device.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    bases = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_reads(0x6b6d6572 + 3, n, L, bases, offsets, genome_len=G)
    ctr = device.Counter(ctx, k, int(sys.argv[4]))
    ctr.add_reads(bases, offsets, n)
    d = ctr.size()
    keys = torch.empty(d, dtype=torch.int64, device="cuda"); counts = torch.empty(d, dtype=torch.int32, device="cuda")
    ctr.export(keys, counts, d)
    print("n", n, "slots", sys.argv[4], "distinct", d, "sum", int(counts.to(torch.int64).sum()), "want", n * (L - k + 1), flush=True)
    os._exit(0)
for env in ({"KT_P2_SWWC": "0", "KT_P2_FAST": "0"}, {"KT_P2_SWWC": "0"}, {}):
    for n, G, slots in ((500000, 1000000, 1 << 27), (100000, 1000000, 1 << 25), (20000, 200000, 1 << 23), (500000, 1 << 30, 1 << 27)):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child", str(n), str(G), str(slots)], env=e, capture_output=True, text=True)
        print(env, r.stdout.strip()[-200:], r.stderr.strip()[-300:] if r.returncode else "")
