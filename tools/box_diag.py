"""What kind of box is this?  oligo k=4 step time next to plain fill / copy rates and what rocm-smi says about
clocks and partitions (boxes of the pool differ by ~15 % on the store-bound kernels and not on the others).
usage: python tools/box_diag.py"""
import json, subprocess, sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch

def smi(*a):
    try:
        return subprocess.run(["rocm-smi", *a], capture_output=True, text=True, timeout=30).stdout.strip()
    except Exception as e:
        return "rocm-smi failed: %r" % (e,)

def rate(fn, nbytes, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9

x = torch.empty(10 * (1 << 30) // 8, dtype=torch.int64, device="cuda")
y = torch.empty_like(x)
print("fill  10 GiB: %.0f GB/s" % rate(lambda: x.fill_(7), x.numel() * 8))
print("copy  10 GiB: %.0f GB/s (read + write)" % rate(lambda: y.copy_(x), 2 * x.numel() * 8))
print("read  10 GiB (sum): %.0f GB/s" % rate(lambda: x.sum(), x.numel() * 8))
del x, y
out = subprocess.run([sys.executable, "bench.py", "--workload", "comp_oligo_k4", "--steps", "20", "--warmup", "5", "--no-cpu"],
                     capture_output=True, text=True, cwd=str(pathlib.Path(__file__).resolve().parent.parent)).stdout
j = json.loads(out.strip().splitlines()[-1])
print("oligo k=4: %.3f ms per step, frac %.3f" % (j["ms_per_step"], j["roofline"]["frac"]))
print(smi("--showclocks"))
print(smi("--showmemorypartition", "--showcomputepartition"))
print(smi("--showpower", "--showperflevel"))
