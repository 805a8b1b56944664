"""Does the ctr k=31 step depend on where its buffers lie?  One process runs bench.py's ctr_k31 workload several times with
a spacer allocation of a different size held alive in front of everything the workload and the library allocate."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.argv = [sys.argv[0], "--workload", "ctr_k31", "--steps", "6", "--warmup", "1", "--no-cpu", "--no-genome"]
import bench
args = bench.parse()
env = bench.Env(args)
torch = env.torch
for gb in (0, 3, 11, 23, 37, 53, 71, 0):
    spacer = torch.empty(gb << 30, dtype=torch.uint8, device="cuda") if gb else None
    res, _ = bench.run_workload(env, "ctr_k31", args)
    print("spacer %2d GiB: %.3f ms per step (median %.3f, min %.3f)" % (gb, res["ms_per_step"], res["ms_median"], res["ms_min"]), flush=True)
    del spacer
    torch.cuda.empty_cache()
env.close()
