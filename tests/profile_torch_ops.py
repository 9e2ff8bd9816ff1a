"""Developer aid: which aten ops (stock torch glue around the HIP kernels) cost device time in a bench.py step.

    python tests/profile_torch_ops.py prototype_full|source_only        # on the GPU box
"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--workload", sys.argv[1], "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
import runpy
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    runpy.run_path("bench.py", run_name="__main__")
rows = prof.key_averages(group_by_input_shape=True)
rows = sorted(rows, key=lambda r: -r.device_time_total)
for r in rows[:60]:
    if r.key.startswith("aten::") and r.device_time_total > 0:
        print("%-40s calls %4d  dev_us %9.0f  %s" % (r.key, r.count, r.device_time_total, str(r.input_shapes)[:110]))
