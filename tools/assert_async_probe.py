"""What a failed torch._assert_async looks like on this stack (ROCm): run in a child process, print how it ended and what it wrote.
The 500-step Fauna line of round 4 aborted ONCE with "HSA hardware exception"; estimate_bones guards its leg quadrants with exactly such
a device-side assert (model/geometry/skinning.py: foot_of).  python tools/assert_async_probe.py"""
import subprocess
import sys

CHILD = r"""
import torch
x = torch.zeros(4, device="cuda")
torch._assert_async((x.sum() > 1.0), "estimate_bones: no vertex in a leg quadrant (probe)")
y = (x + 1).sum()
torch.cuda.synchronize()
print("child survived", float(y))
"""
r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=300)
print("return code:", r.returncode)
print("stdout:", r.stdout[-500:])
print("stderr tail:", r.stderr[-1500:])
