"""The calibration workload (run under rocprofv3 --pmc by tools/pmc_calibrate.sh): every pattern of calib.hip once per round, a few
rounds, buffers far larger than the 32 MiB of L2 and (for the streaming patterns) than the 256 MiB Infinity Cache."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libpmc_calib.so"))
lib.cal_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
big = torch.zeros(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB
out = torch.zeros(4, dtype=torch.float32, device=dev)
g = torch.Generator(device="cpu").manual_seed(0)
N_STREAM = 1 << 26  # float4s: 1 GiB
idx_lines = torch.randint(0, (1 << 28) // 16, (1 << 20,), generator=g).int().to(dev)       # 1M random 64-byte lines of the buffer
idx_words = (idx_lines.long() * 16 + torch.randint(0, 16, (1 << 20,), generator=g).to(dev)).int()
keys = torch.full((1 << 20,), 2 ** 62, dtype=torch.int64, device=dev)  # 8 MB of 64-bit keys (the rasteriser's frame at B=16 is 8 MB)
idx_keys = torch.randint(0, 1 << 20, (400000,), generator=g).int().to(dev)
rows = torch.zeros((140000, 16), dtype=torch.float32, device=dev)  # 9 MB of 64-byte gradient rows
idx_rows = torch.randint(0, 140000, (140000,), generator=g).int().to(dev)
s = torch.cuda.current_stream().cuda_stream
p = lambda t: ctypes.c_void_p(t.data_ptr())
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    assert lib.cal_run(0, p(big), None, N_STREAM, p(out), s) == 0          # cal_fill16: 1 GiB written
    assert lib.cal_run(1, p(big), None, 1 << 28, p(out), s) == 0           # cal_fill4: 1 GiB written, 4 bytes per lane
    assert lib.cal_run(2, p(big), None, N_STREAM, p(out), s) == 0          # cal_read16: 1 GiB read
    assert lib.cal_run(3, p(big), None, 1 << 20, p(out), s) == 0           # cal_store_per_line: 1M stores of 4 B, 1M lines
    assert lib.cal_run(4, p(big), p(idx_words), 1 << 20, p(out), s) == 0   # cal_gather4: 1M random 4-byte loads (+ 4 MB of indices)
    assert lib.cal_run(5, p(keys), p(idx_keys), 400000, p(out), s) == 0    # cal_atomic_min64: 4e5 atomics on 8 MB of keys
    assert lib.cal_run(6, p(rows), p(idx_rows), 140000, p(out), s) == 0    # cal_atomic_rows: 1.4e5 line-coalesced row adds
    torch.cuda.synchronize()
print("calibration workload done")
